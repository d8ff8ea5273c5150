// Cost of a per-timestep exchange between the workgroups of a row group (the structure a column-split ConvLSTM sequence kernel needs):
// G groups of P workgroups; per round every workgroup writes its slice of a [64 x C] fp32 tile, releases, bumps the group's counter,
// spins until all P arrived, acquires, reads the whole tile.  Prints us per round.   hipcc --offload-arch=gfx950 -O3 -o probe_xwg_sync ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void xwg_kernel(float* buf, unsigned* counters, int P, int C, int T, int lds_spin, unsigned* err) {
    extern __shared__ float sm[];
    const int grp = blockIdx.x / P, part = blockIdx.x % P, tid = threadIdx.x;
    const int slice = C / P;                         // channels of this workgroup
    float acc = 0.f;
    for (int t = 0; t < T; ++t) {
        float* tile = buf + ((size_t)(t & 1) * gridDim.x / P + grp) * 64 * C;
        // write the slice: 64 rows x slice channels
        for (int e = tid; e < 64 * slice; e += 256) {
            const int r = e / slice, c = e % slice;
            const float v = acc + (float)(t + r + c);
            if (MODE == 2) __hip_atomic_store(tile + r * C + part * slice + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else tile[r * C + part * slice + c] = v;
        }
        if (MODE == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(counters + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned want = (unsigned)P * (t + 1);
                int spins = 0;
                while (__hip_atomic_load(counters + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 2000000) { *err = 1; break; }
                }
            }
            __syncthreads();
        } else if (MODE == 1) {
            __atomic_thread_fence(__ATOMIC_RELEASE);             // (agent scope by default in HIP)
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(counters + grp, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned want = (unsigned)P * (t + 1);
                int spins = 0;
                while (__hip_atomic_load(counters + grp, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 2000000) { *err = 1; break; }
                }
            }
            __syncthreads();
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        } else {
            __syncthreads();
        }
        // read the whole tile (64 x C fp32), 16 bytes per lane
        const float4* t4 = reinterpret_cast<const float4*>(tile);
        float s = 0.f;
        if (MODE == 2) {
            for (int e = tid; e < 64 * C; e += 256) s += __hip_atomic_load(tile + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            for (int e = tid; e < 64 * C / 4; e += 256) {
                const float4 v = t4[e];
                s += v.x + v.y + v.z + v.w;
            }
        }
        // stand-in for the MFMA phase: spin on LDS a little
        for (int k = 0; k < lds_spin; ++k) { sm[tid] = s; __syncthreads(); s += sm[(tid + 1) & 255] * 1e-9f; }
        acc = s * 1e-9f;
    }
    if (acc == 12345.f) buf[0] = acc;
}

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 10, P = argc > 2 ? atoi(argv[2]) : 12, C = argc > 3 ? atoi(argv[3]) : 384, T = argc > 4 ? atoi(argv[4]) : 21;
    const int spin = argc > 5 ? atoi(argv[5]) : 0;
    float* buf; unsigned* cnt; unsigned* err;
    hipMalloc(&buf, (size_t)2 * G * 64 * C * 4); hipMalloc(&cnt, G * 4); hipMalloc(&err, 4);
    hipMemset(err, 0, 4);
    hipFuncSetAttribute((const void*)xwg_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)xwg_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void*)xwg_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipMemset(cnt, 0, G * 4);
            hipDeviceSynchronize();
            hipEventRecord(a);
            if (mode == 2) hipLaunchKernelGGL(xwg_kernel<2>, dim3(G * P), dim3(256), 100 * 1024, 0, buf, cnt, P, C, T, spin, err);
            else if (mode) hipLaunchKernelGGL(xwg_kernel<1>, dim3(G * P), dim3(256), 100 * 1024, 0, buf, cnt, P, C, T, spin, err);
            else hipLaunchKernelGGL(xwg_kernel<0>, dim3(G * P), dim3(256), 100 * 1024, 0, buf, cnt, P, C, T, spin, err);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        unsigned e; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
        printf("G=%d P=%d C=%d T=%d spin=%d mode=%s: %.1f us total, %.2f us per round, err=%u\n", G, P, C, T, spin, mode == 2 ? "coherent-accesses" : mode ? "exchange" : "no-sync", best * 1000, best * 1000 / T, e);
    }
    return 0;
}
