// Breakdown of the cross-workgroup exchange cost (see probe_xwg_sync.hip): flags bit0 = write slice (coherent stores), bit1 = counter
// sync, bit2 = read tile (coherent 4-byte loads), bit3 = read tile with plain 16-byte loads, bit4 = fetch_add returns (no separate spin load
// before the first check), bit5 = no s_sleep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void k(float* buf, unsigned* counters, int P, int C, int T, int flags, unsigned* err) {
    const int grp = blockIdx.x / P, part = blockIdx.x % P, tid = threadIdx.x;
    const int slice = C / P;
    float acc = 0.f;
    for (int t = 0; t < T; ++t) {
        float* tile = buf + ((size_t)(t & 1) * gridDim.x / P + grp) * 64 * C;
        if (flags & 1)
            for (int e = tid; e < 64 * slice; e += 256) {
                const int r = e / slice, c = e % slice;
                __hip_atomic_store(tile + r * C + part * slice + c, acc + (float)(t + r + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        if (flags & 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const unsigned want = (unsigned)P * (t + 1);
                unsigned seen = __hip_atomic_fetch_add(counters + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
                int spins = 0;
                while (seen < want) {
                    if (!(flags & 32)) __builtin_amdgcn_s_sleep(1);
                    seen = __hip_atomic_load(counters + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (++spins > 2000000) { *err = 1; break; }
                }
            }
            __syncthreads();
        }
        float s = 0.f;
        if (flags & 4) for (int e = tid; e < 64 * C; e += 256) s += __hip_atomic_load(tile + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (flags & 8) {
            const float4* t4 = reinterpret_cast<const float4*>(tile);
            for (int e = tid; e < 64 * C / 4; e += 256) { const float4 v = t4[e]; s += v.x + v.y + v.z + v.w; }
        }
        acc = s * 1e-9f;
    }
    if (acc == 12345.f) buf[0] = acc;
}
int main(int argc, char** argv) {
    const int G = atoi(argv[1]), P = atoi(argv[2]), C = atoi(argv[3]), T = atoi(argv[4]);
    float* buf; unsigned* cnt; unsigned* err;
    (void)hipMalloc(&buf, (size_t)2 * G * 64 * C * 4); (void)hipMalloc(&cnt, G * 4); (void)hipMalloc(&err, 4);
    (void)hipMemset(err, 0, 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int fl[] = {0, 1, 2, 2 | 32, 3, 4, 8, 1 | 2 | 4, 1 | 2 | 8, 1 | 2 | 8 | 32};
    for (int f : fl) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            (void)hipMemset(cnt, 0, G * 4);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(k, dim3(G * P), dim3(256), 0, 0, buf, cnt, P, C, T, f, err);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        unsigned e; (void)hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
        printf("G=%d P=%d C=%d T=%d flags=%2d (%s%s%s%s%s): %.2f us per round, err=%u\n", G, P, C, T, f, f & 1 ? "write " : "", f & 2 ? "sync " : "",
               f & 4 ? "read4 " : "", f & 8 ? "read16plain " : "", f & 32 ? "nosleep" : "", best * 1000 / T, e);
    }
    return 0;
}
