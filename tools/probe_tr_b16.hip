#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s4 lds_s4;
// hypothesis: within each 16-lane group, lane t supplies the address of 4 contiguous 16-bit elements = row (t>>2), cols 4*(t&3)..+3 of a
// [4][16] block; lane t receives column t of the block (elements [0..3][t]).
__global__ void k(unsigned short* out, int stride_elems) {
    __shared__ __attribute__((aligned(16))) unsigned short sm[16 * 64];
    const int lane = threadIdx.x;
    for (int e = lane; e < 16 * 64; e += 64) sm[e] = 0xFFFF;
    __syncthreads();
    for (int e = lane; e < 256; e += 64) { int kk = e / 16, n = e % 16; sm[kk * stride_elems + n] = (unsigned short)(kk * 16 + n); }
    __syncthreads();
    const int i = lane & 15, q = lane >> 4;
    unsigned short* p = sm + (4 * q + (i >> 2)) * stride_elems + 4 * (i & 3);
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)p);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)v[j];
}
__global__ void mm(const float* A, const float* B, float* C) {   // C[16][16] = A[16][16] * B[16][16] via one bf16 MFMA, A row-major [i][k], B row-major [k][j]
    const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
    typedef __bf16 b4 __attribute__((ext_vector_type(4)));
    s4 a, b;
    for (int j = 0; j < 4; ++j) {
        __bf16 x = (__bf16)A[i * 16 + 4 * q + j], y = (__bf16)B[(4 * q + j) * 16 + i];
        a[j] = __builtin_bit_cast(short, x); b[j] = __builtin_bit_cast(short, y);
    }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(4 * q + r) * 16 + i] = c[r];
}
int main() {
    unsigned short* d; hipMalloc(&d, 512);
    unsigned short h[256];
    for (int stride : {16, 20, 80}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
        hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { int want = (4 * (l >> 4) + j) * 16 + (l & 15); if (h[l * 4 + j] != want) ++bad; }
        printf("tr_b16 stride %d: %s (lane0: %d %d %d %d; lane17: %d %d %d %d)\n", stride, bad ? "MISMATCH" : "ok as hypothesised", h[0], h[1], h[2], h[3], h[68], h[69], h[70], h[71]);
    }
    float hA[256], hB[256], hC[256], *dA, *dB, *dC;
    for (int e = 0; e < 256; ++e) { hA[e] = (float)((e * 7) % 13 - 6); hB[e] = (float)((e * 5) % 11 - 5); }
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 1024);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mm, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int kk = 0; kk < 16; ++kk) s += hA[i * 16 + kk] * hB[kk * 16 + j]; if (s != hC[i * 16 + j]) ++bad; }
    printf("mfma_f32_16x16x16_bf16 operand layout: %s\n", bad ? "MISMATCH" : "ok");
    return 0;
}
