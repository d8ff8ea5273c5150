// Token-row contractions of the RVT backbone: weight gradients of the Linear layers and their per-stream workspace.  All tensors channels-last ("rows" = tokens of an NHWC map).
// C-ABI declared in include/leod_hip.h.
#include "linear_common.hpp"
#include "wgrad_bf16.hpp"
#include "wgrad_dma.hpp"

// Workspace of the weight-gradient kernels for launches on `stream` (wgrad_bf16.hpp: partial tiles, leod_workspace_bytes() bytes, 16-byte
// aligned, caller-owned and alive until replaced; ws == NULL withdraws it).
LEOD_API long leod_workspace_bytes() { return (long)kWgwScratchBytes; }
LEOD_API int leod_set_workspace(void* ws, long bytes, hipStream_t stream) {
    if (ws && (bytes <= 0 || (reinterpret_cast<uintptr_t>(ws) & 15))) return LEOD_ERR_ARG;
    wgrad_wide_register_scratch(stream, ws, (size_t)(bytes > 0 ? bytes : 0));
    return LEOD_OK;
}

// Profiling aid (LEOD_FAMILY_MARKERS=1, used by the PMC passes of tools/pmc_bench_traffic.sh only): one-thread marker kernels in front of
// and behind every launch of the roofline family, so that tools/roofline_traffic.py sums the HBM counters of exactly the dispatches the
// event probe of bench.py brackets (the kernel names alone do not separate the Linear weight gradients from the 1x1-conv ones).
__global__ void leod_family_marker_kernel(int begin) { (void)begin; }
struct FamilyMarker {
    hipStream_t s; bool on;
    explicit FamilyMarker(hipStream_t st) : s(st) {
        static const bool env = getenv("LEOD_FAMILY_MARKERS") && atoi(getenv("LEOD_FAMILY_MARKERS"));
        on = env;
        if (on) hipLaunchKernelGGL(leod_family_marker_kernel, dim3(1), dim3(1), 0, s, 1);
    }
    ~FamilyMarker() { if (on) hipLaunchKernelGGL(leod_family_marker_kernel, dim3(1), dim3(1), 0, s, 0); }
};

// dW[N,K] += dy[M,N]^T @ X[M,K] ; dbias[N] += colsum(dy)  with X = x, LN(x) (stats + ln_w/ln_b) or [x | x2]
LEOD_API int leod_linear_wgrad(const float* dy, long lddy, const float* x, long ldx, const float* stats,
                               const float* ln_w, const float* ln_b, const float* x2, long ldx2, int K1,
                               float* dW, float* dbias, int M, int N, int K, int dy_bf16, hipStream_t stream) {
    if (!dy || !x || !dW) return LEOD_ERR_ARG;
    FamilyMarker fm(stream);
    XRows xl{x, ldx, stats, ln_w, ln_b, x2, ldx2, K1};
    const int df = (dy_bf16 & 1) ? 1 : 0;
    if (dy_bf16 & 6) {                              // bit 1: x holds bf16 rows, bit 2: fp16 rows (precision mode 16f) -- the wide kernel only
        if (stats || x2) return LEOD_ERR_ARG;
        xl.fmt = (dy_bf16 & 4) ? 3 : 2;
        if (!use_wgrad_wide(xl, lddy, M, N, K, df) && !(wgrad_dma_ok(xl, lddy, M, N, K, df))) return LEOD_ERR_UNSUPPORTED;
    }
    if (wgrad_dma_ok(xl, lddy, M, N, K, df)) {
        const int rc = launch_wgrad_dma(dy, lddy, xl, dW, (long)K, dbias, M, N, K, stream, df);
        if (rc != LEOD_ERR_UNSUPPORTED) return rc;
    }
    if (use_wgrad_wide(xl, lddy, M, N, K, df)) return launch_wgrad_wide(dy, lddy, xl, dW, (long)K, dbias, M, N, K, stream, df);
    if (use_wgradw(M)) return launch_wgradw(dy, lddy, xl, dW, (long)K, dbias, M, N, K, stream, df);
    if (N % 48 == 0 && K % 48 == 0) return launch_wgrad16<3, 3>(dy, lddy, xl, dW, (long)K, dbias, M, N, K, stream, df);
    if (N % 32 == 0 && K % 32 == 0 && (N % 64 || K % 64)) return launch_wgrad16<2, 2>(dy, lddy, xl, dW, (long)K, dbias, M, N, K, stream, df);
    if (N >= 64 && K >= 64) return launch_wgrad16<4, 4>(dy, lddy, xl, dW, (long)K, dbias, M, N, K, stream, df);
    if (K >= 64) return launch_wgrad16<1, 4>(dy, lddy, xl, dW, (long)K, dbias, M, N, K, stream, df);
    if (N >= 64) return launch_wgrad16<4, 1>(dy, lddy, xl, dW, (long)K, dbias, M, N, K, stream, df);
    return launch_wgrad16<1, 1>(dy, lddy, xl, dW, (long)K, dbias, M, N, K, stream, df);
}

// n <= 4 Linear weight gradients of ONE row count M in one preparation launch, one contraction launch and one reduce launch (the LDS-DMA
// kernel of wgrad_dma.hpp with a problem table): the four weight gradients of an attention block -- qkv and fc1 from LayerNorm inputs, proj
// from the 16-bit attention rows, fc2 through GELU of the fp16 pre-activation (maxvit.py:110-118,252-270) -- which the block's backward
// issues together.  dy_fmt[k]: 0 fp32 rows, 1 bf16 rows.  x_fmt[k]: 0 fp32 rows, 1 fp32 rows through LayerNorm (stats / ln_w / ln_b of
// problem k), 2 fp16 pre-activation through GELU, 3 bf16 rows, 4 fp16 rows.  Dense rows (strides N[k] / K[k]).
// LEOD_ERR_UNSUPPORTED: not coverable (precision mode, row count, widths) -- nothing was launched, run the problems singly.
LEOD_API int leod_linear_wgrad_group(int n, const void* const* dy, const int* dy_fmt, const void* const* x, const int* x_fmt,
                                     const float* const* stats, const float* const* ln_w, const float* const* ln_b, float* const* dW,
                                     float* const* dbias, int M, const int* N, const int* K, hipStream_t stream) {
    if (n < 1 || n > 4 || !dy || !dy_fmt || !x || !x_fmt || !dW || !N || !K) return LEOD_ERR_ARG;
    WgdHostProb hp[4];
    int T = 0;
    size_t need = 0;
    for (int k = 0; k < n; ++k) {
        if (!dy[k] || !x[k] || !dW[k] || x_fmt[k] < 0 || x_fmt[k] > 4) return LEOD_ERR_ARG;
        XRows xl{reinterpret_cast<const float*>(x[k]), (long)K[k], nullptr, nullptr, nullptr, nullptr, 0, 0};
        if (x_fmt[k] == 1) {
            if (!stats || !ln_w || !ln_b || !stats[k] || !ln_w[k] || !ln_b[k]) return LEOD_ERR_ARG;
            xl.stats = stats[k]; xl.ln_w = ln_w[k]; xl.ln_b = ln_b[k];
        }
        xl.fmt = x_fmt[k] == 2 ? 1 : x_fmt[k] == 3 ? 2 : x_fmt[k] == 4 ? 3 : 0;
        const int t = wgd_tile(N[k], K[k]);
        if (leod_precision() != 1 || !t || (T && t != T) || M < 8192 || (M % kWgdRC) || M > (t == 6 ? 60000 : 400000) || (N[k] & 7) || (K[k] & 7))
            return LEOD_ERR_UNSUPPORTED;
        T = t;
        need += (size_t)M * ((dy_fmt[k] ? 0 : N[k]) + (xl.x_mode() == 3 ? 0 : K[k])) * 2;
        hp[k] = WgdHostProb{dy[k], (long)N[k], xl, dW[k], (long)K[k], dbias ? dbias[k] : nullptr, N[k], K[k], dy_fmt[k] ? 1 : 0};
    }
    if (need > kWgdOperandBytes) return LEOD_ERR_UNSUPPORTED;
    FamilyMarker fm(stream);
    switch (T) {
        case 6: return launch_wgrad_dma_group_t<6, 64, 3>(n, hp, M, stream);
        case 8: return launch_wgrad_dma_group_t<8, 64, 2>(n, hp, M, stream);
        default: return launch_wgrad_dma_group_t<4, 64, 3>(n, hp, M, stream);
    }
}

// 1: an attention block of this geometry may keep its attention output O and the gradient dO as bf16 rows in precision mode bf16 --
// every kernel that touches them has a 16-bit path: the bf16-tile attention kernels, proj forward (LDS-staged GEMM), the dgrad of proj
// (row epilogue) and the proj weight gradient (wide kernel)
extern "C" int leod_partition_attn_o16_ok(int B, int H, int W, int C, int heads, int ph, int pw);
LEOD_API int leod_attn_block_o16_ok(int B, int H, int W, int C, int heads, int ph, int pw) {
    const long M = (long)B * H * W;
    if (M > 0x7fffffffL || !leod_partition_attn_o16_ok(B, H, W, C, heads, ph, pw)) return 0;
    XRows xl{}; xl.ld = C; xl.fmt = 2;
    return (C % 8 == 0) && use_gemm_lds((int)M, cdiv(C, 16 * pick_nt(C))) && use_wgrad_wide(xl, (long)C, (int)M, C, C, 0);
}

// dW[N,K] += dy[M,N]^T @ gelu(u16[M,K]) ; dbias[N] += colsum(dy)
LEOD_API int leod_linear_wgrad_gelu16(const float* dy, long lddy, const void* u16, float* dW, float* dbias, int M, int N, int K,
                                      hipStream_t stream) {
    if (!dy || !u16 || !dW) return LEOD_ERR_ARG;
    FamilyMarker fm(stream);
    XRows xl{reinterpret_cast<const float*>(u16), (long)K, nullptr, nullptr, nullptr, nullptr, 0, 0, 1};
    if (wgrad_dma_ok(xl, lddy, M, N, K, 0)) {
        const int rc = launch_wgrad_dma(dy, lddy, xl, dW, (long)K, dbias, M, N, K, stream, 0);
        if (rc != LEOD_ERR_UNSUPPORTED) return rc;
    }
    if (use_wgrad_wide(xl, lddy, M, N, K, 0)) return launch_wgrad_wide(dy, lddy, xl, dW, (long)K, dbias, M, N, K, stream, 0);
    if (use_wgradw(M)) return launch_wgradw(dy, lddy, xl, dW, (long)K, dbias, M, N, K, stream);
    if (N % 48 == 0 && K % 48 == 0) return launch_wgrad16<3, 3>(dy, lddy, xl, dW, (long)K, dbias, M, N, K, stream);
    return launch_wgrad16<4, 4>(dy, lddy, xl, dW, (long)K, dbias, M, N, K, stream);
}
