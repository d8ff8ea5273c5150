// Token-row contractions of the RVT backbone, forward entry points of the 16-bit precision modes whose inputs / outputs are 16-bit rows (fp16 MLP hidden, bf16 / fp16 qkv and attention output).  All tensors channels-last ("rows" = tokens of an NHWC map).
// C-ABI declared in include/leod_hip.h.
#include "linear_common.hpp"

static int ln_linear_16_generic(const float* x, const float* ln_w, const float* ln_b, float eps, const float* W, const float* bias,
                                void* out16, float* stats_out, int M, int N, int K, int out_fmt, hipStream_t stream);

// ---------------------------------------------------------------------------------------------------------------------
// Precision mode bf16, stages 1-2: the MLP hidden u = LN(x) W1^T + b1 is stored ONCE as fp16 (as the reference does under
// autocast); fc2, the dgrad through GELU and the fc2 weight gradient evaluate GELU / GELU' on load.  8 -> 2 bytes per hidden
// element in the forward pass, 4 -> 2 on each of its three reads.
// ---------------------------------------------------------------------------------------------------------------------
// u16[M,N] = fp16(LN(x) W^T + bias); stats_out [M,2].  LEOD_ERR_UNSUPPORTED unless the row-streaming kernel covers (M, N, K)
// in precision mode bf16 -- the caller then uses leod_ln_linear_fwd with its fp32 (u, gelu(u)) pair.
LEOD_API int leod_ln_linear_gelu16_fwd(const float* x, const float* ln_w, const float* ln_b, float eps, const float* W, const float* bias,
                                       void* u16, float* stats_out, int M, int N, int K, hipStream_t stream) {
    LeodFwdScope fwd_scope;                                   // forward contraction: fp16 operands in precision mode 16f
    if (!x || !W || !u16 || !ln_w || !stats_out) return LEOD_ERR_ARG;
    static const int on = 1;
    const int slab = rowstream_slab(M, N, K);
    static const int gen16 = 1;
    if (!on || leod_precision() != 1) return LEOD_ERR_UNSUPPORTED;
    if (!slab) return gen16 ? ln_linear_16_generic(x, ln_w, ln_b, eps, W, bias, u16, stats_out, M, N, K, 1, stream) : LEOD_ERR_UNSUPPORTED;
    const int slabs = N / (16 * slab);
#define U16_CASE(KCV, NTTV)                                                                                                          \
    if (K == 16 * KCV && slab == NTTV) {                                                                                             \
        const int per_cu = (KCV == 3 && NTTV <= 9) ? 3 : 2;                                                                          \
        const int gx = min(cdiv(cdiv(M, 16), 4), max(8, (256 * per_cu / slabs) & ~7));                                               \
        LEOD_BY_OPFMT16(hipLaunchKernelGGL((rowstream48_kernel<KCV, NTTV, true, true, 0, OF, true>), dim3(gx, slabs), dim3(256), 0, stream, x, (long)K,   \
                           stats_out, ln_w, ln_b, eps, W, bias, nullptr, reinterpret_cast<float*>(u16), M, N));                      \
        return leod_launch_status();                                                                                                 \
    }
    U16_CASE(3, 12) U16_CASE(6, 8) U16_CASE(4, 8)
#undef U16_CASE
    return LEOD_ERR_UNSUPPORTED;
}

// Generic 16-bit producers (round 3: stages 3-4 and every geometry the row-streaming kernels do not cover): the LDS-staged / wide-tile
// GEMMs with a 16-bit row epilogue.  out_fmt 1 = fp16 (the MLP hidden pre-activation), 2 = bf16 (qkv).  ln_w may be NULL (plain rows).
static int ln_linear_16_generic(const float* x, const float* ln_w, const float* ln_b, float eps, const float* W, const float* bias,
                                void* out16, float* stats_out, int M, int N, int K, int out_fmt, hipStream_t stream) {
    if (leod_precision() != 1 || (K & 3) || (N & 3) || (ln_w && !stats_out)) return LEOD_ERR_UNSUPPORTED;
    const int nt = pick_nt(N);
    if (!use_gemm_lds(M, cdiv(N, 16 * nt))) return LEOD_ERR_UNSUPPORTED;
    ALRows al{}; al.x = x; al.ld = K; al.ln_w = ln_w; al.ln_b = ln_b; al.eps = eps; al.K = K;
    EpStore ep = ep_store(reinterpret_cast<float*>(out16), N, N);
    ep.bias = bias; ep.out_fmt = out_fmt;
    int rc = LEOD_OK;
    if (ln_w) { rc = launch_row_stats(x, K, stats_out, M, K, eps, stream); if (rc) return rc; al.stats_in = stats_out; }
    DISPATCH_NT(nt, { BLRows bl{W, (long)K, N, NT}; rc = launch_gemm_lds<NT>(al, bl, ep, M, K, cdiv(N, 16 * NT), stream); });
    return rc;
}

// out16[M,N] = bf16(LN(x) W^T + bias) (the qkv rows of stages 1-2: q, k, v only ever enter bf16 MFMAs); stats_out [M,2]
LEOD_API int leod_ln_linear_bf16_fwd(const float* x, const float* ln_w, const float* ln_b, float eps, const float* W, const float* bias,
                                     void* out16, float* stats_out, int M, int N, int K, hipStream_t stream) {
    LeodFwdScope fwd_scope;                                   // forward contraction: fp16 operands in precision mode 16f
    if (!x || !W || !out16 || (ln_w && !stats_out)) return LEOD_ERR_ARG;
    const int slab = ln_w ? rowstream_slab(M, N, K) : 0;
    static const int gen16 = 1;
    if (leod_precision() != 1) return LEOD_ERR_UNSUPPORTED;
    // the stored rows are the attention kernels' MFMA operands: bf16, or fp16 in precision mode 16f
    if (!slab) return gen16 ? ln_linear_16_generic(x, ln_w, ln_b, eps, W, bias, out16, stats_out, M, N, K, leod_opfmt() == 2 ? 1 : 2, stream) : LEOD_ERR_UNSUPPORTED;
    const int slabs = N / (16 * slab);
#define O16_CASE(KCV, NTTV)                                                                                                          \
    if (K == 16 * KCV && slab == NTTV) {                                                                                             \
        const int per_cu = (KCV == 3 && NTTV <= 9) ? 3 : 2;                                                                          \
        const int gx = min(cdiv(cdiv(M, 16), 4), max(8, (256 * per_cu / slabs) & ~7));                                               \
        LEOD_BY_OPFMT16(hipLaunchKernelGGL((rowstream48_kernel<KCV, NTTV, true, false, 0, OF, false, OF>), dim3(gx, slabs), dim3(256), 0, stream, x, (long)K, \
                           stats_out, ln_w, ln_b, eps, W, bias, reinterpret_cast<float*>(out16), nullptr, M, N));                    \
        return leod_launch_status();                                                                                                 \
    }
    O16_CASE(3, 9) O16_CASE(6, 9) O16_CASE(4, 12)
#undef O16_CASE
    return LEOD_ERR_UNSUPPORTED;
}

// out = res + gamma * (a16 W^T + bias) with bf16 rows a16 (proj + LayerScale + residual on the bf16 attention output)
LEOD_API int leod_linear_lsres_bf16_fwd(const void* a16, const float* W, const float* bias, const float* gamma, const float* res,
                                        float* out, int M, int N, int K, hipStream_t stream) {
    LeodFwdScope fwd_scope;                                   // forward contraction: fp16 operands in precision mode 16f
    if (!a16 || !W || !res || !out || (K & 7) || leod_precision() != 1) return LEOD_ERR_ARG;
    ALRows al{}; al.x = reinterpret_cast<const float*>(a16); al.ld = K; al.K = K; al.fmt = leod_opfmt() == 2 ? 3 : 2;   // bf16 rows | fp16 rows (mode 16f)
    EpLsRes ep{out, nullptr, res, bias, gamma, (long)N, N};
    const int nt = pick_nt(N);
    if (!use_gemm_lds(M, cdiv(N, 16 * nt))) return LEOD_ERR_UNSUPPORTED;
    int rc = LEOD_OK;
    DISPATCH_NT(nt, { BLRows bl{W, (long)K, N, NT}; rc = launch_gemm_lds<NT>(al, bl, ep, M, K, cdiv(N, 16 * NT), stream); });
    return rc;
}

// out = res + gamma * (gelu(u16) W^T + bias)     (fc2 + LayerScale + residual on the fp16 pre-activation)
LEOD_API int leod_linear_lsres_gelu16_fwd(const void* u16, const float* W, const float* bias, const float* gamma, const float* res,
                                          float* out, int M, int N, int K, hipStream_t stream) {
    LeodFwdScope fwd_scope;                                   // forward contraction: fp16 operands in precision mode 16f
    if (!u16 || !W || !res || !out || (K & 3) || leod_precision() != 1) return LEOD_ERR_ARG;
    const float* a = reinterpret_cast<const float*>(u16);
    if (K == 384 && use_rowstream_narrow96(M, K, N)) {
        LEOD_BY_OPFMT16(hipLaunchKernelGGL((rowstream_narrow_kernel<24, 0, OF, 1, 6>), dim3(min(cdiv(cdiv(M, 16), 8), 256)), dim3(512), 0, stream, a, W, bias, gamma, res, out, M));
        return leod_launch_status();
    }
    if (use_rowstream_narrow(M, K, N)) {
        const int grid = min(cdiv(cdiv(M, 16), 4), 256 * 2);
        LEOD_BY_OPFMT16({
            if (K == 192) hipLaunchKernelGGL((rowstream_narrow_kernel<12, 0, OF, 1>), dim3(grid), dim3(256), 0, stream, a, W, bias, gamma, res, out, M);
            else hipLaunchKernelGGL((rowstream_narrow_kernel<9, 0, OF, 1>), dim3(grid), dim3(256), 0, stream, a, W, bias, gamma, res, out, M);
        });
        return leod_launch_status();
    }
    ALRows al{}; al.x = a; al.ld = K; al.K = K; al.fmt = 1;
    EpLsRes ep{out, nullptr, res, bias, gamma, (long)N, N};
    const int nt = pick_nt(N);
    int rc = LEOD_OK;
    if (use_gemm_lds(M, cdiv(N, 16 * nt))) {
        DISPATCH_NT(nt, { BLRows bl{W, (long)K, N, NT}; rc = launch_gemm_lds<NT>(al, bl, ep, M, K, cdiv(N, 16 * NT), stream); });
        return rc;
    }
    DISPATCH_NT(nt, { BLRows bl{W, (long)K, N, NT}; rc = launch_gemm16<NT>(al, bl, ep, M, K, cdiv(N, 16 * NT), stream); });
    return rc;
}
