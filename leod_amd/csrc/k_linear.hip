// Token-row contractions of the RVT backbone, forward with fp32 tensors: LayerNorm->Linear(+GELU), Linear->LayerScale+residual, the fused ConvLSTM cell.  All tensors channels-last ("rows" = tokens of an NHWC map).
// C-ABI declared in include/leod_hip.h.
#include "linear_common.hpp"

// out[M,N] = LN(x)[M,K] @ W[N,K]^T + bias ; optionally also out_act = gelu(out)
// ln_w == NULL -> no LayerNorm.  stats_out (optional) [M,2] = (mean, rstd) for the backward pass.
// Reference: models/layers/maxvit/maxvit.py:267-269 (norm1 -> qkv, :347) and :110-118 (norm2 -> fc1 -> GELU)
LEOD_API int leod_ln_linear_fwd(const float* x, long ldx, const float* ln_w, const float* ln_b, float eps,
                                const float* W, const float* bias, float* out, float* out_act, float* stats_out,
                                int M, int N, int K, hipStream_t stream) {
    LeodFwdScope fwd_scope;                                   // forward contraction: fp16 operands in precision mode 16f
    if (!x || !W || !out || (K & 3) || (ldx & 3)) return LEOD_ERR_ARG;
    ALRows al{}; al.x = x; al.ld = ldx; al.ln_w = ln_w; al.ln_b = ln_b; al.eps = eps; al.stats_out = stats_out; al.K = K;
    EpStore ep = ep_store(out, N, N);
    ep.bias = bias;
    if (out_act) { ep.act = ACT_GELU_DUAL; ep.out2 = out_act; ep.ld2 = N; }
    const int nt = pick_nt(N);
    int rc = LEOD_OK;
    if (const int slab = ((!ln_w || stats_out) && ldx == K) ? rowstream_slab(M, N, K) : 0) {
        float* st = ln_w ? stats_out : nullptr;                 // the kernel derives (mean, rstd) itself and leaves them here
#define RS_CASE(KCV, NTTV)                                                                                                         \
        if (K == 16 * KCV && slab == NTTV)                                                                                         \
            return out_act ? launch_rowstream48<KCV, NTTV, true>(x, ldx, st, ln_w, ln_b, eps, W, bias, out, out_act, M, N, stream)   \
                           : launch_rowstream48<KCV, NTTV, false>(x, ldx, st, ln_w, ln_b, eps, W, bias, out, nullptr, M, N, stream);
        RS_CASE(3, 9) RS_CASE(3, 12) RS_CASE(6, 9) RS_CASE(6, 8) RS_CASE(4, 12) RS_CASE(4, 8)
#undef RS_CASE
    }
    if (use_gemm_lds(M, cdiv(N, 16 * nt)) && (!ln_w || stats_out)) {
        if (ln_w) { rc = launch_row_stats(x, ldx, stats_out, M, K, eps, stream); if (rc) return rc; al.stats_in = stats_out; }
        DISPATCH_NT(nt, { BLRows bl{W, (long)K, N, NT}; rc = launch_gemm_lds<NT>(al, bl, ep, M, K, cdiv(N, 16 * NT), stream); });
        return rc;
    }
    DISPATCH_NT(nt, { BLRows bl{W, (long)K, N, NT}; rc = launch_gemm16<NT>(al, bl, ep, M, K, cdiv(N, 16 * NT), stream); });
    return rc;
}

// t = a @ W^T + bias ; tout = t (optional) ; out = res + gamma * t        (maxvit.py:268-269, LayerScale :51-53)
LEOD_API int leod_linear_lsres_fwd(const float* a, const float* W, const float* bias, const float* gamma,
                                   const float* res, float* out, float* tout, int M, int N, int K, hipStream_t stream) {
    LeodFwdScope fwd_scope;                                   // forward contraction: fp16 operands in precision mode 16f
    if (!a || !W || !res || !out || (K & 3)) return LEOD_ERR_ARG;
    ALRows al{}; al.x = a; al.ld = K; al.K = K;
    EpLsRes ep{out, tout, res, bias, gamma, (long)N, N};
    const int nt = pick_nt(N);
    int rc = LEOD_OK;
    if (!tout && use_rowstream_narrow(M, K, N)) return launch_rowstream_narrow<0>(a, W, bias, gamma, res, out, M, K, stream);
    if (use_gemm_lds(M, cdiv(N, 16 * nt))) {
        DISPATCH_NT(nt, { BLRows bl{W, (long)K, N, NT}; rc = launch_gemm_lds<NT>(al, bl, ep, M, K, cdiv(N, 16 * NT), stream); });
        return rc;
    }
    DISPATCH_NT(nt, { BLRows bl{W, (long)K, N, NT}; rc = launch_gemm16<NT>(al, bl, ep, M, K, cdiv(N, 16 * NT), stream); });
    return rc;
}

// Fused ConvLSTM cell (models/layers/rnn.py:37-70, dws_conv=False): gates = [x | h_prev] @ W[4C,2C]^T + b,
// (f,i,o) = sigmoid, g = tanh, c = f*c_prev + i*g, h = o*tanh(c).  h_prev/c_prev NULL = zero state.
// gates_out (optional) [M,4,C] keeps the post-activation gates for the backward pass.
LEOD_API int leod_convlstm_fwd(const float* x, const float* h_prev, const float* c_prev, const float* W,
                               const float* bias, float* h_out, float* c_out, float* gates_out, int M, int C,
                               hipStream_t stream) {
    LeodFwdScope fwd_scope;                                   // forward contraction: fp16 operands in precision mode 16f
    if (!x || !W || !bias || !h_out || !c_out || (C & 15)) return LEOD_ERR_ARG;
    ALConcat2 al{x, (long)C, C, h_prev, (long)C};
    BLGates bl{W, (long)2 * C, C};
    EpLstm ep{bias, c_prev, h_out, c_out, gates_out, C};
    // a zero initial state contributes nothing: stop the contraction at K = C
    const int K = h_prev ? 2 * C : C;
    if (use_gemm_lds(M, C / 16)) return launch_gemm_lds<4>(al, bl, ep, M, K, C / 16, stream);
    return launch_gemm16<4>(al, bl, ep, M, K, C / 16, stream);
}
