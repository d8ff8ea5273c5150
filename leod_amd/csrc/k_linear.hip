// Token-row contractions of the RVT backbone: LayerNorm->Linear(+GELU), Linear->LayerScale+residual,
// the fused ConvLSTM cell, and the matching dgrad / wgrad GEMMs.  All tensors fp32, channels-last
// ("rows" = tokens of an NHWC map).  C-ABI declared in include/leod_hip.h.
#include "gemm16.hpp"

static inline int pick_nt(int N) {
    int best = 1; long bestpad = 1L << 60;
    for (int nt = 4; nt >= 1; --nt) {
        const long pad = (long)cdiv(N, 16 * nt) * 16 * nt;
        if (pad < bestpad) { bestpad = pad; best = nt; }
    }
    return best;
}

#define DISPATCH_NT(NTV, ...)                                          \
    switch (NTV) {                                                     \
        case 1: { constexpr int NT = 1; __VA_ARGS__; } break;          \
        case 2: { constexpr int NT = 2; __VA_ARGS__; } break;          \
        case 3: { constexpr int NT = 3; __VA_ARGS__; } break;          \
        default: { constexpr int NT = 4; __VA_ARGS__; } break;         \
    }

static inline EpStore ep_store(float* out, long ld, int N) {
    EpStore e{};
    e.out = out; e.ld = ld; e.N = N; e.act = ACT_NONE;
    return e;
}

// out[M,N] = LN(x)[M,K] @ W[N,K]^T + bias ; optionally also out_act = gelu(out)
// ln_w == NULL -> no LayerNorm.  stats_out (optional) [M,2] = (mean, rstd) for the backward pass.
// Reference: models/layers/maxvit/maxvit.py:267-269 (norm1 -> qkv, :347) and :110-118 (norm2 -> fc1 -> GELU)
LEOD_API int leod_ln_linear_fwd(const float* x, long ldx, const float* ln_w, const float* ln_b, float eps,
                                const float* W, const float* bias, float* out, float* out_act, float* stats_out,
                                int M, int N, int K, hipStream_t stream) {
    if (!x || !W || !out || (K & 3) || (ldx & 3)) return LEOD_ERR_ARG;
    ALRows al{}; al.x = x; al.ld = ldx; al.ln_w = ln_w; al.ln_b = ln_b; al.eps = eps; al.stats_out = stats_out; al.K = K;
    EpStore ep = ep_store(out, N, N);
    ep.bias = bias;
    if (out_act) { ep.act = ACT_GELU_DUAL; ep.out2 = out_act; ep.ld2 = N; }
    const int nt = pick_nt(N);
    int rc = LEOD_OK;
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
    if (!a || !W || !res || !out || (K & 3)) return LEOD_ERR_ARG;
    ALRows al{}; al.x = a; al.ld = K; al.K = K;
    EpLsRes ep{out, tout, res, bias, gamma, (long)N, N};
    const int nt = pick_nt(N);
    int rc = LEOD_OK;
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
    if (!x || !W || !bias || !h_out || !c_out || (C & 15)) return LEOD_ERR_ARG;
    ALConcat2 al{x, (long)C, C, h_prev, (long)C};
    BLGates bl{W, (long)2 * C, C};
    EpLstm ep{bias, c_prev, h_out, c_out, gates_out, C};
    // a zero initial state contributes nothing: stop the contraction at K = C
    const int K = h_prev ? 2 * C : C;
    if (use_gemm_lds(M, C / 16)) return launch_gemm_lds<4>(al, bl, ep, M, K, C / 16, stream);
    return launch_gemm16<4>(al, bl, ep, M, K, C / 16, stream);
}

// dx[M,K] (=|+=) (dy[M,N] * kscale[N]) @ W[N,K]          (dgrad of y = x W^T)
//   aux_u != NULL : dx *= gelu'(aux_u[M,K])                (through GELU, maxvit.py:107)
//   nsplit > 0    : columns >= nsplit go to dx2[M, K-nsplit] (ConvLSTM: [dx | dh_prev])
//   colsum != NULL: colsum[K] += column sums of the stored dx (bias gradient of the producer)
LEOD_API int leod_linear_dgrad(const float* dy, long lddy, const float* kscale, const float* W, float* dx, long lddx,
                               float* dx2, long lddx2, int nsplit, const float* aux_u, float* colsum,
                               int accumulate, int M, int N, int K, hipStream_t stream) {
    if (!dy || !W || !dx || (N & 3) || (lddy & 3)) return LEOD_ERR_ARG;
    ALRows al{}; al.x = dy; al.ld = lddy; al.kscale = kscale; al.K = N;
    EpStore ep = ep_store(dx, lddx, K);
    ep.out2 = dx2; ep.ld2 = lddx2; ep.nsplit = nsplit; ep.accumulate = accumulate; ep.colsum = colsum;
    if (aux_u) { ep.act = ACT_MUL_GELU_GRAD; ep.aux = aux_u; ep.ldaux = K; }
    const int nt = pick_nt(K);
    int rc = LEOD_OK;
    if (use_gemm_lds(M, cdiv(K, 16 * nt))) {
        DISPATCH_NT(nt, { BLTrans bl{W, (long)K, K, NT}; rc = launch_gemm_lds<NT>(al, bl, ep, M, N, cdiv(K, 16 * NT), stream); });
        return rc;
    }
    DISPATCH_NT(nt, { BLTrans bl{W, (long)K, K, NT}; rc = launch_gemm16<NT>(al, bl, ep, M, N, cdiv(K, 16 * NT), stream); });
    return rc;
}

// dW[N,K] += dy[M,N]^T @ X[M,K] ; dbias[N] += colsum(dy)  with X = x, LN(x) (stats + ln_w/ln_b) or [x | x2]
LEOD_API int leod_linear_wgrad(const float* dy, long lddy, const float* x, long ldx, const float* stats,
                               const float* ln_w, const float* ln_b, const float* x2, long ldx2, int K1,
                               float* dW, float* dbias, int M, int N, int K, hipStream_t stream) {
    if (!dy || !x || !dW) return LEOD_ERR_ARG;
    XRows xl{x, ldx, stats, ln_w, ln_b, x2, ldx2, K1};
    if (use_wgradw(M)) return launch_wgradw(dy, lddy, xl, dW, (long)K, dbias, M, N, K, stream);
    if (N % 48 == 0 && K % 48 == 0) return launch_wgrad16<3, 3>(dy, lddy, xl, dW, (long)K, dbias, M, N, K, stream);
    if (N % 32 == 0 && K % 32 == 0 && (N % 64 || K % 64)) return launch_wgrad16<2, 2>(dy, lddy, xl, dW, (long)K, dbias, M, N, K, stream);
    if (N >= 64 && K >= 64) return launch_wgrad16<4, 4>(dy, lddy, xl, dW, (long)K, dbias, M, N, K, stream);
    if (K >= 64) return launch_wgrad16<1, 4>(dy, lddy, xl, dW, (long)K, dbias, M, N, K, stream);
    if (N >= 64) return launch_wgrad16<4, 1>(dy, lddy, xl, dW, (long)K, dbias, M, N, K, stream);
    return launch_wgrad16<1, 1>(dy, lddy, xl, dW, (long)K, dbias, M, N, K, stream);
}
