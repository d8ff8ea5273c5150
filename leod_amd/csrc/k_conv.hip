// Implicit-GEMM convolutions of the LEOD path on channels-last fp32 maps (no im2col buffer):
//   * stem: 7x7 stride-4 conv straight from the raw NCHW uint8/fp32 event tensor, the bottom/right
//     zero padding of utils/padding.py:32-58 folded into the bounds predicate
//     (models/layers/maxvit/maxvit.py:160-171, stage 1),
//   * 3x3 stride-2 / stride-1 and 1x1 convs (backbone downsampling, PAFPN, YOLOX head:
//     yolo_pafpn.py:109-140, network_blocks.py:29-51, yolo_head.py:208-222),
//   * their dgrad / wgrad.
// Epilogues: raw store (+bias), raw store + per-channel (sum, sumsq) for training BatchNorm,
// or folded eval-BatchNorm + SiLU.
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

static EpStore conv_epilogue(float* out, int N, const float* bias, double* colstats, const float* bn_w, const float* bn_b,
                             const float* bn_rm, const float* bn_rv, float bn_eps) {
    EpStore ep{};
    ep.out = out; ep.ld = N; ep.N = N; ep.bias = bias; ep.colstats = colstats; ep.act = ACT_NONE;
    if (bn_w) { ep.act = ACT_AFFINE_SILU; ep.bn_w = bn_w; ep.bn_b = bn_b; ep.bn_rm = bn_rm; ep.bn_rv = bn_rv; ep.bn_eps = bn_eps; }
    return ep;
}

// y[B,Ho,Wo,N] = conv(x[B,H,W,Cin] NHWC, w[N,Cin,ks,ks]) (+bias) ; Ho = (H + 2*pad - ks)/stride + 1
//   colstats != NULL : also accumulate per-channel (sum, sumsq) in double for training BatchNorm
//   bn_w != NULL     : eval mode, y = silu(bn(conv)) with running statistics folded in
LEOD_API int leod_conv_nhwc_fwd(const float* x, const float* w, const float* bias, float* y, double* colstats,
                                const float* bn_w, const float* bn_b, const float* bn_rm, const float* bn_rv, float bn_eps,
                                int B, int H, int W, int Cin, int N, int ks, int stride, int pad, hipStream_t stream) {
    if (!x || !w || !y || (Cin & 3)) return LEOD_ERR_ARG;
    const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
    const int M = B * Ho * Wo, K = ks * ks * Cin;
    EpStore ep = conv_epilogue(y, N, bias, colstats, bn_w, bn_b, bn_rm, bn_rv, bn_eps);
    const int nt = pick_nt(N);
    int rc = LEOD_OK;
    const bool lds = use_gemm_lds(M, cdiv(N, 16 * nt));      // large M: coalesced LDS-staged operands
    if (ks == 1 && stride == 1 && pad == 0) {
        ALRows al{}; al.x = x; al.ld = Cin; al.K = Cin;
        DISPATCH_NT(nt, { BLRows bl{w, (long)Cin, N, NT};
                          rc = lds ? launch_gemm_lds<NT>(al, bl, ep, M, K, cdiv(N, 16 * NT), stream)
                                   : launch_gemm16<NT>(al, bl, ep, M, K, cdiv(N, 16 * NT), stream); });
    } else {
        ALConvNHWC al{x, H, W, Cin, Ho, Wo, ks, stride, pad};
        DISPATCH_NT(nt, { BLConvW bl{w, N, Cin, ks * ks, NT};
                          rc = lds ? launch_gemm_lds<NT>(al, bl, ep, M, K, cdiv(N, 16 * NT), stream)
                                   : launch_gemm16<NT>(al, bl, ep, M, K, cdiv(N, 16 * NT), stream); });
    }
    return rc;
}

// Stem: y[B,Ho,Wo,N] = conv(pad(x[B,Cin,H,W] NCHW) , w[N,Cin,ks,ks]) with Hp,Wp the padded size
// (Ho = (Hp + 2*pad - ks)/stride + 1).  x_is_u8: raw uint8 stacked-histogram voxels.
LEOD_API int leod_stem_conv_fwd(const void* x, int x_is_u8, const float* w, float* y, int B, int Cin, int H, int W,
                                int Hp, int Wp, int N, int ks, int stride, int pad, hipStream_t stream) {
    if (!x || !w || !y || ((Cin * ks * ks) & 3)) return LEOD_ERR_ARG;
    if (ks != 7) return LEOD_ERR_UNSUPPORTED;       // stem loaders hard-code the 7x7 tap decode
    const int Ho = (Hp + 2 * pad - ks) / stride + 1, Wo = (Wp + 2 * pad - ks) / stride + 1;
    const int M = B * Ho * Wo, K = Cin * ks * ks;
    EpStore ep = conv_epilogue(y, N, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f);
    const int nt = pick_nt(N);
    int rc = LEOD_OK;
    const bool lds = use_gemm_lds(M, cdiv(N, 16 * nt));
    if (x_is_u8) {
        ALStemNCHW<uint8_t> al{(const uint8_t*)x, Cin, H, W, Ho, Wo, ks, stride, pad};
        DISPATCH_NT(nt, { BLRows bl{w, (long)K, N, NT};
                          rc = lds ? launch_gemm_lds<NT>(al, bl, ep, M, K, cdiv(N, 16 * NT), stream)
                                   : launch_gemm16<NT>(al, bl, ep, M, K, cdiv(N, 16 * NT), stream); });
    } else {
        ALStemNCHW<float> al{(const float*)x, Cin, H, W, Ho, Wo, ks, stride, pad};
        DISPATCH_NT(nt, { BLRows bl{w, (long)K, N, NT};
                          rc = lds ? launch_gemm_lds<NT>(al, bl, ep, M, K, cdiv(N, 16 * NT), stream)
                                   : launch_gemm16<NT>(al, bl, ep, M, K, cdiv(N, 16 * NT), stream); });
    }
    return rc;
}

// dx[B,H,W,Cin] (=|+=) conv_transpose(dy[B,Ho,Wo,N], w)
LEOD_API int leod_conv_nhwc_dgrad(const float* dy, const float* w, float* dx, int accumulate, int B, int H, int W, int Cin,
                                  int N, int ks, int stride, int pad, hipStream_t stream) {
    if (!dy || !w || !dx || (N & 3)) return LEOD_ERR_ARG;
    const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
    const int M = B * H * W, K = ks * ks * N;
    EpStore ep{}; ep.out = dx; ep.ld = Cin; ep.N = Cin; ep.accumulate = accumulate;
    const int nt = pick_nt(Cin);
    int rc = LEOD_OK;
    const int Q = B * (H / 2) * (W / 2);
    if (ks == 3 && stride == 2 && pad == 1 && !(H & 1) && !(W & 1) && Q % 16 == 0) {
        // live-tap formulation: rows grouped by input parity class, 2.25 taps per pixel on average instead of 9
        ALConvT2 al{dy, H, W, Ho, Wo, N, Q};
        ep.rm_Q = Q; ep.rm_H = H; ep.rm_W = W;
        const bool lds2 = use_gemm_lds(M, cdiv(Cin, 16 * nt)) && Q % 128 == 0;    // a (64|128)-row workgroup must not mix classes
        DISPATCH_NT(nt, { BLConvWT2 bl{w, N, Cin, NT};
                          rc = lds2 ? launch_gemm_lds<NT>(al, bl, ep, M, 4 * N, cdiv(Cin, 16 * NT), stream)
                                    : launch_gemm16<NT>(al, bl, ep, M, 4 * N, cdiv(Cin, 16 * NT), stream); });
        return rc;
    }
    const bool lds = use_gemm_lds(M, cdiv(Cin, 16 * nt));
    if (ks == 1 && stride == 1 && pad == 0) {
        ALRows al{}; al.x = dy; al.ld = N; al.K = N;
        DISPATCH_NT(nt, { BLTrans bl{w, (long)Cin, Cin, NT};
                          rc = lds ? launch_gemm_lds<NT>(al, bl, ep, M, K, cdiv(Cin, 16 * NT), stream)
                                   : launch_gemm16<NT>(al, bl, ep, M, K, cdiv(Cin, 16 * NT), stream); });
    } else {
        ALConvT al{dy, H, W, Ho, Wo, N, ks, stride, pad};
        DISPATCH_NT(nt, { BLConvWT bl{w, N, Cin, ks * ks, NT};
                          rc = lds ? launch_gemm_lds<NT>(al, bl, ep, M, K, cdiv(Cin, 16 * NT), stream)
                                   : launch_gemm16<NT>(al, bl, ep, M, K, cdiv(Cin, 16 * NT), stream); });
    }
    return rc;
}

template <class XL>
static int wgrad_any(const float* dy, const XL& xl, float* dW, long ldw, float* dbias, int M, int N, int K, hipStream_t s) {
    static const int conv_w = getenv("LEOD_WGRADW_CONV") ? atoi(getenv("LEOD_WGRADW_CONV")) : 0;
    if (conv_w && use_wgradw(M)) return launch_wgradw(dy, (long)N, xl, dW, ldw, dbias, M, N, K, s);
    if (N % 48 == 0) return launch_wgrad16<3, 4>(dy, (long)N, xl, dW, ldw, dbias, M, N, K, s);
    if (N % 64 == 0) return launch_wgrad16<4, 4>(dy, (long)N, xl, dW, ldw, dbias, M, N, K, s);
    if (N % 32 == 0) return launch_wgrad16<2, 4>(dy, (long)N, xl, dW, ldw, dbias, M, N, K, s);
    return launch_wgrad16<1, 4>(dy, (long)N, xl, dW, ldw, dbias, M, N, K, s);
}

// dw[N,Cin,ks,ks] += dy^T im2col(x) ; dbias[N] += colsum(dy)
LEOD_API int leod_conv_nhwc_wgrad(const float* dy, const float* x, float* dw, float* dbias, int B, int H, int W, int Cin,
                                  int N, int ks, int stride, int pad, hipStream_t stream) {
    if (!dy || !x || !dw) return LEOD_ERR_ARG;
    const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
    const int M = B * Ho * Wo, K = ks * ks * Cin;
    if (ks == 1 && stride == 1 && pad == 0) {
        XRows xl{x, (long)Cin, nullptr, nullptr, nullptr, nullptr, 0, 0};
        return wgrad_any(dy, xl, dw, (long)Cin, dbias, M, N, K, stream);
    }
    XConvNHWC xl{x, H, W, Cin, Ho, Wo, ks, stride, pad};
    return wgrad_any(dy, xl, dw, (long)K, dbias, M, N, K, stream);
}

LEOD_API int leod_stem_conv_wgrad(const float* dy, const void* x, int x_is_u8, float* dw, int B, int Cin, int H, int W,
                                  int Hp, int Wp, int N, int ks, int stride, int pad, hipStream_t stream) {
    if (!dy || !x || !dw) return LEOD_ERR_ARG;
    if (ks != 7) return LEOD_ERR_UNSUPPORTED;
    const int Ho = (Hp + 2 * pad - ks) / stride + 1, Wo = (Wp + 2 * pad - ks) / stride + 1;
    const int M = B * Ho * Wo, K = Cin * ks * ks;
    if (x_is_u8) {
        XStemNCHW<uint8_t> xl{(const uint8_t*)x, Cin, H, W, Ho, Wo, ks, stride, pad};
        return wgrad_any(dy, xl, dw, (long)K, nullptr, M, N, K, stream);
    }
    XStemNCHW<float> xl{(const float*)x, Cin, H, W, Ho, Wo, ks, stride, pad};
    return wgrad_any(dy, xl, dw, (long)K, nullptr, M, N, K, stream);
}
