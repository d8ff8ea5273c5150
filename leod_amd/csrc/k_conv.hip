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
#include "conv3.hpp"
#include "stem.hpp"

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

static EpStore conv_epilogue(float* out, int N, const float* bias, double* colstats, int stat_rep, const float* bn_w, const float* bn_b,
                             const float* bn_rm, const float* bn_rv, float bn_eps) {
    EpStore ep{};
    ep.out = out; ep.ld = N; ep.N = N; ep.bias = bias; ep.colstats = colstats; ep.stat_rep = stat_rep; ep.act = ACT_NONE;
    if (bn_w) { ep.act = ACT_AFFINE_SILU; ep.bn_w = bn_w; ep.bn_b = bn_b; ep.bn_rm = bn_rm; ep.bn_rv = bn_rv; ep.bn_eps = bn_eps; }
    return ep;
}

// K-contiguous copies of a conv weight w[N][Cin][KK] for the implicit GEMMs (the native layout makes every weight
// float4 four 36-byte-strided scalar loads):  mode 0: wf[n][tap*Cin + c]  (forward, B rows = output channels)
//                                              mode 1: wd[c][tap*N + n]    (dgrad,   B rows = input channels)
__global__ __launch_bounds__(256) void conv_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int N, int Cin,
                                                        int KK, int mode) {
    const long total = (long)N * Cin * KK;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int tap = (int)(e % KK); const long r = e / KK; const int c = (int)(r % Cin), n = (int)(r / Cin);
        const long o = mode == 0 ? ((long)n * KK + tap) * Cin + c : ((long)c * KK + tap) * N + n;
        out[o] = w[e];
    }
}
static inline void launch_conv_pack(const float* w, float* out, int N, int Cin, int KK, int mode, hipStream_t s) {
    const long total = (long)N * Cin * KK;
    hipLaunchKernelGGL(conv_pack_kernel, dim3((int)min((long)1024, (total + 255) / 256)), dim3(256), 0, s, w, out, N, Cin, KK, mode);
}

// y[B,Ho,Wo,N] = conv(x[B,H,W,Cin] NHWC, w[N,Cin,ks,ks]) (+bias) ; Ho = (H + 2*pad - ks)/stride + 1
//   colstats != NULL : also accumulate per-channel (sum, sumsq) in double for training BatchNorm, spread over stat_rep
//                      replicas [stat_rep][2][N] (power of two, zero-initialised by the caller; leod_bn_silu_fwd folds them)
//   bn_w != NULL     : eval mode, y = silu(bn(conv)) with running statistics folded in
LEOD_API int leod_conv_nhwc_fwd(const float* x, const float* w, const float* bias, float* y, double* colstats, int stat_rep,
                                const float* bn_w, const float* bn_b, const float* bn_rm, const float* bn_rv, float bn_eps,
                                int B, int H, int W, int Cin, int N, int ks, int stride, int pad, float* wpack, int wpack_valid,
                                hipStream_t stream) {
    LeodFwdScope fwd_scope;                                   // forward contraction: fp16 operands in precision mode 16f
    if (!x || !w || !y || (Cin & 3) || (stat_rep > 1 && (stat_rep & (stat_rep - 1)))) return LEOD_ERR_ARG;
    const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
    const int M = B * Ho * Wo, K = ks * ks * Cin;
    // PAFPN / head 3x3 convs in precision mode bf16: direct convolution from an LDS-resident input halo (k_conv3.hip)
    // (eval mode, bn_w != NULL: the folded BatchNorm + SiLU run in the direct kernel's row epilogue -- the pseudo-label pass spent a third
    // of its device time in the implicit-GEMM form of these convs)
    if (ks == 3 && stride == 1 && pad == 1 && !bias && !(bn_w && colstats) && wpack && conv3s1_supported(H, W, Cin, N))
        return conv3s1_launch(x, w, y, colstats, stat_rep, 0, B, H, W, Cin, N, 0, wpack, stream, 1, wpack_valid, bn_w, bn_b, bn_rm, bn_rv, bn_eps);
    if (ks == 3 && stride == 2 && pad == 1 && !bias && !(bn_w && colstats) && wpack && conv3s2_fwd_supported(B, H, W, Cin, N))
        return conv3s1_launch(x, w, y, colstats, stat_rep, 0, B, H, W, Cin, N, 0, wpack, stream, 2, wpack_valid, bn_w, bn_b, bn_rm, bn_rv, bn_eps);
    EpStore ep = conv_epilogue(y, N, bias, colstats, stat_rep, bn_w, bn_b, bn_rm, bn_rv, bn_eps);
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
        if (wpack && lds) {
            // scratch given: repack the weights K-contiguous first (N*Cin*ks*ks floats, a few microseconds), then the B
            // operand is a plain row-major matrix like a Linear weight
            if (!wpack_valid) launch_conv_pack(w, wpack, N, Cin, ks * ks, 0, stream);
            DISPATCH_NT(nt, { BLRows bl{wpack, (long)K, N, NT}; rc = launch_gemm_lds<NT>(al, bl, ep, M, K, cdiv(N, 16 * NT), stream); });
            return rc;
        }
        DISPATCH_NT(nt, { BLConvW bl{w, N, Cin, ks * ks, NT};
                          rc = lds ? launch_gemm_lds<NT>(al, bl, ep, M, K, cdiv(N, 16 * NT), stream)
                                   : launch_gemm16<NT>(al, bl, ep, M, K, cdiv(N, 16 * NT), stream); });
    }
    return rc;
}

// =====================================================================================================================
// Dedicated stem forward for the raw uint8 voxels (7x7, stride 4, pad 3 -- the only stem RVT has).
// The generic implicit GEMM gathers every A element with its own byte load (the 7x7/s4 windows overlap 3x, so each input
// byte is fetched ~3 times in 1-byte pieces: TA-bound, 42 TFLOP/s).  Here one workgroup owns a 4 x 16 block of output
// pixels; the 19 x 72-byte x Cin input patch it needs is copied ONCE into LDS with aligned dword loads (zero-filled
// outside the stored H x W frame = the bottom/right padding of utils/padding.py), and the im2col operand fragments are
// assembled from LDS bytes through a per-workgroup k -> byte-offset table.  Weights stream through LDS in 64-wide K chunks
// exactly as in gemm_lds_kernel; outputs leave through the row-layout float4 epilogue.
// =====================================================================================================================
template <int NT, int BF = 0>
__global__ __launch_bounds__(256, 3) void stem_u8_fwd_kernel(const uint8_t* __restrict__ x, const float* __restrict__ w,
                                                             float* __restrict__ y, int Cin, int H, int W, int Ho, int Wo,
                                                             int N, int tiles_x, int tiles_y) {
    constexpr int PR = 19, PC = 72, PD = PC / 4;              // patch rows, bytes per row, dwords per row
    constexpr int KCH = 64, LD = KCH + 8, BN = NT * 16, K4 = KCH / 4;
    constexpr int RB = (BN * K4 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int K = Cin * 49, KP = (K + 15) & ~15;
    const int patch_bytes = Cin * PR * PC;                    // + one zero dword for the k >= K tail
    unsigned char* patch = smem_raw;
    int* toff = reinterpret_cast<int*>(smem_raw + ((patch_bytes + 4 + 15) & ~15));
    float* sB = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(toff) + ((KP * 4 + 15) & ~15));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; const int b = bid / tiles_y;
    const int oy0 = 4 * ty, ox0 = 16 * tx;
    // ---- k -> byte offset inside the patch (relative to the pixel's window origin) ------------------------------------------
    for (int k = tid; k < KP; k += 256) {
        const int c = k / 49, r = k - c * 49, kh = r / 7, kw = r - kh * 7;
        toff[k] = k < K ? c * (PR * PC) + kh * PC + kw + 1 : 0;                // tail: any valid byte (its weights are 0)
    }
    // ---- input patch: rows 4*oy0-3 .. +18, byte columns 4*ox0-4 .. +71, all channels ------------------------------------------
    {
        const int iy0 = 4 * oy0 - 3, ix0 = 4 * ox0 - 4;
        const uint32_t* xb = reinterpret_cast<const uint32_t*>(x + (long)b * Cin * H * W);
        uint32_t* pd = reinterpret_cast<uint32_t*>(patch);
        const int total = Cin * PR * PD;
        // nine loads in flight per thread before the first LDS store (a load -> store loop pays one HBM round trip per iteration:
        // 27 of them for the 20 x 19 x 18 dwords of a patch)
        constexpr int U = 9;
        for (int e0 = tid; e0 < total; e0 += 256 * U) {
            uint32_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = min(e0 + 256 * u, total - 1);
                const int c = e / (PR * PD), rem = e - c * (PR * PD), r = rem / PD, dw = rem - r * PD;
                const int iy = iy0 + r, ix = ix0 + 4 * dw;
                const bool in = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const uint32_t w = xb[in ? (((long)c * H + iy) * W + ix) >> 2 : 0];
                v[u] = in ? w : 0u;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (e0 + 256 * u < total) pd[e0 + 256 * u] = v[u];
        }
        if (tid == 0) pd[total] = 0;
    }
    // ---- weight chunk staging (as gemm_lds_kernel, non-transposed B) ------------------------------------------------------------
    int bn[RB], bk[RB], boffs[RB]; bool bok[RB];
#pragma unroll
    for (int p = 0; p < RB; ++p) {
        const int e = tid + 256 * p, nl = e / K4, k4 = (e - nl * K4) * 4;
        bok[p] = e < BN * K4 && nl < N; bn[p] = nl; bk[p] = k4; boffs[p] = nl * LD + k4;
    }
    f4 rb[RB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            const int k = k0 + bk[p];
            rb[p] = (bok[p] && k < K) ? ld4(w + (long)bn[p] * K + k) : zero4();      // K % 4 == 0
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int p = 0; p < RB; ++p) if (tid + 256 * p < BN * K4) *reinterpret_cast<f4*>(sB + boffs[p]) = rb[p];
    };
    f4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = zero4();
    fetch(0);
    stash();
    __syncthreads();
    const unsigned char* pix = patch + (4 * wave) * PC + 4 * i;        // window origin of pixel (oy0 + wave, ox0 + i)
    const float* pb = sB + i * LD + 4 * q;
    const int nch = (K + KCH - 1) / KCH;
    for (int ch = 0; ch < nch; ++ch) {
        const bool more = ch + 1 < nch;
        if (more) fetch((ch + 1) * KCH);
#pragma unroll
        for (int c = 0; c < KCH / 16; ++c) {
            const int k = ch * KCH + 16 * c + 4 * q;
            f4 av = zero4();
            if (k < KP) {
                const int4 o = *reinterpret_cast<const int4*>(toff + k);
                av.x = (float)pix[o.x]; av.y = (float)pix[o.y]; av.z = (float)pix[o.z]; av.w = (float)pix[o.w];
            }
            if constexpr (BF) {                               // uint8 counts are exact in bf16; the weights are rounded
                const s4 pa = pack16_raw<BF>(av);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = mfma16_16<BF>(pa, pack16_raw<BF>(*reinterpret_cast<const f4*>(pb + 16 * t * LD + 16 * c)), acc[t]);
            } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f4 bv = *reinterpret_cast<const f4*>(pb + 16 * t * LD + 16 * c);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t] = mfma16(av[j], bv[j], acc[t]);
            }
            }
        }
        if (more) {
            __syncthreads();
            stash();
            __syncthreads();
        }
    }
    // ---- epilogue: 16 pixels x BN channels per wave through a 256-byte-row LDS tile, float4 row stores --------------------
    __syncthreads();
    float* so = reinterpret_cast<float*>(smem_raw) + wave * 16 * 64;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) so[(4 * q + r) * 64 + 16 * t + i] = acc[t][r];
    __syncthreads();
    const int oy = oy0 + wave;
    if (oy < Ho) {
        const int c4 = lane & 15, qq = lane >> 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int px = qq + 4 * p, ox = ox0 + px;
            if (ox < Wo && 4 * c4 < N)
                *reinterpret_cast<f4*>(y + (((long)b * Ho + oy) * Wo + ox) * N + 4 * c4) = *reinterpret_cast<const f4*>(so + px * 64 + 4 * c4);
        }
    }
}

template <int NT>
static int launch_stem_u8(const uint8_t* x, const float* w, float* y, int B, int Cin, int H, int W, int Ho, int Wo, int N,
                          hipStream_t s) {
    const int tiles_x = cdiv(Wo, 16), tiles_y = cdiv(Ho, 4);
    const int K = Cin * 49, KP = (K + 15) & ~15;
    const size_t patch = ((size_t)Cin * 19 * 72 + 4 + 15) & ~(size_t)15;
    size_t lds = patch + (((size_t)KP * 4 + 15) & ~(size_t)15) + (size_t)NT * 16 * 72 * 4;
    lds = lds < 4 * 16 * 64 * 4 ? 4 * 16 * 64 * 4 : lds;                 // the epilogue tile aliases the front of the buffer
    LEOD_BY_OPFMT({
        static const bool reg = (leod_register_input_kernel(reinterpret_cast<const void*>(&stem_u8_fwd_kernel<NT, OF>), 0, 11), true);
        (void)reg;
        hipLaunchKernelGGL((stem_u8_fwd_kernel<NT, OF>), dim3(B * tiles_x * tiles_y), dim3(256), lds, s, x, w, y, Cin, H, W, Ho, Wo, N, tiles_x, tiles_y);
    });
    return leod_launch_status();
}

// Stem: y[B,Ho,Wo,N] = conv(pad(x[B,Cin,H,W] NCHW) , w[N,Cin,ks,ks]) with Hp,Wp the padded size
// (Ho = (Hp + 2*pad - ks)/stride + 1).  x_is_u8: raw uint8 stacked-histogram voxels.
LEOD_API int leod_stem_conv_fwd(const void* x, int x_is_u8, const float* w, float* y, int B, int Cin, int H, int W,
                                int Hp, int Wp, int N, int ks, int stride, int pad, hipStream_t stream) {
    LeodFwdScope fwd_scope;                                   // forward contraction: fp16 operands in precision mode 16f
    if (!x || !w || !y || ((Cin * ks * ks) & 3)) return LEOD_ERR_ARG;
    if (ks < 1 || ks > 15) return LEOD_ERR_UNSUPPORTED;
    const int Ho = (Hp + 2 * pad - ks) / stride + 1, Wo = (Wp + 2 * pad - ks) / stride + 1;
    const int M = B * Ho * Wo, K = Cin * ks * ks;
    EpStore ep = conv_epilogue(y, N, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0.f);
    const int nt = pick_nt(N);
    int rc = LEOD_OK;
    const bool lds = use_gemm_lds(M, cdiv(N, 16 * nt));
    static const int stem_patch = 1;
    if (stem_patch && ks == 7 && x_is_u8 && leod_precision() == 1 && stem_fwd_bf16_supported(x, Cin, H, W, N, stride, pad))
        return stem_fwd_bf16_launch(x, w, y, B, Cin, H, W, Ho, Wo, N, stream);         // k_stem.hip: bf16 patch, weights resident in LDS
    if (stem_patch && ks == 7 && x_is_u8 && stride == 4 && pad == 3 && N <= 64 && !(N & 15) && !(W & 3) && Cin * 19 * 72 <= 60000 &&
        ((uintptr_t)x & 3) == 0 && ((long)Cin * H * W) % 4 == 0) {
        // LDS-resident uint8 patch kernel (dedicated to the RVT stem geometry); anything else takes the generic path
        switch (N / 16) {
            case 1: return launch_stem_u8<1>((const uint8_t*)x, w, y, B, Cin, H, W, Ho, Wo, N, stream);
            case 2: return launch_stem_u8<2>((const uint8_t*)x, w, y, B, Cin, H, W, Ho, Wo, N, stream);
            case 3: return launch_stem_u8<3>((const uint8_t*)x, w, y, B, Cin, H, W, Ho, Wo, N, stream);
            default: return launch_stem_u8<4>((const uint8_t*)x, w, y, B, Cin, H, W, Ho, Wo, N, stream);
        }
    }
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
                                  int N, int ks, int stride, int pad, float* wpack, int wpack_valid, hipStream_t stream) {
    if (!dy || !w || !dx || (N & 3)) return LEOD_ERR_ARG;
    const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
    const int M = B * H * W, K = ks * ks * N;
    if (ks == 3 && stride == 1 && pad == 1 && wpack && conv3s1_supported(H, W, N, Cin))
        return conv3s1_launch(dy, w, dx, nullptr, 0, accumulate, B, H, W, N, Cin, 1, wpack, stream, 1, wpack_valid);
    if (ks == 3 && stride == 2 && pad == 1 && wpack && conv3s2_dgrad_supported(H, W, Cin, N))
        return conv3s2_dgrad_launch(dy, w, dx, accumulate, B, H, W, Cin, N, wpack, stream, wpack_valid);
    EpStore ep{}; ep.out = dx; ep.ld = Cin; ep.N = Cin; ep.accumulate = accumulate;
    const int nt = pick_nt(Cin);
    int rc = LEOD_OK;
    const int Q = B * (H / 2) * (W / 2);
    if (ks == 3 && stride == 2 && pad == 1 && !(H & 1) && !(W & 1) && Q % 16 == 0) {
        // live-tap formulation: rows grouped by input parity class, 2.25 taps per pixel on average instead of 9
        ALConvT2 al{dy, H, W, Ho, Wo, N, Q};
        ep.rm_Q = Q; ep.rm_H = H; ep.rm_W = W;
        const bool lds2 = use_gemm_lds(M, cdiv(Cin, 16 * nt)) && Q % 128 == 0;    // a (64|128)-row workgroup must not mix classes
        if (wpack && lds2) {
            if (!wpack_valid) launch_conv_pack(w, wpack, N, Cin, 9, 1, stream);
            DISPATCH_NT(nt, { BLPackT2 bl{wpack, N, Cin, NT}; rc = launch_gemm_lds<NT>(al, bl, ep, M, 4 * N, cdiv(Cin, 16 * NT), stream); });
            return rc;
        }
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
        if (wpack && lds) {
            if (!wpack_valid) launch_conv_pack(w, wpack, N, Cin, ks * ks, 1, stream);         // wd[c][tap*N + n]: B(col = c, k' = tap*N + n)
            DISPATCH_NT(nt, { BLRows bl{wpack, (long)K, Cin, NT}; rc = launch_gemm_lds<NT>(al, bl, ep, M, K, cdiv(Cin, 16 * NT), stream); });
            return rc;
        }
        DISPATCH_NT(nt, { BLConvWT bl{w, N, Cin, ks * ks, NT};
                          rc = lds ? launch_gemm_lds<NT>(al, bl, ep, M, K, cdiv(Cin, 16 * NT), stream)
                                   : launch_gemm16<NT>(al, bl, ep, M, K, cdiv(Cin, 16 * NT), stream); });
    }
    return rc;
}

template <class XL>
static int wgrad_any(const float* dy, const XL& xl, float* dW, long ldw, float* dbias, int M, int N, int K, hipStream_t s) {
    static const int conv_w = 0;
    if (conv_w && use_wgradw(M)) return launch_wgradw(dy, (long)N, xl, dW, ldw, dbias, M, N, K, s);
    if (N % 48 == 0) return launch_wgrad16<3, 4>(dy, (long)N, xl, dW, ldw, dbias, M, N, K, s);
    if (N % 64 == 0) return launch_wgrad16<4, 4>(dy, (long)N, xl, dW, ldw, dbias, M, N, K, s);
    if (N % 32 == 0) return launch_wgrad16<2, 4>(dy, (long)N, xl, dW, ldw, dbias, M, N, K, s);
    return launch_wgrad16<1, 4>(dy, (long)N, xl, dW, ldw, dbias, M, N, K, s);
}

// dw[N,Cin,ks,ks] += dy^T im2col(x) ; dbias[N] += colsum(dy)
// floats of workspace leod_conv_nhwc_wgrad wants for this shape in the current precision mode (0: none)
LEOD_API long leod_conv_nhwc_wgrad_workspace_floats(int B, int H, int W, int Cin, int N, int ks, int stride, int pad, int has_bias) {
    if (ks == 3 && pad == 1 && !has_bias && conv3_wgrad_supported(H, W, Cin, N, stride))
        return (long)conv3_wgrad_workspace_floats(B, H, W, Cin, N, stride);
    return 0;
}

LEOD_API int leod_conv_nhwc_wgrad(const float* dy, const float* x, float* dw, float* dbias, float* ws, int B, int H, int W, int Cin,
                                  int N, int ks, int stride, int pad, hipStream_t stream) {
    if (!dy || !x || !dw) return LEOD_ERR_ARG;
    const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
    const int M = B * Ho * Wo, K = ks * ks * Cin;
    if (ks == 1 && stride == 1 && pad == 0) {
        XRows xl{x, (long)Cin, nullptr, nullptr, nullptr, nullptr, 0, 0};
        // 1 x 1 convs are Linear layers over the pixel rows: the wave-tiled weight gradient of the Linear layers for the large maps in
        // bf16 mode (22 -> 15, 28 -> 23, 18 -> 12 us on the PAFPN shapes; fp32 mode: 24 -> 28, 22 -> 26 us, not used)
        static const int w11 = 1;
        if (w11 && leod_precision() == 1 && use_wgradw(M)) return launch_wgradw(dy, (long)N, xl, dw, (long)Cin, dbias, M, N, K, stream);
        return wgrad_any(dy, xl, dw, (long)Cin, dbias, M, N, K, stream);
    }
    if (ks == 3 && pad == 1 && !dbias && ws && conv3_wgrad_supported(H, W, Cin, N, stride))
        return conv3_wgrad_launch(dy, x, dw, ws, B, H, W, Cin, N, stride, stream);
    XConvNHWC xl{x, H, W, Cin, Ho, Wo, ks, stride, pad};
    return wgrad_any(dy, xl, dw, (long)K, dbias, M, N, K, stream);
}

// =====================================================================================================================
// Dedicated stem weight gradient for uint8 voxels: dW[n][k] += sum_pixels dY[pixel][n] * im2col(x)[pixel][k].
// Same LDS-resident patch as stem_u8_fwd_kernel.  A workgroup walks over 4 x 16-pixel tiles (grid-stride), owns one
// quarter of the 62 k-tiles (blockIdx.y) and keeps its 3 x 4 MFMA tiles per wave in registers over all of its tiles; the
// next tile's patch and dY rows are prefetched into registers while the current one is multiplied.  One fp32 atomic per
// dW element and workgroup at the end.
// =====================================================================================================================
template <int NT, bool BF = false>
__global__ __launch_bounds__(256, 2) void stem_u8_wgrad_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ x,
                                                               float* __restrict__ dW, int B, int Cin, int H, int W, int Ho,
                                                               int Wo, int N, int tiles_x, int tiles_y) {
    constexpr int PR = 19, PC = 72, PD = PC / 4;
    constexpr int BN = NT * 16, LDN = BN + 4, KT = 4;         // k-tiles per wave
    constexpr int RX = 27, RY = (64 * BN / 4 + 255) / 256;    // staging registers: patch dwords, dY float4s per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int K = Cin * 49;
    const int patch_dw = Cin * PR * PD;
    uint32_t* pd = reinterpret_cast<uint32_t*>(smem_raw);
    float* sdy = reinterpret_cast<float*>(smem_raw + (((size_t)patch_dw * 4 + 15) & ~(size_t)15));
    const unsigned char* patch = smem_raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int ntiles = B * tiles_x * tiles_y;
    // this wave's k columns: k-tiles (blockIdx.y * 4 + wave) * KT + b, column i of each
    int toffk[KT]; int kcol[KT];
#pragma unroll
    for (int b = 0; b < KT; ++b) {
        const int k = ((blockIdx.y * 4 + wave) * KT + b) * 16 + i;
        kcol[b] = k;
        const int kk = k < K ? k : 0;
        const int c = kk / 49, r = kk - c * 49, kh = r / 7, kw = r - kh * 7;
        toffk[b] = c * (PR * PC) + kh * PC + kw + 1;
    }
    f4 acc[NT][KT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < KT; ++b) acc[a][b] = zero4();
    uint32_t rx[RX]; f4 ry[RY];
    auto fetch = [&](int tile) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y; const int bb = t / tiles_y;
        const int iy0 = 16 * ty - 3, ix0 = 64 * tx - 4;
        const uint32_t* xb = reinterpret_cast<const uint32_t*>(x + (long)bb * Cin * H * W);
#pragma unroll
        for (int p = 0; p < RX; ++p) {
            const int e = tid + 256 * p;
            uint32_t v = 0;
            if (e < patch_dw) {
                const int c = e / (PR * PD), rem = e - c * (PR * PD), r = rem / PD, dw = rem - r * PD;
                const int iy = iy0 + r, ix = ix0 + 4 * dw;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = xb[(((long)c * H + iy) * W + ix) >> 2];
            }
            rx[p] = v;
        }
#pragma unroll
        for (int p = 0; p < RY; ++p) {
            const int e = tid + 256 * p, px = e / (BN / 4), c4 = e - px * (BN / 4);       // pixel 0..63 = 16*row + col
            const int oy = 4 * ty + (px >> 4), ox = 16 * tx + (px & 15);
            ry[p] = (e < 64 * BN / 4 && oy < Ho && ox < Wo && 4 * c4 < N)
                        ? ld4(dy + (((long)bb * Ho + oy) * Wo + ox) * N + 4 * c4) : zero4();
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int p = 0; p < RX; ++p) { const int e = tid + 256 * p; if (e < patch_dw) pd[e] = rx[p]; }
#pragma unroll
        for (int p = 0; p < RY; ++p) {
            const int e = tid + 256 * p, px = e / (BN / 4), c4 = e - px * (BN / 4);
            if (e < 64 * BN / 4) *reinterpret_cast<f4*>(sdy + px * LDN + 4 * c4) = ry[p];
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) { fetch(tile); stash(); }
    __syncthreads();
    for (; tile < ntiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        if (next < ntiles) fetch(next);
#pragma unroll
        for (int st = 0; st < 4; ++st) {                      // 16 pixels per step: output row st of the tile, ox = 4q + j
            f4 av[NT], bv[KT];
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                const float* t = sdy + (16 * st + 4 * q) * LDN + 16 * a + i;
                av[a].x = t[0]; av[a].y = t[LDN]; av[a].z = t[2 * LDN]; av[a].w = t[3 * LDN];
            }
            const unsigned char* pb = patch + (4 * st) * PC + 16 * q;
#pragma unroll
            for (int b = 0; b < KT; ++b) {
                const unsigned char* t = pb + toffk[b];
                bv[b].x = (float)t[0]; bv[b].y = (float)t[4]; bv[b].z = (float)t[8]; bv[b].w = (float)t[12];
            }
            if constexpr (BF) {
                s4 pa[NT], pk[KT];
#pragma unroll
                for (int a = 0; a < NT; ++a) pa[a] = pack_bf16(av[a]);
#pragma unroll
                for (int b = 0; b < KT; ++b) pk[b] = pack_bf16(bv[b]);
#pragma unroll
                for (int a = 0; a < NT; ++a)
#pragma unroll
                    for (int b = 0; b < KT; ++b) acc[a][b] = mfma16_bf16(pa[a], pk[b], acc[a][b]);
            } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < NT; ++a)
#pragma unroll
                    for (int b = 0; b < KT; ++b) acc[a][b] = mfma16(av[a][j], bv[b][j], acc[a][b]);
            }
        }
        __syncthreads();
        if (next < ntiles) stash();
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < KT; ++b) {
            if (kcol[b] >= K) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 16 * a + 4 * q + r;
                if (n < N) atomicAdd(dW + (long)n * K + kcol[b], acc[a][b][r]);
            }
        }
}

template <int NT>
static int launch_stem_u8_wgrad(const float* dy, const uint8_t* x, float* dW, int B, int Cin, int H, int W, int Ho, int Wo,
                                int N, hipStream_t s) {
    const int tiles_x = cdiv(Wo, 16), tiles_y = cdiv(Ho, 4);
    const int K = Cin * 49, ktiles = cdiv(K, 16);
    const int zs = cdiv(ktiles, 16);                                       // 16 k-tiles per workgroup (4 per wave)
    const size_t lds = (((size_t)Cin * 19 * 72 + 15) & ~(size_t)15) + (size_t)64 * (NT * 16 + 4) * 4;
    const int ntiles = B * tiles_x * tiles_y;
    const int gx = ntiles < 192 ? ntiles : 192;
    static const bool reg = (leod_register_input_kernel(reinterpret_cast<const void*>(&stem_u8_wgrad_kernel<NT, true>), 1, 12),
                             leod_register_input_kernel(reinterpret_cast<const void*>(&stem_u8_wgrad_kernel<NT>), 1, 12), true);
    (void)reg;
    if (leod_precision() == 1)
        hipLaunchKernelGGL((stem_u8_wgrad_kernel<NT, true>), dim3(gx, zs), dim3(256), lds, s, dy, x, dW, B, Cin, H, W, Ho, Wo, N, tiles_x,
                           tiles_y);
    else
        hipLaunchKernelGGL((stem_u8_wgrad_kernel<NT>), dim3(gx, zs), dim3(256), lds, s, dy, x, dW, B, Cin, H, W, Ho, Wo, N, tiles_x,
                           tiles_y);
    return leod_launch_status();
}

// ---- grouped 3x3 / stride-1 / pad-1 convolutions: n <= 8 independent problems of ONE channel geometry in one launch (k_conv3.hip, C3Group) ----
// The convs of equal depth in the cls / reg towers of the three head levels (yolo_head.py:61-145 of the reference builds them as separate
// modules; yolo_head.py:208-222 runs them level by level).  Arrays are HOST arrays of length n.  -3: not coverable (run the problems singly).
// 1: leod_conv3x3_group_fwd covers these n maps for Cin -> Cout in the current precision mode (callers ask BEFORE they hand out weight-pack
// buffers: a refused call must not leave a pack marked valid); the dgrad of the same convs: ask with (Cout, Cin)
LEOD_API int leod_conv3x3_group_supported(int n, const int* H, const int* W, int Cin, int Cout) {
    return (H && W && conv3s1_group_supported(n, H, W, Cin, Cout)) ? 1 : 0;
}
LEOD_API int leod_conv3x3_group_fwd(int n, const float* const* x, const float* const* w, float* const* y, double* const* colstats,
                                    const int* stat_rep, void* const* wpack, const int* wpack_valid, const int* B, const int* H, const int* W,
                                    int Cin, int Cout, hipStream_t stream) {
    if (n < 1 || n > 8 || !x || !w || !y || !colstats || !stat_rep || !wpack || !wpack_valid || !B || !H || !W) return LEOD_ERR_ARG;
    for (int k = 0; k < n; ++k) if (!x[k] || !w[k] || !y[k] || !wpack[k]) return LEOD_ERR_ARG;
    if (!conv3s1_group_supported(n, H, W, Cin, Cout)) return LEOD_ERR_UNSUPPORTED;
    LeodFwdScope fwd_scope;
    int zeros[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    return conv3s1_group(n, x, w, y, colstats, stat_rep, zeros, B, H, W, Cin, Cout, 0, wpack, wpack_valid, stream);
}
// dx_k [B,H,W,Cin] (+)= dgrad of y_k = conv3x3(x_k, w_k[N][Cin][3][3]) from dy_k [B,H,W,N]; accumulate[k] != 0: dx_k += (problems of one
// launch must write DIFFERENT dx buffers)
LEOD_API int leod_conv3x3_group_dgrad(int n, const float* const* dy, const float* const* w, float* const* dx, const int* accumulate,
                                      void* const* wpack, const int* wpack_valid, const int* B, const int* H, const int* W, int Cin, int N,
                                      hipStream_t stream) {
    if (n < 1 || n > 8 || !dy || !w || !dx || !accumulate || !wpack || !wpack_valid || !B || !H || !W) return LEOD_ERR_ARG;
    for (int k = 0; k < n; ++k) {
        if (!dy[k] || !w[k] || !dx[k] || !wpack[k]) return LEOD_ERR_ARG;
        for (int j = 0; j < k; ++j) if (dx[j] == dx[k]) return LEOD_ERR_ARG;
    }
    if (!conv3s1_group_supported(n, H, W, N, Cin)) return LEOD_ERR_UNSUPPORTED;
    double* nostats[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int zeros[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    return conv3s1_group(n, dy, w, dx, nostats, zeros, accumulate, B, H, W, N, Cin, 1, wpack, wpack_valid, stream);
}

// dw_k [N,Cin,3,3] += weight gradient of y_k = conv3x3(x_k [B,H,W,Cin], w_k) from dy_k [B,H,W,N] (stride 1), n <= 8 problems of one geometry in
// one launch + one reduce launch.  ws[k]: leod_conv3x3_group_wgrad_workspace_floats(B[k], H[k], W[k], Cin, N) floats (0: not coverable)
LEOD_API long leod_conv3x3_group_wgrad_workspace_floats(int B, int H, int W, int Cin, int N) {
    return conv3_wgrad_supported(H, W, Cin, N, 1) ? (long)conv3_wgrad_group_workspace_floats(B, H, W, Cin, N, 1) : 0;
}
LEOD_API int leod_conv3x3_group_wgrad(int n, const float* const* dy, const float* const* x, float* const* dw, float* const* ws, const int* B,
                                      const int* H, const int* W, int Cin, int N, hipStream_t stream) {
    if (n < 1 || n > 8 || !dy || !x || !dw || !ws || !B || !H || !W) return LEOD_ERR_ARG;
    for (int k = 0; k < n; ++k) {
        if (!dy[k] || !x[k] || !dw[k] || !ws[k]) return LEOD_ERR_ARG;
        if (!conv3_wgrad_supported(H[k], W[k], Cin, N, 1)) return LEOD_ERR_UNSUPPORTED;
    }
    return conv3_wgrad_group_launch(n, dy, x, dw, ws, B, H, W, Cin, N, 1, stream, true);
}

LEOD_API int leod_stem_conv_wgrad(const float* dy, const void* x, int x_is_u8, float* dw, int B, int Cin, int H, int W,
                                  int Hp, int Wp, int N, int ks, int stride, int pad, hipStream_t stream) {
    if (!dy || !x || !dw) return LEOD_ERR_ARG;
    if (ks < 1 || ks > 15) return LEOD_ERR_UNSUPPORTED;
    const int Ho = (Hp + 2 * pad - ks) / stride + 1, Wo = (Wp + 2 * pad - ks) / stride + 1;
    const int M = B * Ho * Wo, K = Cin * ks * ks;
    static const int stem_patch = 1;
    if (stem_patch && ks == 7 && x_is_u8 && leod_precision() == 1 && stem_wgrad_bf16_supported(x, Cin, H, W, N, stride, pad))
        return stem_wgrad_bf16_launch(dy, x, dw, B, Cin, H, W, Ho, Wo, N, stream);     // k_stem.hip
    if (stem_patch && ks == 7 && x_is_u8 && stride == 4 && pad == 3 && N <= 64 && !(N & 15) && !(W & 3) && Cin * 19 * 18 <= 27 * 256 &&
        ((uintptr_t)x & 3) == 0 && ((long)Cin * H * W) % 4 == 0) {
        switch (N / 16) {
            case 1: return launch_stem_u8_wgrad<1>(dy, (const uint8_t*)x, dw, B, Cin, H, W, Ho, Wo, N, stream);
            case 2: return launch_stem_u8_wgrad<2>(dy, (const uint8_t*)x, dw, B, Cin, H, W, Ho, Wo, N, stream);
            case 3: return launch_stem_u8_wgrad<3>(dy, (const uint8_t*)x, dw, B, Cin, H, W, Ho, Wo, N, stream);
            default: return launch_stem_u8_wgrad<4>(dy, (const uint8_t*)x, dw, B, Cin, H, W, Ho, Wo, N, stream);
        }
    }
    if (x_is_u8) {
        XStemNCHW<uint8_t> xl{(const uint8_t*)x, Cin, H, W, Ho, Wo, ks, stride, pad};
        return wgrad_any(dy, xl, dw, (long)K, nullptr, M, N, K, stream);
    }
    XStemNCHW<float> xl{(const float*)x, Cin, H, W, Ho, Wo, ks, stride, pad};
    return wgrad_any(dy, xl, dw, (long)K, nullptr, M, N, K, stream);
}
