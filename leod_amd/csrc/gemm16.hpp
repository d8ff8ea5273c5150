// fp32 MFMA (v_mfma_f32_16x16x4_f32) GEMM skeletons shared by every contraction on the LEOD path.
//
//   gemm_lds_kernel   Out[m][n]  = sum_k A(m,k) * B(n,k)   forward / dgrad / implicit-GEMM conv at M >= 2048: operands
//                     fetched with coalesced 16-byte loads, staged in ONE LDS buffer (register prefetch of the next K
//                     chunk), conflict-free fragment reads, row-layout float4 epilogue, XCD-aware 1-D grid
//   gemm16_kernel     same contraction for small M (per-timestep LSTM cells of stages 3/4, micro shapes): operands
//                     straight from global memory in MFMA layout, optional 4-way K split over the waves
//   wgradw_kernel     dW[n][k] += sum_m dY(m,n) * X(m,k) for M >= 8192 (Linear layers of the time-batched step):
//                     workgroup tile chosen per layer shape, 3x3 MFMA tiles per wave, natural-layout LDS tiles
//   wgrad16_kernel    the same for small M and for the im2col / stem X loaders
//
// Common conventions:
//   * K-permutation: a chunk is 16 consecutive k; lane (i = lane & 15, q = lane >> 4) holds k0+4q..+3 as one float4 and
//     MFMA j of the chunk consumes component j -- every fragment is a single 16-byte access, for both operands.
//   * A loaders (AL*): rows of an activation / gradient matrix (plain rows, [x | h] concatenation, im2col of an NHWC map,
//     the raw NCHW uint8 stem input, transposed-conv gathers), optional LayerNorm / per-k scale applied on load.
//   * B loaders (BL*): weights in their stored layout (row-major, transposed, ConvLSTM gate-major, conv [N][Cin][ks][ks]).
//   * Epilogues (Ep*): bias / GELU (dual store) / gelu' / folded BatchNorm+SiLU / LayerScale+residual / LSTM gates,
//     column statistics for BatchNorm and bias gradients.
//   * LDS bank rules that fix the layouts (MI355X_MICROARCH.md, LDS): ds_read_b128 is served in the lane groups
//     {0-3,12-15,20-27},...: row-major tiles read as fragments need a row stride == 8 (mod 16) dwords; operands that arrive
//     transposed stay in natural layout (16-byte stores) and are read with 4 x ds_read_b32 at a stride == 4 (mod 8).
//
#pragma once
#include <type_traits>
#include <stdlib.h>
#include "common.hpp"

// =================================================================================================
// A loaders (row operand).  init() is called by all 64 lanes (may shuffle); load() returns the float4
// A(row, k..k+3) or zeros when out of range.
// =================================================================================================
struct ALRows {                 // plain rows, optional LayerNorm prologue, optional per-k scale
    const float* x; long ld;
    const float* ln_w; const float* ln_b; float eps;   // ln_w != nullptr -> normalise on the fly
    const float* kscale;                                // optional per-k multiplier (LayerScale bwd)
    float* stats_out;                                   // optional [M,2] (mean, rstd) written by n-block 0
    int K;                                              // row length used for the LN statistics
    const float* stats_in;                              // optional precomputed [M,2] (mean, rstd): skips the statistics passes
    int fmt;                                            // 2: bf16 rows; 3: fp16 rows (plain); 0: x is fp32; 1: x is an fp16 PRE-activation, A = gelu(x) (the MLP hidden of stages 1-2 is stored
                                                        // once, as fp16, in precision mode bf16: maxvit.py:110-118 under the reference's autocast)
    struct St { const float* p; float mean, rstd; bool ok; };
    __device__ __forceinline__ int klen(const St&, int K) const { return K; }
    __device__ __forceinline__ int aux(const St&) const { return 0; }
    __device__ __forceinline__ St init(int row, int M, int lane, bool write_stats) const {
        St s; s.ok = row < M; s.p = x + (long)(s.ok ? row : M - 1) * ld; s.mean = 0.f; s.rstd = 1.f;
        if (fmt) s.p = reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(x) + (long)(s.ok ? row : M - 1) * ld);
        if (ln_w && stats_in) { s.mean = stats_in[2 * (long)(s.ok ? row : M - 1)]; s.rstd = stats_in[2 * (long)(s.ok ? row : M - 1) + 1]; }
        else if (ln_w) {
            const int q = lane >> 4;
            float sum = 0.f;
            for (int k = 4 * q; k < K; k += 16) { f4 v = ld4(s.p + k); sum += (v.x + v.y) + (v.z + v.w); }
            sum = quad16_sum(sum);
            const float mean = sum / (float)K;
            float var = 0.f;
            for (int k = 4 * q; k < K; k += 16) {
                f4 v = ld4(s.p + k);
                const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
                var += (a * a + b * b) + (c * c + d * d);
            }
            var = quad16_sum(var) / (float)K;
            s.mean = mean; s.rstd = rsqrtf(var + eps);
            if (write_stats && stats_out && s.ok && q == 0) { stats_out[2 * (long)row] = s.mean; stats_out[2 * (long)row + 1] = s.rstd; }
        }
        return s;
    }
    __device__ __forceinline__ f4 load(const St& s, int k, int Kt) const {
        if (k >= Kt) return zero4();
        if (fmt == 1) {
            f4 v = unpack_h16(*reinterpret_cast<const s4*>(reinterpret_cast<const unsigned short*>(s.p) + k));
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
            return s.ok ? v : zero4();
        }
        if (fmt == 2) {                                  // bf16 gradient rows (precision mode bf16: du / dqkv are stored as the bf16 the MFMAs consume)
            f4 v = unpack_bf16(*reinterpret_cast<const s4*>(reinterpret_cast<const unsigned short*>(s.p) + k));
            if (kscale) v = v * ld4(kscale + k);
            return s.ok ? v : zero4();
        }
        if (fmt == 3) {                                  // fp16 activation rows (precision mode 16f: the attention output)
            f4 v = unpack_h16(*reinterpret_cast<const s4*>(reinterpret_cast<const unsigned short*>(s.p) + k));
            return s.ok ? v : zero4();
        }
        f4 v = ld4(s.p + k);
        if (ln_w) {
            const f4 g = ld4(ln_w + k), b = ld4(ln_b + k);
            v = (v - s.mean) * s.rstd * g + b;
        }
        if (kscale) v = v * ld4(kscale + k);
        return s.ok ? v : zero4();
    }
    template <int KCH> __device__ __forceinline__ f4 load2(const St& st, int k0, int ak, int Kt) const { return load(st, k0 + ak, Kt); }
};

// ALRows with its modes fixed at compile time and the access split in two phases (gemm_lds_kernel): raw() only ISSUES the loads of a
// slot -- row data, LayerNorm weight / bias, per-k scale -- and fin() applies ALRows::load's arithmetic later (same operations in the
// same order: bit-identical values).  With the arithmetic sitting behind each load and the modes tested at run time, the compiler
// waited for every load of a chunk in turn.  launch_gemm_lds dispatches the hot mode combinations here; the rest stays on ALRows.
template <int FMT, bool LN, bool KS>
struct ALRowsM : ALRows {
    static constexpr bool kTwoPhase = true;
    struct Raw { f4 v; u2_ h; f4 g, b, s; };
    __device__ __forceinline__ void raw(const St& st, int k, Raw& r) const {
        if constexpr (FMT == 0) r.v = ld4(st.p + k);
        else r.h = *reinterpret_cast<const u2_*>(reinterpret_cast<const unsigned short*>(st.p) + k);
        if constexpr (LN) { r.g = ld4(ln_w + k); r.b = ld4(ln_b + k); }
        if constexpr (KS) r.s = ld4(kscale + k);
    }
    // split form for kernels whose staging slots of a thread share ONE k offset (gemm_wide_bf16_kernel): the row data per slot, the
    // per-k vectors (LayerNorm weight / bias, scale) once per chunk into the Raw of slot 0
    __device__ __forceinline__ void raw_x(const St& st, int k, Raw& r) const {
        if constexpr (FMT == 0) r.v = ld4(st.p + k);
        else r.h = *reinterpret_cast<const u2_*>(reinterpret_cast<const unsigned short*>(st.p) + k);
    }
    __device__ __forceinline__ void raw_k(int k, Raw& r) const {
        if constexpr (LN) { r.g = ld4(ln_w + k); r.b = ld4(ln_b + k); }
        if constexpr (KS) r.s = ld4(kscale + k);
    }
    __device__ __forceinline__ f4 fin_k(const St& st, const Raw& r, const Raw& rk) const {
        f4 v;
        if constexpr (FMT == 1) {
            v = unpack_h16(__builtin_bit_cast(s4, r.h));
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
        } else if constexpr (FMT == 2) v = unpack_bf16(__builtin_bit_cast(s4, r.h));
        else if constexpr (FMT == 3) v = unpack_h16(__builtin_bit_cast(s4, r.h));
        else v = r.v;
        if constexpr (LN) v = (v - st.mean) * st.rstd * rk.g + rk.b;
        if constexpr (KS) v = v * rk.s;
        return v;
    }
    __device__ __forceinline__ f4 fin(const St& st, const Raw& r) const {
        f4 v;
        if constexpr (FMT == 1) {
            v = unpack_h16(__builtin_bit_cast(s4, r.h));
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
        } else if constexpr (FMT == 2) v = unpack_bf16(__builtin_bit_cast(s4, r.h));
        else if constexpr (FMT == 3) v = unpack_h16(__builtin_bit_cast(s4, r.h));
        else v = r.v;
        if constexpr (LN) v = (v - st.mean) * st.rstd * r.g + r.b;
        if constexpr (KS) v = v * r.s;
        return v;
    }
};
template <class AL, class = void> struct a_two_phase { static constexpr bool value = false; struct Raw {}; };
template <class AL> struct a_two_phase<AL, decltype((void)AL::kTwoPhase)> { static constexpr bool value = true; typedef typename AL::Raw Raw; };

struct ALConcat2 {              // [x1 (K1 cols) | x2] along k  (ConvLSTM: cat(x, h_prev))
    const float* x1; long ld1; int K1; const float* x2; long ld2;
    struct St { const float* p1; const float* p2; bool ok; };
    __device__ __forceinline__ int klen(const St&, int K) const { return K; }
    __device__ __forceinline__ int aux(const St&) const { return 0; }
    __device__ __forceinline__ St init(int row, int M, int, bool) const {
        St s; s.ok = row < M; const long r = s.ok ? row : M - 1;
        s.p1 = x1 + r * ld1; s.p2 = x2 ? x2 + r * ld2 : nullptr; return s;
    }
    __device__ __forceinline__ f4 load(const St& s, int k, int Kt) const {
        if (k >= Kt || !s.ok) return zero4();
        if (k < K1) return ld4(s.p1 + k);
        return s.p2 ? ld4(s.p2 + (k - K1)) : zero4();      // x2 == nullptr: zero initial state
    }
    template <int KCH> __device__ __forceinline__ f4 load2(const St& st, int k0, int ak, int Kt) const { return load(st, k0 + ak, Kt); }
};

struct ALConvNHWC {             // implicit-GEMM im2col over an NHWC fp32 map; k' = tap*Cin + c
    const float* x; int H, W, Cin, Ho, Wo, ks, stride, pad;
    struct St { int b, iy0, ix0; bool ok; };
    __device__ __forceinline__ int klen(const St&, int K) const { return K; }
    __device__ __forceinline__ int aux(const St&) const { return 0; }
    __device__ __forceinline__ St init(int row, int M, int, bool) const {
        St s; s.ok = row < M; const int r = s.ok ? row : 0;
        const int ox = r % Wo, t = r / Wo; const int oy = t % Ho; s.b = t / Ho;
        s.iy0 = oy * stride - pad; s.ix0 = ox * stride - pad; return s;
    }
    __device__ __forceinline__ f4 load(const St& s, int k, int Kt) const {
        if (k >= Kt || !s.ok) return zero4();
        const int tap = k / Cin, c = k - tap * Cin;
        const int kh = tap / ks, kw = tap - kh * ks;
        const int iy = s.iy0 + kh, ix = s.ix0 + kw;
        if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) return zero4();
        return ld4(x + (((long)s.b * H + iy) * W + ix) * Cin + c);
    }
    // chunk form used by the LDS GEMM: k = k0 + ak with k0 workgroup-uniform.  When Cin % KCH == 0 a chunk never
    // straddles a tap, so (tap, kh, kw) come out of scalar (wave-uniform) divisions instead of ~40 VALU per slot.
    template <int KCH> __device__ __forceinline__ f4 load2(const St& s, int k0, int ak, int Kt) const {
        if (Cin % KCH != 0) return load(s, k0 + ak, Kt);
        if (k0 + ak >= Kt || !s.ok) return zero4();
        const int tap = k0 / Cin, c = k0 - tap * Cin + ak;
        const int kh = tap / ks, kw = tap - kh * ks;
        const int iy = s.iy0 + kh, ix = s.ix0 + kw;
        if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) return zero4();
        return ld4(x + (((long)s.b * H + iy) * W + ix) * Cin + c);
    }
};

template <typename T>
struct ALStemNCHW {             // stem conv over the raw NCHW event tensor (uint8 or fp32), k = c*ks*ks + kh*ks + kw
    const T* x; int Cin, H, W;  // H,W: stored (unpadded) size; anything outside reads as 0 (= bottom/right zero pad)
    int Ho, Wo, ks, stride, pad;
    struct St { const T* p; int iy0, ix0; bool ok; };
    __device__ __forceinline__ int klen(const St&, int K) const { return K; }
    __device__ __forceinline__ int aux(const St&) const { return 0; }
    __device__ __forceinline__ St init(int row, int M, int, bool) const {
        St s; s.ok = row < M; const int r = s.ok ? row : 0;
        const int ox = r % Wo, t = r / Wo; const int oy = t % Ho; const int b = t / Ho;
        s.p = x + (long)b * Cin * H * W; s.iy0 = oy * stride - pad; s.ix0 = ox * stride - pad; return s;
    }
    __device__ __forceinline__ float one(const St& s, int k) const {
        // the shipped RVT stem is 7x7 (patch 4 -> kernel 2*4-1): constant divisors compile to multiply-shift; non-overlapping patches
        // (downsample.overlap = False: ks = stride) take the run-time decode
        int c, kh, kw;
        if (ks == 7) { c = k / 49; const int r = k - c * 49; kh = r / 7; kw = r - kh * 7; }
        else { const int kk = ks * ks; c = k / kk; const int r = k - c * kk; kh = r / ks; kw = r - kh * ks; }
        const int iy = s.iy0 + kh, ix = s.ix0 + kw;
        if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) return 0.f;
        return (float)s.p[((long)c * H + iy) * W + ix];
    }
    __device__ __forceinline__ f4 load(const St& s, int k, int Kt) const {
        if (k >= Kt || !s.ok) return zero4();
        f4 v; v.x = one(s, k); v.y = one(s, k + 1); v.z = one(s, k + 2); v.w = one(s, k + 3); return v;
    }
    template <int KCH> __device__ __forceinline__ f4 load2(const St& st, int k0, int ak, int Kt) const { return load(st, k0 + ak, Kt); }
};

struct ALConvT {                // dgrad of a conv: rows = input pixels, k' = tap*N + n over dY (NHWC [B,Ho,Wo,N])
    const float* dy; int H, W, Ho, Wo, N, ks, stride, pad;
    struct St { int b, iy, ix; bool ok; };
    __device__ __forceinline__ int klen(const St&, int K) const { return K; }
    __device__ __forceinline__ int aux(const St&) const { return 0; }
    __device__ __forceinline__ St init(int row, int M, int, bool) const {
        St s; s.ok = row < M; const int r = s.ok ? row : 0;
        s.ix = r % W; const int t = r / W; s.iy = t % H; s.b = t / H; return s;
    }
    __device__ __forceinline__ f4 load(const St& s, int k, int Kt) const {
        if (k >= Kt || !s.ok) return zero4();
        const int tap = k / N, n = k - tap * N;
        const int kh = tap / ks, kw = tap - kh * ks;
        const int ty = s.iy + pad - kh, tx = s.ix + pad - kw;
        if (ty < 0 || tx < 0 || (ty % stride) || (tx % stride)) return zero4();
        const int oy = ty / stride, ox = tx / stride;
        if (oy >= Ho || ox >= Wo) return zero4();
        return ld4(dy + (((long)s.b * Ho + oy) * Wo + ox) * N + n);
    }
    template <int KCH> __device__ __forceinline__ f4 load2(const St& s, int k0, int ak, int Kt) const {
        if (N % KCH != 0) return load(s, k0 + ak, Kt);
        if (k0 + ak >= Kt || !s.ok) return zero4();
        const int tap = k0 / N, n = k0 - tap * N + ak;               // wave-uniform tap
        const int kh = tap / ks, kw = tap - kh * ks;
        const int ty = s.iy + pad - kh, tx = s.ix + pad - kw;
        if (ty < 0 || tx < 0 || (ty % stride) || (tx % stride)) return zero4();
        const int oy = ty / stride, ox = tx / stride;
        if (oy >= Ho || ox >= Wo) return zero4();
        return ld4(dy + (((long)s.b * Ho + oy) * Wo + ox) * N + n);
    }
};

// dgrad of a 3x3 stride-2 pad-1 conv with rows grouped by input-pixel parity class (py,px): a pixel of class
// (py,px) only receives (1+py)*(1+px) of the 9 taps, so each class contracts over just its live taps (1,2,2,4
// instead of 9).  Row r: class = r / Q, (b, y2, x2) = unravel(r % Q), pixel = (2*y2+py, 2*x2+px); Q = B*(H/2)*(W/2)
// must be a multiple of 16 so that a 16-row wave tile never mixes classes.  k' = t*N + n, t = live-tap index.
struct ALConvT2 {
    const float* dy; int H, W, Ho, Wo, N, Q;
    struct St { int b, iy, ix, cls; bool ok; };
    __device__ __forceinline__ St init(int row, int M, int, bool) const {
        St s; s.ok = row < M; const int r = s.ok ? row : 0;
        s.cls = r / Q; const int rem = r - s.cls * Q;
        const int W2 = W >> 1, H2 = H >> 1;
        const int x2 = rem % W2, t = rem / W2; const int y2 = t % H2; s.b = t / H2;
        s.iy = 2 * y2 + (s.cls >> 1); s.ix = 2 * x2 + (s.cls & 1); return s;
    }
    __device__ __forceinline__ int klen(const St& s, int) const { return (1 + (s.cls >> 1)) * (1 + (s.cls & 1)) * N; }
    __device__ __forceinline__ int aux(const St& s) const { return s.cls; }
    __device__ __forceinline__ f4 load(const St& s, int k, int) const {
        const int py = s.cls >> 1, px = s.cls & 1;
        if (!s.ok || k >= (1 + py) * (1 + px) * N) return zero4();
        const int t = k / N, n = k - t * N;
        const int khi = t / (1 + px), kwi = t - khi * (1 + px);
        // even coordinate: centre tap (k=1, o = i/2); odd: k=0 -> o=(i+1)/2, k=2 -> o=(i-1)/2
        const int oy = py ? (khi == 0 ? (s.iy + 1) >> 1 : (s.iy - 1) >> 1) : s.iy >> 1;
        const int ox = px ? (kwi == 0 ? (s.ix + 1) >> 1 : (s.ix - 1) >> 1) : s.ix >> 1;
        if (oy >= Ho || ox >= Wo) return zero4();
        return ld4(dy + (((long)s.b * Ho + oy) * Wo + ox) * N + n);
    }
    template <int KCH> __device__ __forceinline__ f4 load2(const St& s, int k0, int ak, int Kt) const {
        if (N % KCH != 0) return load(s, k0 + ak, Kt);
        const int py = s.cls >> 1, px = s.cls & 1;
        if (!s.ok || k0 + ak >= (1 + py) * (1 + px) * N) return zero4();
        const int t = k0 / N, n = k0 - t * N + ak;                   // wave-uniform live-tap index
        const int khi = t / (1 + px), kwi = t - khi * (1 + px);
        const int oy = py ? (khi == 0 ? (s.iy + 1) >> 1 : (s.iy - 1) >> 1) : s.iy >> 1;
        const int ox = px ? (kwi == 0 ? (s.ix + 1) >> 1 : (s.ix - 1) >> 1) : s.ix >> 1;
        if (oy >= Ho || ox >= Wo) return zero4();
        return ld4(dy + (((long)s.b * Ho + oy) * Wo + ox) * N + n);
    }
};

// =================================================================================================
// B loaders (column operand = weights).  load(nblk, t, i, k) -> float4 B(n, k..k+3)
// =================================================================================================
struct BLRows {
    static constexpr bool kTrans = false;
    __device__ __forceinline__ f4 load_n4(int, int, int, int) const { return zero4(); }                 // W[n][k], row stride ld (torch Linear / 1x1 conv weight)
    template <int KCH> __device__ __forceinline__ f4 load2(int nblk, int t, int i, int k0, int bk, int Kt, int a = 0) const { return load(nblk, t, i, k0 + bk, Kt, a); }
    const float* w; long ld; int N; int NT;
    const unsigned short* w16 = nullptr;          // bf16 shadow of w (leod_shadow_of), same indexing; precision mode bf16 only
    __device__ __forceinline__ int col(int nblk, int t, int i) const { return (nblk * NT + t) * 16 + i; }
    __device__ __forceinline__ f4 load(int nblk, int t, int i, int k, int Kt, int = 0) const {
        const int n = col(nblk, t, i);
        if (n >= N || k >= Kt) return zero4();
        return ld4(w + (long)n * ld + k);
    }
    // four bf16 W[n][k..k+3] of the shadow (0 outside the matrix)
    __device__ __forceinline__ u2_ load16(int nblk, int t, int i, int k, int Kt) const {
        const int n = col(nblk, t, i);
        if (n >= N || k >= Kt) return u2_{0u, 0u};
        return *reinterpret_cast<const u2_*>(w16 + (long)n * ld + k);
    }
};
struct BLGates {
    static constexpr bool kTrans = false;
    __device__ __forceinline__ f4 load_n4(int, int, int, int) const { return zero4(); }                // ConvLSTM: tile t = gate t (f,i,o,g), columns nblk*16.. of that gate; W[4C][K]
    template <int KCH> __device__ __forceinline__ f4 load2(int nblk, int t, int i, int k0, int bk, int Kt, int a = 0) const { return load(nblk, t, i, k0 + bk, Kt, a); }
    const float* w; long ld; int C;
    __device__ __forceinline__ int col(int nblk, int t, int i) const { return t * C + nblk * 16 + i; }
    __device__ __forceinline__ f4 load(int nblk, int t, int i, int k, int Kt, int = 0) const {
        if (nblk * 16 + i >= C || k >= Kt) return zero4();
        return ld4(w + (long)col(nblk, t, i) * ld + k);
    }
};
struct BLTrans {
    static constexpr bool kTrans = true;          // memory is contiguous along n (W[k][n]): stage with float4 along n                // B(n,k) = W[k][n]  (dgrad of a Linear: W is [Kred][N])
    template <int KCH> __device__ __forceinline__ f4 load2(int nblk, int t, int i, int k0, int bk, int Kt, int a = 0) const { return load(nblk, t, i, k0 + bk, Kt, a); }
    const float* w; long ld; int N; int NT;
    const unsigned short* w16 = nullptr;          // bf16 shadow of w (leod_shadow_of), same indexing; precision mode bf16 only
    __device__ __forceinline__ u2_ load_n4_16(int nblk, int nl, int k, int Kt) const {
        const int n = nblk * NT * 16 + nl;
        if (n >= N || k >= Kt) return u2_{0u, 0u};
        return *reinterpret_cast<const u2_*>(w16 + (long)k * ld + n);
    }
    __device__ __forceinline__ int col(int nblk, int t, int i) const { return (nblk * NT + t) * 16 + i; }
    __device__ __forceinline__ f4 load(int nblk, int t, int i, int k, int Kt, int = 0) const {
        const int n = col(nblk, t, i);
        if (n >= N || k >= Kt) return zero4();
        const float* p = w + (long)k * ld + n;
        f4 v; v.x = p[0]; v.y = p[ld]; v.z = p[2 * ld]; v.w = p[3 * ld]; return v;
    }
    // (B(n,k), B(n+1,k), B(n+2,k), B(n+3,k)) = W[k][n..n+3] for the local column nl of n-block nblk (N % 4 == 0)
    __device__ __forceinline__ f4 load_n4(int nblk, int nl, int k, int Kt) const {
        const int n = nblk * NT * 16 + nl;
        if (n >= N || k >= Kt) return zero4();
        return ld4(w + (long)k * ld + n);
    }
};
// weight loaders that stage their tiles from the bf16 shadow (w16 != nullptr, set by the launchers): separate TYPES, so that the kernels
// carry no run-time branch around their loads (a uniform branch inside the fetch block made the compiler wait for all loads at the join,
// i.e. before the MFMAs the loads are supposed to fly under: 16.15 -> 16.8 ms per step)
struct BLRows16 : BLRows {};
struct BLTrans16 : BLTrans {};
template <class BL> struct bl_is16 { static constexpr bool value = false; };
template <> struct bl_is16<BLRows16> { static constexpr bool value = true; };
template <> struct bl_is16<BLTrans16> { static constexpr bool value = true; };
// weight loaders of FORWARD contractions (W[n][k] as stored): the only ones instantiated with fp16 operands (precision mode 16f)
template <class BL> struct bl_fwd { static constexpr bool value = !BL::kTrans; };
struct BLConvWT; struct BLConvWT2; struct BLPackT2;
template <> struct bl_fwd<BLConvWT> { static constexpr bool value = false; };
template <> struct bl_fwd<BLConvWT2> { static constexpr bool value = false; };
template <> struct bl_fwd<BLPackT2> { static constexpr bool value = false; };
template <class BL> struct bl_shadow_type { typedef void type; };
template <> struct bl_shadow_type<BLRows> { typedef BLRows16 type; };
template <> struct bl_shadow_type<BLTrans> { typedef BLTrans16 type; };
// (the four bf16 travel in the first two dwords of the f4 staging register; whole-vector bit casts -- per-element casts inside a braced
// initialiser were folded by the compiler into ONE dword load duplicated into both halves)
typedef unsigned u4s_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 shadow_bits(u2_ h) { const u4s_ u = {h.x, h.y, 0u, 0u}; return __builtin_bit_cast(f4, u); }
__device__ __forceinline__ s4 shadow_s4(f4 v) { const u4s_ u = __builtin_bit_cast(u4s_, v); const u2_ r = {u.x, u.y}; return __builtin_bit_cast(s4, r); }

struct BLConvW {
    static constexpr bool kTrans = false;
    __device__ __forceinline__ f4 load_n4(int, int, int, int) const { return zero4(); }                // conv weight [N][Cin][ks][ks] read as B(n, k' = tap*Cin + c)
    const float* w; int N, Cin, KK; int NT;
    __device__ __forceinline__ int col(int nblk, int t, int i) const { return (nblk * NT + t) * 16 + i; }
    __device__ __forceinline__ f4 load(int nblk, int t, int i, int k, int Kt, int = 0) const {
        const int n = col(nblk, t, i);
        if (n >= N || k >= Kt) return zero4();
        const int tap = k / Cin, c = k - tap * Cin;
        const float* p = w + ((long)n * Cin + c) * KK + tap;
        f4 v; v.x = p[0]; v.y = p[KK]; v.z = p[2 * KK]; v.w = p[3 * KK]; return v;
    }
    template <int KCH> __device__ __forceinline__ f4 load2(int nblk, int t, int i, int k0, int bk, int Kt, int = 0) const {
        if (Cin % KCH != 0) return load(nblk, t, i, k0 + bk, Kt);
        const int n = col(nblk, t, i);
        if (n >= N || k0 + bk >= Kt) return zero4();
        const int tap = k0 / Cin, c = k0 - tap * Cin + bk;           // wave-uniform tap
        const float* p = w + ((long)n * Cin + c) * KK + tap;
        f4 v; v.x = p[0]; v.y = p[KK]; v.z = p[2 * KK]; v.w = p[3 * KK]; return v;
    }
};
struct BLConvWT {
    static constexpr bool kTrans = false;
    __device__ __forceinline__ f4 load_n4(int, int, int, int) const { return zero4(); }               // dgrad: B(col = c, k' = tap*N + n) = W[n][c][tap]
    const float* w; int N, Cin, KK; int NT;
    __device__ __forceinline__ int col(int nblk, int t, int i) const { return (nblk * NT + t) * 16 + i; }
    __device__ __forceinline__ f4 load(int nblk, int t, int i, int k, int Kt, int = 0) const {
        const int c = col(nblk, t, i);
        if (c >= Cin || k >= Kt) return zero4();
        const int tap = k / N, n = k - tap * N;
        const long sn = (long)Cin * KK;
        const float* p = w + ((long)n * Cin + c) * KK + tap;
        f4 v; v.x = p[0]; v.y = p[sn]; v.z = p[2 * sn]; v.w = p[3 * sn]; return v;
    }
    template <int KCH> __device__ __forceinline__ f4 load2(int nblk, int t, int i, int k0, int bk, int Kt, int = 0) const {
        if (N % KCH != 0) return load(nblk, t, i, k0 + bk, Kt);
        const int c = col(nblk, t, i);
        if (c >= Cin || k0 + bk >= Kt) return zero4();
        const int tap = k0 / N, n = k0 - tap * N + bk;               // wave-uniform tap
        const long sn = (long)Cin * KK;
        const float* p = w + ((long)n * Cin + c) * KK + tap;
        f4 v; v.x = p[0]; v.y = p[sn]; v.z = p[2 * sn]; v.w = p[3 * sn]; return v;
    }
};

struct BLConvWT2 {
    static constexpr bool kTrans = false;
    __device__ __forceinline__ f4 load_n4(int, int, int, int) const { return zero4(); }              // parity-class dgrad of a 3x3/s2 conv: B(col = c, k' = t*N + n) = W[n][c][kh][kw], (kh,kw) from (class, t)
    const float* w; int N, Cin; int NT;
    __device__ __forceinline__ int col(int nblk, int t, int i) const { return (nblk * NT + t) * 16 + i; }
    __device__ __forceinline__ f4 load(int nblk, int t, int i, int k, int, int cls) const {
        const int c = col(nblk, t, i);
        const int py = cls >> 1, px = cls & 1;
        if (c >= Cin || k >= (1 + py) * (1 + px) * N) return zero4();
        const int tt = k / N, n = k - tt * N;
        const int khi = tt / (1 + px), kwi = tt - khi * (1 + px);
        const int kh = py ? 2 * khi : 1, kw = px ? 2 * kwi : 1;
        const long sn = (long)Cin * 9;
        const float* p = w + ((long)n * Cin + c) * 9 + kh * 3 + kw;
        f4 v; v.x = p[0]; v.y = p[sn]; v.z = p[2 * sn]; v.w = p[3 * sn]; return v;
    }
    template <int KCH> __device__ __forceinline__ f4 load2(int nblk, int t, int i, int k0, int bk, int Kt, int cls) const {
        if (N % KCH != 0) return load(nblk, t, i, k0 + bk, Kt, cls);
        const int c = col(nblk, t, i);
        const int py = cls >> 1, px = cls & 1;
        if (c >= Cin || k0 + bk >= (1 + py) * (1 + px) * N) return zero4();
        const int tt = k0 / N, n = k0 - tt * N + bk;                 // wave-uniform live-tap index
        const int khi = tt / (1 + px), kwi = tt - khi * (1 + px);
        const int kh = py ? 2 * khi : 1, kw = px ? 2 * kwi : 1;
        const long sn = (long)Cin * 9;
        const float* p = w + ((long)n * Cin + c) * 9 + kh * 3 + kw;
        f4 v; v.x = p[0]; v.y = p[sn]; v.z = p[2 * sn]; v.w = p[3 * sn]; return v;
    }
};

struct BLPackT2 {                 // parity-class dgrad on PACKED weights wd[c][tap*N + n] (see conv_pack_kernel)
    static constexpr bool kTrans = false;
    __device__ __forceinline__ f4 load_n4(int, int, int, int) const { return zero4(); }
    const float* w; int N, Cin; int NT;
    __device__ __forceinline__ int col(int nblk, int t, int i) const { return (nblk * NT + t) * 16 + i; }
    __device__ __forceinline__ f4 load(int nblk, int t, int i, int k, int, int cls) const {
        const int c = col(nblk, t, i);
        const int py = cls >> 1, px = cls & 1;
        if (c >= Cin || k >= (1 + py) * (1 + px) * N) return zero4();
        const int tt = k / N, n = k - tt * N;
        const int khi = tt / (1 + px), kwi = tt - khi * (1 + px);
        const int kh = py ? 2 * khi : 1, kw = px ? 2 * kwi : 1;
        return ld4(w + ((long)c * 9 + kh * 3 + kw) * N + n);
    }
    template <int KCH> __device__ __forceinline__ f4 load2(int nblk, int t, int i, int k0, int bk, int Kt, int cls) const {
        return load(nblk, t, i, k0 + bk, Kt, cls);
    }
};

// =================================================================================================
// Epilogues.  acc[t][r] = Out[row0 + 4*(lane>>4) + r][col(nblk,t,lane&15)]
// =================================================================================================
enum { ACT_NONE = 0, ACT_GELU_DUAL = 1, ACT_AFFINE_SILU = 2, ACT_MUL_GELU_GRAD = 3 };
struct RowPre { f4 v[4]; };
// f(std::integral_constant<int, mode>) for a run-time mode in [0, NM]
template <int NM, class F>
__device__ __forceinline__ void dispatch_fast_mode(int mode, F&& f) {
    if (mode == 0) f(std::integral_constant<int, 0>{});
    else if (NM >= 1 && mode == 1) f(std::integral_constant<int, (NM >= 1 ? 1 : 0)>{});
    else if (NM >= 2 && mode == 2) f(std::integral_constant<int, (NM >= 2 ? 2 : 0)>{});
    else if (NM >= 3 && mode == 3) f(std::integral_constant<int, (NM >= 3 ? 3 : 0)>{});
    else if (NM >= 4 && mode == 4) f(std::integral_constant<int, (NM >= 4 ? 4 : 0)>{});
    else if (NM >= 5 && mode == 5) f(std::integral_constant<int, (NM >= 5 ? 5 : 0)>{});
    else if (NM >= 6 && mode == 6) f(std::integral_constant<int, (NM >= 6 ? 6 : 0)>{});
    else f(std::integral_constant<int, 0>{});
}        // prefetched global operand of a row epilogue: one 16-byte piece per row group of the lane

struct EpStore {
    float* out; long ld;            // primary output
    float* out2; long ld2;          // ACT_GELU_DUAL: gelu(pre) goes to out2, pre to out;  split: cols >= nsplit
    int nsplit;                     // columns >= nsplit are routed to out2 (col - nsplit); 0 = no split
    const float* bias;              // optional per-column bias
    const float* aux; long ldaux;   // ACT_MUL_GELU_GRAD: u (pre-activation) ; ACT_AFFINE_SILU: unused
    const float* bn_w; const float* bn_b; const float* bn_rm; const float* bn_rv; float bn_eps;  // eval BN fold
    double* colstats;               // optional [R][2][N] (sum, sumsq) accumulated atomically (train BN); R = stat_rep replicas
    int stat_rep;                   // power of two (0 / 1 = one copy): wave-tile t adds into replica t & (R - 1).  All waves of a
                                    // launch hammering the SAME 2N doubles (12 cache lines for N = 96) serialised in the L2 atomic
                                    // units: +200 us on a 12 us 1x1 conv at M = 40960 (tools/kbench_conv.py); the consumer folds
    float* colsum;                  // optional [N]: += column sums of the stored value (bias gradient)
    int act; int accumulate;
    int N;
    const float* addsrc;            // optional [M][ld]: out = addsrc + value (second gradient source of a residual branch; not with nsplit / accumulate)
    int out_fmt;                    // row epilogue only (run_rows): 0 = fp32 out; 1 = out holds fp16 (clamped), 2 = bf16 -- the 16-bit tensors of
    int aux_fmt;                    // precision mode bf16 (MLP hidden pre-activation, qkv, du); aux_fmt 1: aux is an fp16 pre-activation
    int rm_Q, rm_H, rm_W;           // rm_Q > 0: GEMM rows are parity-class ordered (ALConvT2) -> remap to pixel rows of the [B,H,W] map
    __device__ __forceinline__ long maprow(int row) const {
        if (rm_Q <= 0) return row;
        const int cls = row / rm_Q, rem = row - cls * rm_Q;
        const int W2 = rm_W >> 1, H2 = rm_H >> 1;
        const int x2 = rem % W2, t = rem / W2; const int y2 = t % H2, b = t / H2;
        return ((long)b * rm_H + 2 * y2 + (cls >> 1)) * rm_W + 2 * x2 + (cls & 1);
    }
    template <int NT, class BL>
    __device__ __forceinline__ void run(f4 (&acc)[NT], const BL& bl, int row0, int nblk, int lane, int M) const {
        const int i = lane & 15, rg = lane >> 4;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = bl.col(nblk, t, i);
            const bool nok = n < N;
            const float bv = (bias && nok) ? bias[n] : 0.f;
            float sc = 1.f, sh = 0.f;
            if (act == ACT_AFFINE_SILU && nok) {
                sc = bn_w[n] * rsqrtf(bn_rv[n] + bn_eps); sh = bn_b[n] - bn_rm[n] * sc;
            }
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * rg + r;
                float v = acc[t][r] + bv;
                const bool ok = nok && grow < M;
                const long row = maprow(grow);
                if (act == ACT_AFFINE_SILU) v = siluf_(v * sc + sh);
                if (act == ACT_MUL_GELU_GRAD && ok) v *= gelu_erf_grad(aux[row * ldaux + n]);
                if (ok) {
                    if (nsplit > 0 && n >= nsplit) {
                        float* p = out2 + row * ld2 + (n - nsplit);
                        *p = accumulate ? *p + v : v;
                    } else {
                        float* p = out + row * ld + n;
                        if (addsrc) v += addsrc[row * ld + n];
                        *p = accumulate ? *p + v : v;
                        if (act == ACT_GELU_DUAL) out2[row * ld2 + n] = gelu_erf(v);
                    }
                    s1 += v; s2 += v * v;
                }
            }
            if (colstats || colsum) {
                s1 = quad16_sum(s1);
                if (colstats) s2 = quad16_sum(s2);
                if (rg == 0 && nok) {
                    if (colstats) {
                        double* cs = colstats + (stat_rep > 1 ? (size_t)((row0 >> 4) & (stat_rep - 1)) * 2 * N : 0);
                        atomicAdd(cs + n, (double)s1); atomicAdd(cs + N + n, (double)s2);
                    }
                    if (colsum) atomicAdd(colsum + n, s1);
                }
            }
        }
    }
    // Row-layout epilogue of the LDS-staged GEMM: the wave's 16 x (NT*16) accumulator tile has been transposed through
    // LDS (so[row][col], leading dimension ldo), lane (c4 = lane & 15, q = lane >> 4) owns columns 4*c4..4*c4+3 of rows
    // q, q+4, q+8, q+12 -> every global access is a 16-byte access and a wave instruction covers whole row segments
    // (NT*64 contiguous bytes per row) instead of 64-byte column slivers of 4-byte stores.
    static constexpr bool kRowEpilogue = true;
    static constexpr bool kPrefetchRows = false;
    // Branch-free body for a fragment that lies inside the matrix, in the configurations of the Linear layers (OUTF: 0 fp32 / 1 fp16 /
    // 2 bf16 store; GG: times gelu'(fp16 pre-activation); ADD: plus a second gradient source).  The generic body below has a
    // data-dependent branch per row group (`row < M`) and wave-uniform ones per option: at every join hipcc's waitcnt insertion falls
    // back to `s_waitcnt vmcnt(0)`, so each of the four row groups waited for the write acknowledgements of the previous one's stores
    // (and for every load in flight) -- 16 exposed memory round trips per 64-row tile of a one-workgroup-per-CU kernel.
    template <int OUTF, bool GG, bool ADD, bool STATS = false>
    __device__ __forceinline__ void run_rows_fast(const float* so, int ldo, int row0, int n, int c4, int q) const {
        const f4 bv = bias ? ld4(bias + n) : zero4();
        f4 s1 = zero4(), s2 = zero4();
        f4 ux[4];
        if constexpr (GG) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const f2_ h = *reinterpret_cast<const f2_*>(reinterpret_cast<const unsigned short*>(aux) + (long)(row0 + q + 4 * p) * ldaux + n);
                ux[p].x = h.x; ux[p].y = h.y;
            }
        }
        if constexpr (ADD) {
#pragma unroll
            for (int p = 0; p < 4; ++p) ux[p] = ld4(addsrc + (long)(row0 + q + 4 * p) * ld + n);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int lr = q + 4 * p;
            const long row = row0 + lr;
            f4 v = *reinterpret_cast<const f4*>(so + lr * ldo + 4 * c4) + bv;
            if constexpr (GG) {
                const f4 u = unpack_h16(__builtin_bit_cast(s4, f2_{ux[p].x, ux[p].y}));
                v.x *= gelu_erf_grad(u.x); v.y *= gelu_erf_grad(u.y); v.z *= gelu_erf_grad(u.z); v.w *= gelu_erf_grad(u.w);
            }
            if constexpr (ADD) v += ux[p];
            if constexpr (OUTF != 0) {
                unsigned short* pp = reinterpret_cast<unsigned short*>(out) + row * ld + n;
                *reinterpret_cast<s4*>(pp) = OUTF == 1 ? pack_h16(v) : pack_bf16(v);
            } else {
                *reinterpret_cast<f4*>(out + row * ld + n) = v;
            }
            if constexpr (STATS) { s1 += v; s2 += v * v; }
        }
        if constexpr (STATS) stats_tail(s1, s2, row0, n, q, true);
    }
    // column sums of the stored values of one fragment (BatchNorm statistics / bias gradient): 4 row groups folded by DPP, one atomic
    // per column from the q = 0 lanes.  Every lane of the wave must call (nok: this lane owns valid columns).
    __device__ __forceinline__ void stats_tail(const f4& s1, const f4& s2, int row0, int n, int q, bool nok) const {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = quad16_sum(s1[j]);
            const float b = colstats ? quad16_sum(s2[j]) : 0.f;
            if (q == 0 && nok) {
                if (colstats) {
                    double* cs = colstats + (stat_rep > 1 ? (size_t)((row0 >> 4) & (stat_rep - 1)) * 2 * N : 0);
                    atomicAdd(cs + n + j, (double)a); atomicAdd(cs + N + n + j, (double)b);
                }
                if (colsum) atomicAdd(colsum + n + j, a);
            }
        }
    }
    // Configuration of a launch as one of the branch-free bodies (wave-uniform; the kernels pick the body ONCE per tile, outside their
    // fragment loop: a choice inside run_rows joins the generic body after every fragment, and at that join the waitcnt pass drains
    // again).  0 generic; 1 fp32; 2 fp32 + addsrc; 3 fp16; 4 bf16; 5 gelu' x -> bf16; 6 fp32 + column statistics
    static constexpr int kFastModes = 6;
    __device__ __forceinline__ int fast_mode() const {
        const bool gg = act == ACT_MUL_GELU_GRAD, st = colstats || colsum;
        return (nsplit != 0 || accumulate || rm_Q > 0 || !(act == ACT_NONE || (gg && aux_fmt == 1))) ? 0
               : st ? ((!gg && out_fmt == 0 && !addsrc) ? 6 : 0)
               : gg ? ((out_fmt == 2 && !addsrc) ? 5 : 0)
               : out_fmt == 0 ? (addsrc ? 2 : 1) : (addsrc ? 0 : out_fmt == 1 ? 3 : 4);
    }
    template <int MODE, int NT, class BL>
    __device__ __forceinline__ RowPre prefetch_full(const BL&, int, int, int) const { return RowPre{}; }
    // fragment rows row0 .. row0 + 15 all inside the matrix
    template <int MODE, int NT, class BL>
    __device__ __forceinline__ void run_rows_full(const float* so, int ldo, const BL& bl, int row0, int nblk, int lane, const RowPre&) const {
        const int c4 = lane & 15, q = lane >> 4;
        const int n = bl.col(nblk, 0, 0) + 4 * c4;
        const bool nok = c4 < NT * 4 && n < N;
        if constexpr (MODE == 6) {                             // the DPP sums need every lane
            f4 s1 = zero4(), s2 = zero4();
            if (nok) {
                const f4 bv = bias ? ld4(bias + n) : zero4();
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int lr = q + 4 * p;
                    const f4 v = *reinterpret_cast<const f4*>(so + lr * ldo + 4 * c4) + bv;
                    *reinterpret_cast<f4*>(out + (long)(row0 + lr) * ld + n) = v;
                    s1 += v; s2 += v * v;
                }
            }
            stats_tail(s1, s2, row0, n, q, nok);
        } else if (nok) {
            if constexpr (MODE == 1) run_rows_fast<0, false, false>(so, ldo, row0, n, c4, q);
            else if constexpr (MODE == 2) run_rows_fast<0, false, true>(so, ldo, row0, n, c4, q);
            else if constexpr (MODE == 3) run_rows_fast<1, false, false>(so, ldo, row0, n, c4, q);
            else if constexpr (MODE == 4) run_rows_fast<2, false, false>(so, ldo, row0, n, c4, q);
            else run_rows_fast<2, true, false>(so, ldo, row0, n, c4, q);
        }
    }
    template <int NT, class BL>
    __device__ __forceinline__ void run_rows(const float* so, int ldo, const BL& bl, int row0, int nblk, int lane, int M) const {
        const int c4 = lane & 15, q = lane >> 4;
        const int n = bl.col(nblk, 0, 0) + 4 * c4;
        const bool nok = c4 < NT * 4 && n < N;
        f4 bv = zero4(), sc = {1.f, 1.f, 1.f, 1.f}, sh = zero4();
        if (nok) {
            if (bias) bv = ld4(bias + n);
            if (act == ACT_AFFINE_SILU) {
                const f4 w = ld4(bn_w + n), b = ld4(bn_b + n), rm = ld4(bn_rm + n), rv = ld4(bn_rv + n);
#pragma unroll
                for (int j = 0; j < 4; ++j) { sc[j] = w[j] * rsqrtf(rv[j] + bn_eps); sh[j] = b[j] - rm[j] * sc[j]; }
            }
        }
        f4 s1 = zero4(), s2 = zero4();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int lr = q + 4 * p, grow = row0 + lr;
            if (!nok || grow >= M) continue;
            f4 v = *reinterpret_cast<const f4*>(so + lr * ldo + 4 * c4) + bv;
            const long row = maprow(grow);
            if (act == ACT_AFFINE_SILU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = siluf_(v[j] * sc[j] + sh[j]);
            }
            if (act == ACT_MUL_GELU_GRAD) {
                const f4 u = aux_fmt ? unpack_h16(*reinterpret_cast<const s4*>(reinterpret_cast<const unsigned short*>(aux) + row * ldaux + n))
                                     : ld4(aux + row * ldaux + n);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= gelu_erf_grad(u[j]);
            }
            if (out_fmt) {                                      // 16-bit primary output (no split / accumulate / dual store in this mode)
                unsigned short* pp = reinterpret_cast<unsigned short*>(out) + row * ld + n;
                *reinterpret_cast<s4*>(pp) = out_fmt == 1 ? pack_h16(v) : pack_bf16(v);
            } else if (nsplit > 0 && n >= nsplit) {
                float* pp = out2 + row * ld2 + (n - nsplit);
                if (accumulate) v += ld4(pp);
                *reinterpret_cast<f4*>(pp) = v;
            } else {
                float* pp = out + row * ld + n;
                if (addsrc) v += ld4(addsrc + row * ld + n);
                if (accumulate) v += ld4(pp);
                *reinterpret_cast<f4*>(pp) = v;
                if (act == ACT_GELU_DUAL) {
                    f4 g;
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] = gelu_erf(v[j]);
                    *reinterpret_cast<f4*>(out2 + row * ld2 + n) = g;
                }
            }
            s1 += v; s2 += v * v;
        }
        if (colstats || colsum) stats_tail(s1, s2, row0, n, q, nok);
    }
};

struct EpLsRes {                    // t = acc + bias; tout = t; out = res + gamma * t   (LayerScale + residual)
    float* out; float* tout; const float* res; const float* bias; const float* gamma; long ld; int N;
    template <int NT, class BL>
    __device__ __forceinline__ void run(f4 (&acc)[NT], const BL& bl, int row0, int nblk, int lane, int M) const {
        const int i = lane & 15, rg = lane >> 4;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = bl.col(nblk, t, i);
            if (n >= N) continue;
            const float bv = bias ? bias[n] : 0.f, g = gamma ? gamma[n] : 1.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 4 * rg + r;
                if (row >= M) continue;
                const long o = (long)row * ld + n;
                const float tv = acc[t][r] + bv;
                if (tout) tout[o] = tv;
                out[o] = res[o] + g * tv;
            }
        }
    }
    static constexpr bool kRowEpilogue = true;
    // the residual rows of a 16-row fragment, loaded for all four row groups of the lane at once and -- by the callers -- one fragment
    // ahead of their use (read inside the row loop every row group paid its own memory round trip)
    static constexpr bool kPrefetchRows = true;
    // branch-free bodies for fragments inside the matrix, chosen once per tile by the kernels (see EpStore::fast_mode): 1 = no `tout`
    static constexpr int kFastModes = 1;
    __device__ __forceinline__ int fast_mode() const { return tout ? 0 : 1; }
    template <int MODE, int NT, class BL>
    __device__ __forceinline__ RowPre prefetch_full(const BL& bl, int row0, int nblk, int lane) const {
        RowPre r;
        const int c4 = lane & 15, q = lane >> 4;
        const int n = bl.col(nblk, 0, 0) + 4 * c4;
        if (c4 < NT * 4 && n < N) {
#pragma unroll
            for (int p = 0; p < 4; ++p) r.v[p] = ld4(res + (long)(row0 + q + 4 * p) * ld + n);
        }
        return r;
    }
    template <int MODE, int NT, class BL>
    __device__ __forceinline__ void run_rows_full(const float* so, int ldo, const BL& bl, int row0, int nblk, int lane, const RowPre& pre) const {
        const int c4 = lane & 15, q = lane >> 4;
        const int n = bl.col(nblk, 0, 0) + 4 * c4;
        if (c4 >= NT * 4 || n >= N) return;
        const f4 bv = bias ? ld4(bias + n) : zero4();
        const f4 g = gamma ? ld4(gamma + n) : f4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int lr = q + 4 * p;
            const f4 tv = *reinterpret_cast<const f4*>(so + lr * ldo + 4 * c4) + bv;
            *reinterpret_cast<f4*>(out + (long)(row0 + lr) * ld + n) = pre.v[p] + g * tv;
        }
    }
    template <int NT, class BL>
    __device__ __forceinline__ RowPre prefetch_rows(const BL& bl, int row0, int nblk, int lane, int M) const {
        RowPre r;
        const int c4 = lane & 15, q = lane >> 4;
        const int n = bl.col(nblk, 0, 0) + 4 * c4;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = row0 + q + 4 * p;
            r.v[p] = (c4 < NT * 4 && n < N && row < M) ? ld4(res + (long)row * ld + n) : zero4();
        }
        return r;
    }
    template <int NT, class BL>
    __device__ __forceinline__ void run_rows(const float* so, int ldo, const BL& bl, int row0, int nblk, int lane, int M) const {
        run_rows<NT, BL>(so, ldo, bl, row0, nblk, lane, M, prefetch_rows<NT, BL>(bl, row0, nblk, lane, M));
    }
    template <int NT, class BL>
    __device__ __forceinline__ void run_rows(const float* so, int ldo, const BL& bl, int row0, int nblk, int lane, int M, const RowPre& pre) const {
        const int c4 = lane & 15, q = lane >> 4;
        const int n = bl.col(nblk, 0, 0) + 4 * c4;
        if (c4 >= NT * 4 || n >= N) return;
        const f4 bv = bias ? ld4(bias + n) : zero4();
        const f4 g = gamma ? ld4(gamma + n) : f4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int lr = q + 4 * p, row = row0 + lr;
            if (row >= M) continue;
            const long o = (long)row * ld + n;
            const f4 tv = *reinterpret_cast<const f4*>(so + lr * ldo + 4 * c4) + bv;
            if (tout) *reinterpret_cast<f4*>(tout + o) = tv;
            *reinterpret_cast<f4*>(out + o) = pre.v[p] + g * tv;
        }
    }
};

struct EpLstm {                     // NT must be 4: tiles = (f, i, o, g) of channels nblk*16 + lane&15
    static constexpr bool kRowEpilogue = false;
    const float* bias; const float* c_prev; float* h_out; float* c_out; float* gates_out;  // gates_out [M][4][C] post-activation (optional)
    int C;
    template <int NT, class BL>
    __device__ __forceinline__ void run(f4 (&acc)[NT], const BL&, int row0, int nblk, int lane, int M) const {
        static_assert(NT == 4, "LSTM epilogue needs the four gate tiles");
        const int i = lane & 15, rg = lane >> 4;
        const int c = nblk * 16 + i;
        if (c >= C) return;
        const float bf = bias[c], bi = bias[C + c], bo = bias[2 * C + c], bg = bias[3 * C + c];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + 4 * rg + r;
            if (row >= M) continue;
            const float f = sigmoidf_(acc[0][r] + bf), ig = sigmoidf_(acc[1][r] + bi), o = sigmoidf_(acc[2][r] + bo);
            const float g = tanhf(acc[3][r] + bg);
            const long idx = (long)row * C + c;
            const float cp = c_prev ? c_prev[idx] : 0.f;
            const float cn = f * cp + ig * g;
            c_out[idx] = cn;
            h_out[idx] = o * tanhf(cn);
            if (gates_out) {
                float* gp = gates_out + (long)row * 4 * C + c;
                gp[0] = f; gp[C] = ig; gp[2 * C] = o; gp[3 * C] = g;
            }
        }
    }
};

// =================================================================================================
// row GEMM kernel
//   KS == 1: the 4 waves of a workgroup own 4 consecutive 16-row tiles (64 rows) and the whole K range.
//   KS == 4: small-M regime (RVT stages 3/4: a few hundred tokens, K up to 1536): the 4 waves share ONE 16-row
//            tile and split the K chunks round-robin, then reduce through LDS -- 4x more wavefronts in flight
//            and 4x shorter dependent MFMA chains, no atomics.
// Operand loads run two chunks ahead of the MFMAs (register ring of depth 2).
// =================================================================================================
template <int NT, int KS, int BF, class AL, class BL, class EP>
__global__ __launch_bounds__(256) void gemm16_kernel(AL al, BL bl, EP ep, int M, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int row0 = KS == 1 ? (blockIdx.x * 4 + wave) * 16 : blockIdx.x * 16;
    const int nblk = blockIdx.y;
    if (row0 >= M) return;                       // wave-uniform (block-uniform for KS > 1)
    typename AL::St st = al.init(row0 + i, M, lane, nblk == 0 && (KS == 1 || wave == 0));
    f4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = zero4();
    K = al.klen(st, K);                          // wave-uniform (parity-class conv dgrad shortens K per tile)
    const int aux = al.aux(st);
    const int KC = (K + 15) >> 4;
    const int kstep = KS == 1 ? 1 : KS;
    int kc = KS == 1 ? 0 : wave;
    f4 a0 = zero4(), a1 = zero4(), b0[NT], b1[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { b0[t] = zero4(); b1[t] = zero4(); }
    if (kc < KC) {
        a0 = al.load(st, kc * 16 + 4 * q, K);
#pragma unroll
        for (int t = 0; t < NT; ++t) b0[t] = bl.load(nblk, t, i, kc * 16 + 4 * q, K, aux);
    }
    if (kc + kstep < KC) {
        a1 = al.load(st, (kc + kstep) * 16 + 4 * q, K);
#pragma unroll
        for (int t = 0; t < NT; ++t) b1[t] = bl.load(nblk, t, i, (kc + kstep) * 16 + 4 * q, K, aux);
    }
    for (; kc < KC; kc += 2 * kstep) {
        {   // chunk kc from ring slot 0, refill slot 0 with chunk kc + 2*kstep
            const f4 ta = a0; f4 tb[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) tb[t] = b0[t];
            const int kn = kc + 2 * kstep;
            if (kn < KC) {
                a0 = al.load(st, kn * 16 + 4 * q, K);
#pragma unroll
                for (int t = 0; t < NT; ++t) b0[t] = bl.load(nblk, t, i, kn * 16 + 4 * q, K, aux);
            }
            if constexpr (BF) {
                Frag16<BF> fa; fa.set(ta);
#pragma unroll
                for (int t = 0; t < NT; ++t) { Frag16<BF> fb; fb.set(tb[t]); acc[t] = mfma_frag<BF>(fa, fb, acc[t]); }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = mfma16(ta[j], tb[t][j], acc[t]);
            }
        }
        if (kc + kstep < KC) {   // chunk kc + kstep from ring slot 1
            const f4 ta = a1; f4 tb[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) tb[t] = b1[t];
            const int kn = kc + 3 * kstep;
            if (kn < KC) {
                a1 = al.load(st, kn * 16 + 4 * q, K);
#pragma unroll
                for (int t = 0; t < NT; ++t) b1[t] = bl.load(nblk, t, i, kn * 16 + 4 * q, K, aux);
            }
            if constexpr (BF) {
                Frag16<BF> fa; fa.set(ta);
#pragma unroll
                for (int t = 0; t < NT; ++t) { Frag16<BF> fb; fb.set(tb[t]); acc[t] = mfma_frag<BF>(fa, fb, acc[t]); }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = mfma16(ta[j], tb[t][j], acc[t]);
            }
        }
    }
    if (KS > 1) {
        __shared__ float red[KS > 1 ? 3 : 1][NT * 256];
        if (wave > 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave - 1][(t * 4 + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = (t * 4 + r) * 64 + lane;
                acc[t][r] += red[0][o] + red[1][o] + red[2][o];
            }
    }
    ep.template run<NT, BL>(acc, bl, row0, nblk, lane, M);
}

template <int NT, class AL, class BL, class EP>
static inline int launch_gemm16(const AL& al, const BL& bl, const EP& ep, int M, int K, int nblocks_n, hipStream_t s) {
    if (M <= 0) return LEOD_OK;
    // fewer than ~2 workgroups per CU and a long K loop: split K across the 4 waves of each workgroup
    const bool ksplit = (long)cdiv(M, 64) * nblocks_n < 512 && K >= 128;
    if (ksplit) {
        dim3 grid(cdiv(M, 16), nblocks_n);
        LEOD_BY_OPFMT_IF(bl_fwd<BL>::value, hipLaunchKernelGGL((gemm16_kernel<NT, 4, OF, AL, BL, EP>), grid, dim3(256), 0, s, al, bl, ep, M, K));
    } else {
        dim3 grid(cdiv(M, 64), nblocks_n);
        LEOD_BY_OPFMT_IF(bl_fwd<BL>::value, hipLaunchKernelGGL((gemm16_kernel<NT, 1, OF, AL, BL, EP>), grid, dim3(256), 0, s, al, bl, ep, M, K));
    }
    return leod_launch_status();
}

// =================================================================================================
// LDS-staged row GEMM (large M):  64 rows x NT*16 columns per workgroup, K swept in chunks of KCH.
// Why: the register-direct kernels above load MFMA operands in operand layout, i.e. one wave instruction gathers
// 16 rows x 64 B.  Measured on MI355X that gather pattern (16 partial cache lines per instruction, 4x the line
// requests of a contiguous stream) caps those kernels at ~1 TB/s, whereas row-contiguous 1 KiB wave loads stream at
// 2.5-3 TB/s.  Here both operands are fetched with fully coalesced 16-byte loads (each thread owns fixed (row, k4)
// slots), double-buffered through LDS, and read back in operand layout with conflict-free ds_read_b128.
// =================================================================================================
// RW = 16-row fragments per wave (workgroup = 64*RW rows): RW = 2 halves the B-fragment LDS reads and the B staging per
// output row -- the MFMA-bound shapes (RVT stages 3/4, 3x3 convs) were LDS-bandwidth limited at RW = 1 (every wave
// re-reads the whole B tile: 5 ds_read_b128 per 16 MFMAs, x 16 resident waves per CU > 128 B/clk).
// BF = true (precision mode bf16): both operands are rounded to bf16 when they are stashed, LDS holds bf16 (half the bytes, half
// the fragment-read traffic) and one v_mfma_f32_16x16x16_bf16 replaces the four fp32 MFMAs of a 16-k chunk.  Row-major operand tiles
// keep rows of KCH + 8 bf16: a row stride of 4 * odd dwords puts the 32 fragment starts of a ds_read_b64 lane group on 32
// distinct bank pairs.  Transposed weights (dgrad: W[k][n], n contiguous) are stored as [16 k][16 n] blocks in their natural
// orientation (8-byte stores of 4 n; block stride 512 + 32 bytes keeps the 16-lane store groups conflict-free) and read back with
// ds_read_b64_tr_b16, whose 16-lane groups return the [4 k][16 n] block column-wise: lane (i, q) gets W[4q..4q+3][n = i].
template <int NT, int KCH, int NBUF, int RW, int BF, class AL, class BL, class EP>
__global__ __launch_bounds__(256, NBUF == 1 ? (RW == 1 ? 4 : 3) : 2) void gemm_lds_kernel(AL al, BL bl, EP ep, int M, int K, int nblocks_n) {
    constexpr int BM = 64 * RW;
    // ds_read_b128 is serviced in 4 groups of 16 lanes, {0-3,12-15,20-27}, ...: rows {0-3,12-15} at k-offset 4q and rows
    // {4-11} at 4(q+1) share a group.  A row stride == 8 (mod 16) dwords puts the 16 starts on 16 distinct 4-bank slots
    // (conflict-free, 4 LDS cycles); the KCH + 4 used first was 2-way conflicted (PMC: 39 % of LDS cycles were conflicts)
    constexpr int LD = KCH + 8;
    constexpr int K4 = KCH / 4;                      // float4 slots per staged row
    constexpr int BN = NT * 16;
    constexpr int RA = (BM * K4 + 255) / 256, RB = (BN * K4 + 255) / 256;
    // transposed weights (dgrad: W is [k][n], n contiguous) stay in their natural [k][n] layout in LDS: 16-byte stores,
    // B fragments by 4 x ds_read_b32 (rows 4q+j land in distinct 16-bank halves when LDN == 4 mod 8).  Transposing on
    // the way in cost 4 scalar stores per float4 with 8- to 16-way bank conflicts.
    constexpr int LDN = BN + 4;
    constexpr int BST = 16 * 16 + 16;                // bf16 elements per [16 k][16 n] block of transposed weights (+ 32 bytes)
    // operand buffer sizes in FLOAT units (bf16 tiles are counted in pairs)
    constexpr int ASZ = BF ? (BM * LD) / 2 : BM * LD;
    constexpr int BSZ = BF ? (BL::kTrans ? ((KCH / 16) * NT * BST) / 2 : (BN * LD) / 2) : (BL::kTrans ? KCH * LDN : BN * LD);
    constexpr int LDO = 64;                          // accumulator transposition tile of the row-layout epilogue (aliases A/B);
                                                     // 256-byte rows: the (row q, 16-byte column c4) reads are conflict-free
    constexpr int OPS = NBUF * (ASZ + BSZ);
    constexpr int SMEM = OPS >= 64 * LDO ? OPS : 64 * LDO;      // the epilogue tile must fit in the operand buffers
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float (*sA)[ASZ] = reinterpret_cast<float (*)[ASZ]>(smem);
    float (*sB)[BSZ] = reinterpret_cast<float (*)[BSZ]>(smem + NBUF * ASZ);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    // XCD-aware 1-D grid: consecutive workgroup ids go round-robin to the 8 XCDs (each with its own L2).  All n-blocks of
    // one 64-row block get ids with the same (id % 8) and adjacent dispatch slots, so the A tile is re-read from that XCD's
    // L2 and the pieces of an output row are written close together (merged into full lines in L2).
    const int per = 8 * nblocks_n;
    const int grp = blockIdx.x / per, rem = blockIdx.x - grp * per;
    const int nblk = rem >> 3;
    const int brow0 = (grp * 8 + (rem & 7)) * BM;
    if (brow0 >= M) return;                          // workgroup-uniform (padding of the last group)
    // ---- fixed staging slots of this thread ------------------------------------------------------------------------
    typename AL::St ast[RA];
    int ak[RA], al_off[RA]; bool aok[RA];
#pragma unroll
    for (int p = 0; p < RA; ++p) {
        const int e = tid + 256 * p, r = e / K4, k4 = (e - r * K4) * 4;
        aok[p] = e < BM * K4;
        ast[p] = al.init(brow0 + (aok[p] ? r : 0), M, 0, false);
        ak[p] = k4; al_off[p] = r * LD + k4;        // element offset (fp32 or bf16 elements alike)
        // 256 % K4 == 0: every slot of a thread has the same k offset -> say so, and the k -> (tap, channel) decode of the
        // im2col / stem loaders is computed once per chunk instead of once per slot
        if constexpr (256 % K4 == 0) ak[p] = ak[0];
    }
    int bk[RB], bn[RB], bl_off[RB]; bool bok[RB];
#pragma unroll
    for (int p = 0; p < RB; ++p) {
        const int e = tid + 256 * p;
        bok[p] = e < BN * K4;
        if (!BL::kTrans) { const int nl = e / K4, k4 = (e - nl * K4) * 4; bn[p] = nl; bk[p] = k4; bl_off[p] = nl * LD + k4; }
        else {
            const int kl = e / (BN / 4), n4 = (e - kl * (BN / 4)) * 4; bn[p] = n4; bk[p] = kl;
            bl_off[p] = BF ? ((kl >> 4) * NT + (n4 >> 4)) * BST + (kl & 15) * 16 + (n4 & 15) : kl * LDN + n4;
        }
    }
    // parity-class conv dgrad: the live-tap count (hence K) depends on the class of the rows; the caller guarantees
    // that a 64-row workgroup never mixes classes, so both are workgroup-uniform
    K = al.klen(ast[0], K);
    const int aux = al.aux(ast[0]);
    f4 ra[RA], rb[RB];
    constexpr bool ATP = a_two_phase<AL>::value;
    typename a_two_phase<AL>::Raw raw_a[RA];                  // two-phase A loaders: raw registers of the next chunk
    auto fetch = [&](int k0) {
        if constexpr (ATP) {
#pragma unroll
            for (int p = 0; p < RA; ++p) al.raw(ast[p], min(k0 + ak[p], K - 4), raw_a[p]);     // unconditional, clamped: all in flight together
        } else {
#pragma unroll
        for (int p = 0; p < RA; ++p) ra[p] = aok[p] ? al.template load2<KCH>(ast[p], k0, ak[p], K) : zero4();
        }
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            rb[p] = zero4();
            if (bok[p]) {
                if constexpr (BF && bl_is16<BL>::value) {   // four bf16 of the shadow, carried in rb[p].xy
                    if constexpr (!BL::kTrans) rb[p] = shadow_bits(bl.load16(nblk, bn[p] >> 4, bn[p] & 15, k0 + bk[p], K));
                    else rb[p] = shadow_bits(bl.load_n4_16(nblk, bn[p], k0 + bk[p], K));
                } else if (!BL::kTrans) rb[p] = bl.template load2<KCH>(nblk, bn[p] >> 4, bn[p] & 15, k0, bk[p], K, aux);
                else rb[p] = bl.load_n4(nblk, bn[p], k0 + bk[p], K);
            }
        }
    };
    auto finish = [&](int k0) {                               // raw registers -> operand values (two-phase A loaders)
        if constexpr (ATP) {
#pragma unroll
            for (int p = 0; p < RA; ++p) {
                const f4 v = al.fin(ast[p], raw_a[p]);
                ra[p] = (aok[p] && ast[p].ok && k0 + ak[p] < K) ? v : zero4();
            }
        }
    };
    auto stash = [&](int buf) {
        if constexpr (BF) {
            unsigned short* __restrict__ a16 = reinterpret_cast<unsigned short*>(sA[buf]);
            unsigned short* __restrict__ b16 = reinterpret_cast<unsigned short*>(sB[buf]);
#pragma unroll
            for (int p = 0; p < RA; ++p) if (aok[p]) *reinterpret_cast<s4*>(a16 + al_off[p]) = pack16<BF>(ra[p]);
#pragma unroll
            for (int p = 0; p < RB; ++p) if (bok[p]) *reinterpret_cast<s4*>(b16 + bl_off[p]) = bl_is16<BL>::value ? shadow_s4(rb[p]) : pack16_raw<BF>(rb[p]);
        } else {
#pragma unroll
            for (int p = 0; p < RA; ++p) if (aok[p]) *reinterpret_cast<f4*>(&sA[buf][al_off[p]]) = ra[p];
#pragma unroll
            for (int p = 0; p < RB; ++p) if (bok[p]) {
                *reinterpret_cast<f4*>(&sB[buf][bl_off[p]]) = rb[p];
            }
        }
    };
    f4 acc[RW][NT];
#pragma unroll
    for (int w = 0; w < RW; ++w)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[w][t] = zero4();
    const int nch = (K + KCH - 1) / KCH;
    fetch(0);
    finish(0);
    stash(0);
    __syncthreads();
    const int aoff = (16 * RW * wave + i) * LD + 4 * q;                            // wave owns rows 16*RW*wave ..
    const int boff = BL::kTrans ? (4 * q) * LDN + i : i * LD + 4 * q;
    for (int ch = 0; ch < nch; ++ch) {
        const int buf = NBUF == 1 ? 0 : (ch & 1);
        const bool more = ch + 1 < nch;
        if (more) fetch((ch + 1) * KCH);                      // next chunk's global loads fly under this chunk's MFMAs
        if constexpr (BF) {
            typedef __attribute__((address_space(3))) s4 lds_s4;
            const unsigned short* __restrict__ pa = reinterpret_cast<const unsigned short*>(sA[buf]) + aoff;
            const unsigned short* __restrict__ pb = reinterpret_cast<const unsigned short*>(sB[buf]) +
                                                    (BL::kTrans ? (4 * q + (i >> 2)) * 16 + 4 * (i & 3) : boff);
#pragma unroll
            for (int c = 0; c < KCH / 16; ++c) {
                s4 av[RW];
#pragma unroll
                for (int w = 0; w < RW; ++w) av[w] = *reinterpret_cast<const s4*>(pa + 16 * w * LD + 16 * c);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    s4 bv;
                    if constexpr (BL::kTrans) bv = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(pb + (c * NT + t) * BST));
                    else bv = *reinterpret_cast<const s4*>(pb + 16 * t * LD + 16 * c);
#pragma unroll
                    for (int w = 0; w < RW; ++w) acc[w][t] = mfma16_16<BF>(av[w], bv, acc[w][t]);
                }
            }
        } else {
        const float* __restrict__ pa = sA[buf] + aoff;
        const float* __restrict__ pb = sB[buf] + boff;
#pragma unroll
        for (int c = 0; c < KCH / 16; ++c) {
            f4 av[RW];
#pragma unroll
            for (int w = 0; w < RW; ++w) av[w] = *reinterpret_cast<const f4*>(pa + 16 * w * LD + 16 * c);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f4 bv;
                if constexpr (BL::kTrans) {
                    const float* pt = pb + (16 * c) * LDN + 16 * t;
                    bv.x = pt[0]; bv.y = pt[LDN]; bv.z = pt[2 * LDN]; bv.w = pt[3 * LDN];
                } else {
                    bv = *reinterpret_cast<const f4*>(pb + 16 * t * LD + 16 * c);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int w = 0; w < RW; ++w) acc[w][t] = mfma16(av[w][j], bv[j], acc[w][t]);
            }
        }
        }
        if (NBUF > 1) {
            if (more) { finish((ch + 1) * KCH); stash(buf ^ 1); }
            __syncthreads();
        } else if (more) {                                    // single LDS buffer (half the LDS -> twice the resident
            finish((ch + 1) * KCH);
            __syncthreads();                                  // workgroups): everyone done reading, then refill
            stash(0);
            __syncthreads();
        }
    }
    if constexpr (EP::kRowEpilogue) {
        float* so = smem + wave * 16 * LDO;                   // wave-private 16 x BN tile
        // one body per tile: mode 0 = the generic row epilogue (edge tiles, rare options), modes >= 1 = the epilogue's branch-free
        // bodies (EP::fast_mode()), picked OUTSIDE the fragment loop so that no fragment ends at a join with the generic code
        auto ep_tile = [&](auto modec) {
            constexpr int MODE = decltype(modec)::value;
            RowPre pre;
            if constexpr (MODE > 0) pre = ep.template prefetch_full<MODE, NT, BL>(bl, brow0 + 16 * (RW * wave), nblk, lane);
            else if constexpr (EP::kPrefetchRows) pre = ep.template prefetch_rows<NT, BL>(bl, brow0 + 16 * (RW * wave), nblk, lane, M);
#pragma unroll
            for (int w = 0; w < RW; ++w) {
                const int row0 = brow0 + 16 * (RW * wave + w);
                const RowPre cur = pre;
                if constexpr (MODE > 0) { if (w + 1 < RW) pre = ep.template prefetch_full<MODE, NT, BL>(bl, row0 + 16, nblk, lane); }
                else if constexpr (EP::kPrefetchRows) { if (w + 1 < RW) pre = ep.template prefetch_rows<NT, BL>(bl, row0 + 16, nblk, lane, M); }
                __syncthreads();                              // operand buffers (w = 0) / previous tile (w > 0) are done with
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) so[(4 * q + r) * LDO + 16 * t + i] = acc[w][t][r];
                __syncthreads();
                if constexpr (MODE > 0) ep.template run_rows_full<MODE, NT, BL>(so, LDO, bl, row0, nblk, lane, cur);
                else if (row0 < M) {
                    if constexpr (EP::kPrefetchRows) ep.template run_rows<NT, BL>(so, LDO, bl, row0, nblk, lane, M, cur);
                    else ep.template run_rows<NT, BL>(so, LDO, bl, row0, nblk, lane, M);
                }
            }
        };
        const int fm = brow0 + BM <= M ? ep.fast_mode() : 0;  // workgroup-uniform
        dispatch_fast_mode<EP::kFastModes>(fm, ep_tile);
    } else {
#pragma unroll
        for (int w = 0; w < RW; ++w) {
            const int row0 = brow0 + 16 * (RW * wave + w);
            if (row0 < M) ep.template run<NT, BL>(acc[w], bl, row0, nblk, lane, M);
        }
    }
}

// (mean, rstd) of every row of x[M,K] -> stats[M,2]; 16 lanes per row, fully coalesced (the LN prologue of the
// LDS-staged GEMM reads them instead of re-deriving the statistics per n-block).
static __global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ x, long ld, float* __restrict__ stats,
                                                        int M, int K, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, rg = lane >> 4;
    const long row = ((long)blockIdx.x * 4 + wave) * 4 + rg;
    const bool rok = row < M;
    const float* p = x + (rok ? row : 0) * ld;
    float sum = 0.f;
    for (int k = 4 * i; k < K; k += 64) { const f4 v = ld4(p + k); sum += (v.x + v.y) + (v.z + v.w); }
    const float mean = row16_sum(sum) / (float)K;
    float var = 0.f;
    for (int k = 4 * i; k < K; k += 64) {
        const f4 v = ld4(p + k);
        const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
        var += (a * a + b * b) + (c * c + d * d);
    }
    const float rstd = rsqrtf(row16_sum(var) / (float)K + eps);
    if (rok && i == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}
static inline int launch_row_stats(const float* x, long ld, float* stats, int M, int K, float eps, hipStream_t s) {
    hipLaunchKernelGGL(row_stats_kernel, dim3(cdiv(M, 16)), dim3(256), 0, s, x, ld, stats, M, K, eps);
    return leod_launch_status();
}

// host side: the loader type that reads the bf16 shadow, if the weights lie in a registered buffer with a fresh shadow (precision mode bf16;
// only in translation units that ask for the extra instantiations: LEOD_SHADOW_KERNELS, the Linear layers of k_linear.hip)
template <class BL> static inline const unsigned short* bl_shadow_ptr(const BL& bl) {
#ifdef LEOD_SHADOW_KERNELS
    if constexpr (!std::is_void<typename bl_shadow_type<BL>::type>::value) {
        if (leod_precision() == 1 && !(bl.ld & 3)) return leod_shadow_of(bl.w, leod_opfmt());
    }
#endif
    return nullptr;
}
template <class BL> static inline typename bl_shadow_type<BL>::type bl_as16(const BL& bl, const unsigned short* sh) {
    typename bl_shadow_type<BL>::type b;
    static_cast<BL&>(b) = bl;
    b.w16 = sh;
    return b;
}

template <int NT, int RW, class AL, class BL, class EP>
static inline int launch_gemm_lds_rw(const AL& al, const BL& bl, const EP& ep, int M, int K, int nblocks_n, hipStream_t s) {
#ifdef LEOD_SHADOW_KERNELS
    if constexpr (!std::is_void<typename bl_shadow_type<BL>::type>::value) {
        if (const unsigned short* sh = bl_shadow_ptr(bl)) return launch_gemm_lds_rw<NT, RW>(al, bl_as16(bl, sh), ep, M, K, nblocks_n, s);
    }
#endif
    dim3 grid(cdiv(cdiv(M, 64 * RW), 8) * 8 * nblocks_n);
    // single LDS buffer + register prefetch everywhere: residency (3-6 workgroups per CU) hides the two barriers per chunk
    // better than a double buffer at 2-3 workgroups per CU does (measured)
    static const int nbuf = 1;
    (void)nbuf;                                      // the double-buffered variant is no longer instantiated (never faster, see above)
    // (96-wide chunks in the 16-bit modes, measured: 38.2 vs 34.2 ms per step on -> off; not instantiated)
    if (K % 48 == 0) {
        LEOD_BY_OPFMT_IF(bl_fwd<BL>::value, hipLaunchKernelGGL((gemm_lds_kernel<NT, 48, 1, RW, OF, AL, BL, EP>), grid, dim3(256), 0, s, al, bl, ep, M, K, nblocks_n));
    } else {
        LEOD_BY_OPFMT_IF(bl_fwd<BL>::value, hipLaunchKernelGGL((gemm_lds_kernel<NT, 64, 1, RW, OF, AL, BL, EP>), grid, dim3(256), 0, s, al, bl, ep, M, K, nblocks_n));
    }
    return leod_launch_status();
}
template <int NT, class AL, class BL, class EP>
static inline int launch_gemm_lds(const AL& al, const BL& bl, const EP& ep, int M, int K, int nblocks_n, hipStream_t s) {
    // RW = 2 (128-row workgroups, two row fragments per wave) measured 5-40 % SLOWER on every RVT shape (3 instead of 4
    // resident workgroups, two epilogue rounds), so only RW = 1 is instantiated.
    return launch_gemm_lds_rw<NT, 1>(al, bl, ep, M, K, nblocks_n, s);
}
#include "gemm_bf16.hpp"
// plain-row A operands: the hot mode combinations of the wide GEMMs (NT >= 3) run on the two-phase loader ALRowsM
template <int NT, class BL, class EP>
static inline int launch_gemm_lds(const ALRows& al, const BL& bl, const EP& ep, int M, int K, int nblocks_n, hipStream_t s) {
    static const int two_phase = 1;
    // precision mode bf16, Linear forward / dgrad with >= 144 output columns: the 128 x 192 / 256 wide-tile kernel (gemm_bf16.hpp)
    if constexpr ((std::is_same<BL, BLRows>::value || std::is_same<BL, BLTrans>::value) &&
                  (std::is_same<EP, EpStore>::value || std::is_same<EP, EpLsRes>::value)) {
        const bool ln = al.ln_w != nullptr, ks = al.kscale != nullptr;
        // the persistent wide tiles win where the main loop dominates (LayerNorm on load, contractions of >= 2 N); short contractions
        // into wide outputs are bound by their epilogue traffic, which the many small workgroups of gemm_lds_kernel overlap better
        // (tools/kbench_gemm.py; wide_mode = 2 would route every covered shape here)
        constexpr int wide_mode = 1;
        // (round 5, tools/kbench_gemm.py graph-timed, profiles/r05_j_gemm_wide_routing_ab.txt) launches of <= 60 k rows (stages 3-4): also the
        // square projections (proj + LayerScale, its dgrad: 24 -> 16-22 us) and the plain x projection of the ConvLSTM (85 / 75 -> 70 / 67 us); NOT
        // the dgrad of fc2 through GELU (short contraction, wide output, heavy epilogue: 134 -> 191 us on the wide tiles)
        bool small_ok = M <= 60000 && (K >= bl.N || (std::is_same<BL, BLRows>::value && std::is_same<EP, EpStore>::value));
        if constexpr (std::is_same<EP, EpStore>::value) small_ok = small_ok && ep.act != ACT_MUL_GELU_GRAD;
        const int ntw = ((!ln || al.stats_in) && (wide_mode >= 2 || ln || K >= 2 * bl.N || small_ok)) ? gemm_wide_ntw(M, bl.N, K) : 0;
        if (ntw) {
#define LEOD_WIDE(F, L, S) { ALRowsM<F, L, S> am; static_cast<ALRows&>(am) = al;                                                     \
                             return ntw == 3 ? launch_gemm_wide<3>(am, bl, ep, M, K, bl.N, s) : launch_gemm_wide<4>(am, bl, ep, M, K, bl.N, s); }
            constexpr bool rows = std::is_same<BL, BLRows>::value, store = std::is_same<EP, EpStore>::value;
            if constexpr (rows && store) {                  // LN -> qkv / fc1, plain x projection of the ConvLSTM
                if (al.fmt == 0 && !ln && !ks) LEOD_WIDE(0, false, false)
                if (al.fmt == 0 && ln && !ks) LEOD_WIDE(0, true, false)
            } else if constexpr (rows && !store) {          // proj / fc2 + LayerScale + residual (fc2: gelu of the fp16 pre-activation)
                if (al.fmt == 0 && !ln && !ks) LEOD_WIDE(0, false, false)
                if (al.fmt == 1 && !ln && !ks) LEOD_WIDE(1, false, false)
                if (al.fmt == 3 && !ln && !ks) LEOD_WIDE(3, false, false)
            } else if constexpr (!rows && store) {          // dgrads: fp32 / bf16 gradient rows, optional LayerScale factor
                if (al.fmt == 0 && !ln && !ks) LEOD_WIDE(0, false, false)
                if (al.fmt == 0 && !ln && ks) LEOD_WIDE(0, false, true)
                if (al.fmt == 2 && !ln && !ks) LEOD_WIDE(2, false, false)
            }
#undef LEOD_WIDE
        }
    }
    if constexpr (NT >= 3) {
        const bool ln = al.ln_w != nullptr, ks = al.kscale != nullptr;
        if (two_phase && !(K & 3) && K >= 4 && (!ln || al.stats_in)) {
#define LEOD_ALM(F, L, S) { ALRowsM<F, L, S> am; static_cast<ALRows&>(am) = al; return launch_gemm_lds_rw<NT, 1>(am, bl, ep, M, K, nblocks_n, s); }
            if (al.fmt == 0 && !ln && !ks) LEOD_ALM(0, false, false)
            if (al.fmt == 0 && ln && !ks) LEOD_ALM(0, true, false)
            if (al.fmt == 0 && !ln && ks) LEOD_ALM(0, false, true)
            if (al.fmt == 2 && !ln && !ks) LEOD_ALM(2, false, false)
            if (al.fmt == 2 && !ln && ks) LEOD_ALM(2, false, true)
            if (al.fmt == 1 && !ln && !ks) LEOD_ALM(1, false, false)
            if (al.fmt == 3 && !ln && !ks) LEOD_ALM(3, false, false)
#undef LEOD_ALM
        }
    }
    return launch_gemm_lds_rw<NT, 1>(al, bl, ep, M, K, nblocks_n, s);
}
// enough 64-row workgroups to fill the chip; smaller problems stay on the register-direct kernels (K-split)
static inline bool use_gemm_lds(int M, int nblocks_n) { return (long)cdiv(M, 64) * nblocks_n >= 256 && M >= 2048; }

// =================================================================================================
// wgrad GEMM:  dW[n][k] += sum_m dY(m,n) * X(m,k)   (+ optional dbias[n] += sum_m dY(m,n))
// Each workgroup owns `rows_per_block` rows of M and one (TN*16 x TK*16) tile of dW; its 4 waves
// interleave 16-row chunks, reduce through LDS and issue one fp32 atomicAdd per dW element.
// =================================================================================================
struct XRows {                      // X(m,k) = x[m][k], optional LayerNorm with saved (mean, rstd)
    const float* x; long ld; const float* stats; const float* ln_w; const float* ln_b;
    const float* x2; long ld2; int K1;          // optional concat source for k >= K1
    int fmt;                                    // 1: x is an fp16 pre-activation, X = gelu(x) (see ALRows::fmt); 2: x holds bf16 rows; 3: fp16 rows
    __device__ __forceinline__ float get(int m, int k) const {
        if (fmt == 1) return gelu_erf(unpack_h16_1(reinterpret_cast<const unsigned short*>(x)[(long)m * ld + k]));
        if (x2 && k >= K1) return x2[(long)m * ld2 + (k - K1)];
        float v = x[(long)m * ld + k];
        if (stats) v = (v - stats[2 * (long)m]) * stats[2 * (long)m + 1] * ln_w[k] + ln_b[k];
        return v;
    }
    __device__ __forceinline__ f4 get4(int m, int k) const {      // k % 4 == 0; K1 % 4 == 0
        if (fmt == 1) {
            f4 v = unpack_h16(*reinterpret_cast<const s4*>(reinterpret_cast<const unsigned short*>(x) + (long)m * ld + k));
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
            return v;
        }
        if (x2 && k >= K1) return ld4(x2 + (long)m * ld2 + (k - K1));
        f4 v = ld4(x + (long)m * ld + k);
        if (stats) v = (v - stats[2 * (long)m]) * stats[2 * (long)m + 1] * ld4(ln_w + k) + ld4(ln_b + k);
        return v;
    }
    // Two-phase access for the pipelined weight-gradient kernel: raw4 only ISSUES the loads (row data, and the row's (mean, rstd)
    // in LayerNorm mode) -- nothing consumes a loaded value here, so all loads of a chunk are in flight together; fin4 applies
    // get4's arithmetic later (same operations in the same order: bit-identical values).  The caller clamps (m, k) to valid
    // coordinates and masks the result.
    // XM: 0 plain rows (optionally [x | x2]), 1 LayerNorm(x) with saved statistics, 2 gelu(fp16 pre-activation) -- chosen once per
    // launch (x_mode), so the loop has no mode branches
    template <int XM> __device__ __forceinline__ void raw4(int m, int k, f4& v, u2_& h, f2_& st) const {
        if constexpr (XM == 2) { h = *reinterpret_cast<const u2_*>(reinterpret_cast<const unsigned short*>(x) + (long)m * ld + k); return; }
        if constexpr (XM == 1) st = *reinterpret_cast<const f2_*>(stats + 2 * (long)m);
        const float* p = (x2 && k >= K1) ? x2 + (long)m * ld2 + (k - K1) : x + (long)m * ld + k;
        v = ld4(p);
    }
    template <int XM> __device__ __forceinline__ f4 fin4(f4 v, u2_ h, f2_ st, int k, f4 g, f4 b) const {
        if constexpr (XM == 2) {
            f4 o = unpack_h16(__builtin_bit_cast(s4, h));
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = gelu_erf(o[j]);
            return o;
        }
        if constexpr (XM == 1) { const f4 n = (v - st.x) * st.y * g + b; return (x2 && k >= K1) ? v : n; }
        return v;
    }
    int x_mode() const { return fmt == 3 ? 4 : fmt == 2 ? 3 : fmt == 1 ? 2 : (stats ? 1 : 0); }    // 3 / 4: bf16 / fp16 rows (wgrad_wide_bf16_kernel only)
    __device__ __forceinline__ long waddr(int n, int k, long ldw) const { return (long)n * ldw + k; }
};
// loaders with the raw4 / fin4 pair (see XRows) are staged in two phases by wgradw_kernel
template <class XL, class = void> struct x_two_phase { static constexpr bool value = false; };
template <class XL> struct x_two_phase<XL, decltype((void)&XL::x_mode)> { static constexpr bool value = true; };
struct XConvNHWC {                  // im2col of an NHWC map; k' = tap*Cin + c ; dW laid out [N][Cin][ks][ks]
    const float* x; int H, W, Cin, Ho, Wo, ks, stride, pad;
    __device__ __forceinline__ float get(int m, int k) const {
        const int ox = m % Wo, t = m / Wo; const int oy = t % Ho, b = t / Ho;
        const int tap = k / Cin, c = k - tap * Cin; const int kh = tap / ks, kw = tap - kh * ks;
        const int iy = oy * stride - pad + kh, ix = ox * stride - pad + kw;
        if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) return 0.f;
        return x[(((long)b * H + iy) * W + ix) * Cin + c];
    }
    __device__ __forceinline__ f4 get4(int m, int k) const {      // 4 consecutive channels of one tap (Cin % 4 == 0)
        const int ox = m % Wo, t = m / Wo; const int oy = t % Ho, b = t / Ho;
        const int tap = k / Cin, c = k - tap * Cin; const int kh = tap / ks, kw = tap - kh * ks;
        const int iy = oy * stride - pad + kh, ix = ox * stride - pad + kw;
        if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) return zero4();
        return ld4(x + (((long)b * H + iy) * W + ix) * Cin + c);
    }
    __device__ __forceinline__ long waddr(int n, int k, long) const {
        const int tap = k / Cin, c = k - tap * Cin;
        return ((long)n * Cin + c) * (ks * ks) + tap;
    }
};
template <typename T>
struct XStemNCHW {
    const T* x; int Cin, H, W, Ho, Wo, ks, stride, pad;
    __device__ __forceinline__ float get(int m, int k) const {
        const int ox = m % Wo, t = m / Wo; const int oy = t % Ho, b = t / Ho;
        int c, kh, kw;                                                                              // 7x7: constant divisors (see ALStemNCHW)
        if (ks == 7) { c = k / 49; const int r = k - c * 49; kh = r / 7; kw = r - kh * 7; }
        else { const int kk = ks * ks; c = k / kk; const int r = k - c * kk; kh = r / ks; kw = r - kh * ks; }
        const int iy = oy * stride - pad + kh, ix = ox * stride - pad + kw;
        if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) return 0.f;
        return (float)x[(((long)b * Cin + c) * H + iy) * W + ix];
    }
    __device__ __forceinline__ f4 get4(int m, int k) const { f4 v; v.x = get(m, k); v.y = get(m, k + 1); v.z = get(m, k + 2); v.w = get(m, k + 3); return v; }
    __device__ __forceinline__ long waddr(int n, int k, long ldw) const { return (long)n * ldw + k; }
};

// LDS-transposing wgrad: dY and X row chunks are fetched with coalesced 16-byte loads (each element once per
// workgroup) and stored TRANSPOSED in LDS ([column][row], rows contiguous).  The contraction runs over rows, and with
// the k-permutation "MFMA j of a 16-row step uses rows 4q+j" one ds_read_b128 per operand feeds four MFMAs.  The 4 waves
// split the TN*TK output tiles, keep them in registers over the workgroup's whole row range and finish with one fp32
// atomic per dW element (dW accumulates over timesteps and row splits).
// DYF: dY rows are fp32 (0) / bf16 (1) -- fixed per instantiation, a run-time test put a wait behind every dY load
template <int TN, int TK, bool BF, class XL, int DYF = 0>
__global__ __launch_bounds__(256) void wgrad16_kernel(const float* __restrict__ dy, long lddy, XL xl, float* dW, long ldw,
                                                      float* dbias, int M, int N, int K, int rows_per_block, int dyfmt) {
    constexpr int RC = 32;                                  // rows per staged chunk
    constexpr int LDN = 16 * TN + 4, LDK = 16 * TK + 4;    // natural [row][col] LDS tiles, see wgradw_kernel
    constexpr int BST = 16 * 16 + 16;                       // BF: bf16 [16 row][16 col] blocks + transpose reads, see wgradw_kernel
    constexpr int C4N = TN * 4, C4K = TK * 4;               // float4 slots per staged row
    constexpr int NV = C4N * RC, KV = C4K * RC;
    constexpr int RN = (NV + 255) / 256, RK = (KV + 255) / 256;
    constexpr int NTILE = TN * TK, TPW = (NTILE + 3) / 4;   // tiles per wave
    __shared__ __attribute__((aligned(16))) float sdy[2][BF ? ((RC / 16) * TN * BST) / 2 : RC * LDN];
    __shared__ __attribute__((aligned(16))) float sx[2][BF ? ((RC / 16) * TK * BST) / 2 : RC * LDK];
    __shared__ float sbias[16 * TN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.y * TN * 16, k0 = blockIdx.z * TK * 16;
    const int mbeg = blockIdx.x * rows_per_block;
    const int mend = min(M, mbeg + rows_per_block);
    const bool do_bias = dbias != nullptr && blockIdx.z == 0;
    // ---- chunk-invariant slot geometry / LDS offsets (hoisted: the kernel is VALU-issue sensitive) ----------------------
    int nr[RN], kr[RK], nl[RN], kl[RK], kc[RK];
    long np[RN];                                            // element offsets into dy (fp32, or bf16 when dyfmt)
    bool nok[RN], kok[RK];
#pragma unroll
    for (int e = 0; e < RN; ++e) {
        const int s = tid + 256 * e, r = s / C4N, c = (s - r * C4N) * 4;
        nr[e] = r; nl[e] = BF ? ((r >> 4) * TN + (c >> 4)) * BST + (r & 15) * 16 + (c & 15) : r * LDN + c; nok[e] = s < NV && n0 + c < N;
        np[e] = (long)(mbeg + r) * lddy + n0 + c;
    }
#pragma unroll
    for (int e = 0; e < RK; ++e) {
        const int s = tid + 256 * e, r = s / C4K, c = (s - r * C4K) * 4;
        kr[e] = r; kl[e] = BF ? ((r >> 4) * TK + (c >> 4)) * BST + (r & 15) * 16 + (c & 15) : r * LDK + c; kc[e] = k0 + c; kok[e] = s < KV && k0 + c < K;
    }
    int offA[TPW], offB[TPW]; bool tok[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tile = wave + 4 * t;
        tok[t] = tile < NTILE;
        const int a = tok[t] ? tile / TK : 0, b = tok[t] ? tile - a * TK : 0;
        offA[t] = BF ? a * BST + (4 * q + (i >> 2)) * 16 + 4 * (i & 3) : (4 * q) * LDN + 16 * a + i;
        offB[t] = BF ? b * BST + (4 * q + (i >> 2)) * 16 + 4 * (i & 3) : (4 * q) * LDK + 16 * b + i;
    }
    f4 acc[TPW], bacc[RN];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = zero4();
#pragma unroll
    for (int e = 0; e < RN; ++e) bacc[e] = zero4();
    if (tid < 16 * TN) sbias[tid] = 0.f;
    f4 rn[RN], rk[RK];
    auto fetch = [&](int m0) {
#pragma unroll
        for (int e = 0; e < RN; ++e) {
            if constexpr (DYF == 1) rn[e] = (nok[e] && m0 + nr[e] < mend) ? unpack_bf16(*reinterpret_cast<const s4*>(reinterpret_cast<const unsigned short*>(dy) + np[e])) : zero4();
            else rn[e] = (nok[e] && m0 + nr[e] < mend) ? ld4(dy + np[e]) : zero4();
            np[e] += (long)RC * lddy;
        }
#pragma unroll
        for (int e = 0; e < RK; ++e) rk[e] = (kok[e] && m0 + kr[e] < mend) ? xl.get4(m0 + kr[e], kc[e]) : zero4();
    };
    auto stash = [&](int buf) {
        if constexpr (BF) {
            unsigned short* __restrict__ d16 = reinterpret_cast<unsigned short*>(sdy[buf]);
            unsigned short* __restrict__ x16 = reinterpret_cast<unsigned short*>(sx[buf]);
#pragma unroll
            for (int e = 0; e < RN; ++e) {
                if (tid + 256 * e < NV) *reinterpret_cast<s4*>(d16 + nl[e]) = pack_bf16(rn[e]);
                bacc[e] += rn[e];
            }
#pragma unroll
            for (int e = 0; e < RK; ++e)
                if (tid + 256 * e < KV) *reinterpret_cast<s4*>(x16 + kl[e]) = pack_bf16(rk[e]);
        } else {
#pragma unroll
        for (int e = 0; e < RN; ++e) {
            if (tid + 256 * e < NV) *reinterpret_cast<f4*>(&sdy[buf][nl[e]]) = rn[e];
            bacc[e] += rn[e];
        }
#pragma unroll
        for (int e = 0; e < RK; ++e)
            if (tid + 256 * e < KV) *reinterpret_cast<f4*>(&sx[buf][kl[e]]) = rk[e];
        }
    };
    int buf = 0;
    if (mbeg < mend) { fetch(mbeg); stash(0); }
    __syncthreads();
    for (int m0 = mbeg; m0 < mend; m0 += RC) {
        const bool more = m0 + RC < mend;
        if (more) fetch(m0 + RC);                           // next chunk's global loads fly under this chunk's MFMAs
        const float* __restrict__ pdy = sdy[buf];
        const float* __restrict__ px = sx[buf];
#pragma unroll
        for (int st = 0; st < RC / 16; ++st)
#pragma unroll
            for (int t = 0; t < TPW; ++t)
                if (tok[t]) {
                    if constexpr (BF) {
                        typedef __attribute__((address_space(3))) s4 lds_s4;
                        const unsigned short* ta = reinterpret_cast<const unsigned short*>(pdy) + offA[t] + (st * TN) * BST;
                        const unsigned short* tb = reinterpret_cast<const unsigned short*>(px) + offB[t] + (st * TK) * BST;
                        acc[t] = mfma16_bf16(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)ta),
                                             __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)tb), acc[t]);
                    } else {
                    const float* ta = pdy + offA[t] + (16 * st) * LDN;                    // rows 16st+4q .. +3 of column (a, i)
                    const float* tb = px + offB[t] + (16 * st) * LDK;
                    f4 av, bv;
                    av.x = ta[0]; av.y = ta[LDN]; av.z = ta[2 * LDN]; av.w = ta[3 * LDN];
                    bv.x = tb[0]; bv.y = tb[LDK]; bv.z = tb[2 * LDK]; bv.w = tb[3 * LDK];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t] = mfma16(av[j], bv[j], acc[t]);
                    }
                }
        if (more) stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        if (!tok[t]) continue;
        const int tile = wave + 4 * t;
        const int a = tile / TK, b = tile - a * TK;
        const int k = k0 + 16 * b + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + 16 * a + 4 * q + r;
            if (n < N && k < K) atomicAdd(dW + xl.waddr(n, k, ldw), acc[t][r]);
        }
    }
    if (do_bias) {                                          // column sums of dY from the values this thread staged
#pragma unroll
        for (int e = 0; e < RN; ++e)
            if (tid + 256 * e < NV) {
                const int c = ((tid + 256 * e) - nr[e] * C4N) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) atomicAdd(&sbias[c + j], bacc[e][j]);
            }
        __syncthreads();
        if (tid < 16 * TN && n0 + tid < N) atomicAdd(dbias + n0 + tid, sbias[tid]);
    }
}

template <int TN, int TK, class XL>
static inline int launch_wgrad16(const float* dy, long lddy, const XL& xl, float* dW, long ldw, float* dbias,
                                 int M, int N, int K, hipStream_t s, int dyfmt = 0) {
    if (M <= 0) return LEOD_OK;
    if ((N & 3) || (K & 3) || (lddy & 3)) return LEOD_ERR_ARG;      // 16-byte row loads
    const int tiles = cdiv(N, TN * 16) * cdiv(K, TK * 16);
    // ~768 workgroups in total, at least 4 staged chunks (128 rows) each: every workgroup ends with one fp32 atomic
    // per dW element, so a few fat workgroups beat many thin ones
    static const int tune_blocks = 768;   // tuning knob
    static const int tune_minrows = 128;
    int rpb = cdiv(M, max(1, tune_blocks / tiles));
    rpb = max(tune_minrows, ((rpb + 31) / 32) * 32);
    dim3 grid(cdiv(M, rpb), cdiv(N, TN * 16), cdiv(K, TK * 16));
    if (dyfmt) {                                             // bf16 dY rows: Linear layers in precision mode bf16 only
        if constexpr (x_two_phase<XL>::value) {
            if (leod_precision() != 1) return LEOD_ERR_ARG;
            hipLaunchKernelGGL((wgrad16_kernel<TN, TK, true, XL, 1>), grid, dim3(256), 0, s, dy, lddy, xl, dW, ldw, dbias, M, N, K, rpb, dyfmt);
            return leod_launch_status();
        } else return LEOD_ERR_ARG;
    }
    if (leod_precision() == 1) hipLaunchKernelGGL((wgrad16_kernel<TN, TK, true, XL>), grid, dim3(256), 0, s, dy, lddy, xl, dW, ldw, dbias, M, N, K, rpb, dyfmt);
    else hipLaunchKernelGGL((wgrad16_kernel<TN, TK, false, XL>), grid, dim3(256), 0, s, dy, lddy, xl, dW, ldw, dbias, M, N, K, rpb, dyfmt);
    return leod_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------
// Wave-tiled wgrad for the time-batched schedule (M = T*B*H*W rows: hundreds of thousands).  Same staging as above
// (coalesced 16-byte loads, transposed LDS, double buffer), but
//   * the workgroup tile is (TN x TK) MFMA tiles chosen per layer shape so that dY / X are streamed (almost) once:
//     192x48, 48x192, 96x96 or 48x48 outputs,
//   * each wave owns a contiguous WA x WB block of tiles and keeps its operand fragments in registers across the block:
//     WA + WB ds_read_b128 feed 4*WA*WB MFMAs (3x3: 6 reads per 36 MFMAs; the round-robin kernel above needs 18),
//   * waves left over after tiling the output (48x48) split the 16-row steps of a chunk (MS-way) instead.
// dW accumulates with one fp32 atomic per element and workgroup, dbias from the staged dY values as before.
// ---------------------------------------------------------------------------------------------------------------------
// LDS floats of one row group's double-buffered staging / of the whole workgroup (NG row groups: the staging of all groups, or the
// accumulator exchange of the epilogue that reuses the same bytes)
template <int TN, int TK, int RC, bool BF>
constexpr int wgradw_stage_floats() {
    return BF ? 2 * (((RC / 16) * TN * (16 * 16 + 16)) / 2 + ((RC / 16) * TK * (16 * 16 + 16)) / 2) : 2 * (RC * (16 * TN + 4) + RC * (16 * TK + 4));
}
template <int TN, int TK, int WA, int WB, int RC, bool BF, int NG>
constexpr int wgradw_lds_floats() {
    return NG * wgradw_stage_floats<TN, TK, RC, BF>() > (NG - 1) * WA * WB * 1024 ? NG * wgradw_stage_floats<TN, TK, RC, BF>() : (NG - 1) * WA * WB * 1024;
}
// NG > 1: the workgroup is NG independent 4-wave ROW GROUPS (own staging buffers, alternate row chunks, common barriers) that add
// their accumulators through LDS before the epilogue: the same waves per CU in flight with 1 / NG of the dW atomics (each a fabric
// write of 4 bytes: 9.4 M of them per launch were 20-30 us of a ~100 us launch).  Measured: NG = 2 -13.6 % over the 16 Linear shapes
// of RVT-S in bf16 mode; NG = 4 (16-wave workgroups, exchange in two halves) is SLOWER than NG = 2 on every stage-2 shape
// (91 -> 104, 104 -> 125, 146 -> 180 us): one barrier couples 16 waves per chunk.
template <int TN, int TK, int WA, int WB, int RC, bool BF, class XL, int DYF = -1, int XM = 0, int NG = 1, int PDT = 1>   // DYF: dY fp32 (0) / bf16 (1); -1: runtime dyfmt
__global__ __launch_bounds__(256 * NG, 4) void wgradw_kernel(const float* __restrict__ dy, long lddy, XL xl, float* dW, long ldw,
                                                     float* dbias, int M, int N, int K, int dyfmt) {
    constexpr int NWN = TN / WA, NWK = TK / WB, MS = 4 / (NWN * NWK);      // wave grid over the tile, row-step split
    static_assert(TN % WA == 0 && TK % WB == 0 && NWN * NWK * MS == 4, "4 waves must tile the workgroup");
    constexpr int STEPS = RC / 16;
    static_assert(STEPS % MS == 0, "row steps must split evenly over the waves");
    // dY / X chunks stay in their natural [row][col] layout in LDS: 16-byte stores (conflict-free), MFMA fragments by
    // 4 x ds_read_b32 -- rows 4q+j of a 32-lane group land in distinct 16-bank halves because LDN, LDK == 4 (mod 8).
    // (The first version transposed on the way in: 4 scalar stores per float4 with 6- to 12-way bank conflicts.)
    // BF (precision mode bf16): the chunks are rounded to bf16 when stashed and kept as [16 row][16 col] blocks (8-byte stores of 4
    // columns; block stride 512 + 32 bytes); a fragment = 4 consecutive ROWS of one column per lane is ONE ds_read_b64_tr_b16
    // (the hardware transposes the [4 row][16 col] sub-block of each 16-lane group) instead of four ds_read_b32, and one
    // v_mfma_f32_16x16x16_bf16 contracts the 16 rows of a step.
    constexpr int LDN = 16 * TN + 4, LDK = 16 * TK + 4;
    constexpr int BST = 16 * 16 + 16;
    constexpr int C4N = TN * 4, C4K = TK * 4;
    constexpr int NV = C4N * RC, KV = C4K * RC;
    constexpr int RN = (NV + 255) / 256, RK = (KV + 255) / 256;
    constexpr int SDY = BF ? ((RC / 16) * TN * BST) / 2 : RC * LDN, SX = BF ? ((RC / 16) * TK * BST) / 2 : RC * LDK;
    static_assert(2 * (SDY + SX) == wgradw_stage_floats<TN, TK, RC, BF>(), "staging size");
    __shared__ __attribute__((aligned(16))) float slds[wgradw_lds_floats<TN, TK, WA, WB, RC, BF, NG>()];
    __shared__ float sbias[16 * TN];
    const int grp = NG > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : 0;      // row group of this wave
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    float* const sdy0 = slds + grp * (2 * (SDY + SX));       // [2][SDY] then [2][SX] of this group
    float* const sx0 = sdy0 + 2 * SDY;
    const int i = lane & 15, q = lane >> 4;
    const int wa = wave % NWN, wb = (wave / NWN) % NWK, ws = wave / (NWN * NWK);
    const int n0 = blockIdx.y * TN * 16, k0 = blockIdx.z * TK * 16;
    // chunk c of this workgroup covers rows (blockIdx.x + c * gridDim.x) * RC ..: at any moment the resident workgroups
    // stream ONE contiguous window of dY / X (DRAM-page and TLB friendly) instead of gridDim.x far-apart row ranges
    const int mbeg = (blockIdx.x * NG + grp) * RC;
    const long mstride = (long)gridDim.x * NG * RC;
    const int mend = M;
    const bool do_bias = dbias != nullptr && blockIdx.z == 0;
    int nr[RN], kr[RK], nl[RN], kl[RK], kc[RK];
    long noff[RN];
    const float* __restrict__ dyb = dy;
    bool nok[RN], kok[RK];
#pragma unroll
    for (int e = 0; e < RN; ++e) {
        const int s = tid + 256 * e, r = s / C4N, c = (s - r * C4N) * 4;
        nr[e] = r; nl[e] = BF ? ((r >> 4) * TN + (c >> 4)) * BST + (r & 15) * 16 + (c & 15) : r * LDN + c; nok[e] = s < NV && n0 + c < N;
        noff[e] = (long)r * lddy + n0 + c;
    }
#pragma unroll
    for (int e = 0; e < RK; ++e) {
        const int s = tid + 256 * e, r = s / C4K, c = (s - r * C4K) * 4;
        kr[e] = r; kl[e] = BF ? ((r >> 4) * TK + (c >> 4)) * BST + (r & 15) * 16 + (c & 15) : r * LDK + c; kc[e] = k0 + c; kok[e] = s < KV && k0 + c < K;
    }
    const int offA = BF ? (wa * WA) * BST + (4 * q + (i >> 2)) * 16 + 4 * (i & 3) : (4 * q) * LDN + 16 * wa * WA + i;
    const int offB = BF ? (wb * WB) * BST + (4 * q + (i >> 2)) * 16 + 4 * (i & 3) : (4 * q) * LDK + 16 * wb * WB + i;
    f4 acc[WA][WB], bacc[RN];
#pragma unroll
    for (int a = 0; a < WA; ++a)
#pragma unroll
        for (int b = 0; b < WB; ++b) acc[a][b] = zero4();
#pragma unroll
    for (int e = 0; e < RN; ++e) bacc[e] = zero4();
    if ((int)threadIdx.x < 16 * TN) sbias[threadIdx.x] = 0.f;
    // ---- two-phase staging: issue() only starts the loads of a chunk (unconditional, on clamped coordinates), finish() turns the
    // raw registers into operand values (bf16 unpack, GELU, LayerNorm, edge masks) right before they are stashed.  With the
    // arithmetic of the loaders sitting behind each load the compiler waited for every load in turn: 5-6 serialized memory
    // round trips per 16-row chunk.
    constexpr bool TP = x_two_phase<XL>::value;
    constexpr bool TP_ = x_two_phase<XL>::value;
    // PDT = 2: TWO raw register sets -- the loads of chunk c + 2 are issued before the MFMAs of chunk c, while the set of chunk c + 1 waits
    // to be stashed, so every load has a whole iteration more to land (69 % of the wave cycles of the single-set version sit in
    // s_waitcnt / barriers, profiles/r03_d_wgradw_pmc.txt).  Kept as an option, not instantiated (see launch_wgradw_cfg): no gain.
    constexpr int PD = (TP_ && RC == 16) ? PDT : 1;
    f4 rn_[PD][RN], rk_[PD][RK]; f2_ rst_[PD][RK]; u2_ hn_[PD][RN], hk_[PD][RK];      // 16-byte and 8-byte raw registers are separate: no copies behind a load
    __shared__ __attribute__((aligned(16))) float sln[2][TP ? 16 * TK : 4];   // LayerNorm weight / bias of this workgroup's columns
    long noffc[RN]; int kcc[RK];
#pragma unroll
    for (int e = 0; e < RN; ++e) noffc[e] = nok[e] ? noff[e] : (long)nr[e] * lddy;
#pragma unroll
    for (int e = 0; e < RK; ++e) kcc[e] = kok[e] ? kc[e] : 0;      // (the raw registers stay uninitialised: each is written and read under the same
                                                                    // workgroup-uniform mode test, and an initial value would become a copy behind every load)
    if constexpr (TP && XM == 1) {
        for (int c = threadIdx.x; c < 16 * TK; c += 256 * NG) { sln[0][c] = k0 + c < K ? xl.ln_w[k0 + c] : 0.f; sln[1][c] = k0 + c < K ? xl.ln_b[k0 + c] : 0.f; }
        __syncthreads();
    }
    auto fetch = [&](long m0, auto setc) {
        constexpr int SI = decltype(setc)::value;
        f4 (&rn)[RN] = rn_[SI]; f4 (&rk)[RK] = rk_[SI]; f2_ (&rst)[RK] = rst_[SI]; u2_ (&hn)[RN] = hn_[SI]; u2_ (&hk)[RK] = hk_[SI];
        if constexpr (!TP) {                                  // loaders without the raw / finish pair: values in one go
#pragma unroll
            for (int e = 0; e < RN; ++e) {
                if (dyfmt) rn[e] = (nok[e] && m0 + nr[e] < mend) ? unpack_bf16(*reinterpret_cast<const s4*>(reinterpret_cast<const unsigned short*>(dyb) + (m0 * lddy + noff[e]))) : zero4();
                else rn[e] = (nok[e] && m0 + nr[e] < mend) ? ld4(dyb + (m0 * lddy + noff[e])) : zero4();
            }
#pragma unroll
            for (int e = 0; e < RK; ++e) rk[e] = (kok[e] && m0 + kr[e] < mend) ? xl.get4((int)m0 + kr[e], kc[e]) : zero4();
            return;
        } else {
#pragma unroll
        for (int e = 0; e < RN; ++e) {
            const long mr = min(m0 + nr[e], (long)mend - 1) - nr[e];                 // clamped chunk origin of this slot
            if constexpr (DYF == 1) hn[e] = *reinterpret_cast<const u2_*>(reinterpret_cast<const unsigned short*>(dyb) + (mr * lddy + noffc[e]));
            else rn[e] = ld4(dyb + (mr * lddy + noffc[e]));
        }
#pragma unroll
        for (int e = 0; e < RK; ++e) xl.template raw4<XM>((int)min(m0 + kr[e], (long)mend - 1), kcc[e], rk[e], hk[e], rst[e]);
        }
    };
    auto finish = [&](long m0, auto setc) {
        constexpr int SI = decltype(setc)::value;
        f4 (&rn)[RN] = rn_[SI]; f4 (&rk)[RK] = rk_[SI]; f2_ (&rst)[RK] = rst_[SI]; u2_ (&hn)[RN] = hn_[SI]; u2_ (&hk)[RK] = hk_[SI];
        if constexpr (TP) {
#pragma unroll
        for (int e = 0; e < RN; ++e) {
            f4 v;
            if constexpr (DYF == 1) v = unpack_bf16(__builtin_bit_cast(s4, hn[e])); else v = rn[e];
            rn[e] = (nok[e] && m0 + nr[e] < mend) ? v : zero4();
        }
#pragma unroll
        for (int e = 0; e < RK; ++e) {
            f4 g = zero4(), b = zero4();
            if constexpr (XM == 1) { g = *reinterpret_cast<const f4*>(&sln[0][kcc[e] - k0]); b = *reinterpret_cast<const f4*>(&sln[1][kcc[e] - k0]); }
            const f4 v = xl.template fin4<XM>(rk[e], hk[e], rst[e], kcc[e], g, b);
            rk[e] = (kok[e] && m0 + kr[e] < mend) ? v : zero4();
        }
        }
    };
    auto stash = [&](int buf, auto setc) {
        constexpr int SI = decltype(setc)::value;
        f4 (&rn)[RN] = rn_[SI]; f4 (&rk)[RK] = rk_[SI];
        if constexpr (BF) {
            unsigned short* __restrict__ d16 = reinterpret_cast<unsigned short*>(sdy0 + buf * SDY);
            unsigned short* __restrict__ x16 = reinterpret_cast<unsigned short*>(sx0 + buf * SX);
#pragma unroll
            for (int e = 0; e < RN; ++e) {
                if (tid + 256 * e < NV) *reinterpret_cast<s4*>(d16 + nl[e]) = pack_bf16(rn[e]);
                bacc[e] += rn[e];
            }
#pragma unroll
            for (int e = 0; e < RK; ++e)
                if (tid + 256 * e < KV) *reinterpret_cast<s4*>(x16 + kl[e]) = pack_bf16(rk[e]);
        } else {
#pragma unroll
        for (int e = 0; e < RN; ++e) {
            if (tid + 256 * e < NV) *reinterpret_cast<f4*>(sdy0 + buf * SDY + nl[e]) = rn[e];
            bacc[e] += rn[e];
        }
#pragma unroll
        for (int e = 0; e < RK; ++e)
            if (tid + 256 * e < KV) *reinterpret_cast<f4*>(sx0 + buf * SX + kl[e]) = rk[e];
        }
    };
    int buf = 0;
    const std::integral_constant<int, 0> S0{};
    const std::integral_constant<int, PD - 1> S1{};
    auto mfma_chunk = [&](int buf) {
        if constexpr (BF) {
            typedef __attribute__((address_space(3))) s4 lds_s4;
            const unsigned short* __restrict__ pdy = reinterpret_cast<const unsigned short*>(sdy0 + buf * SDY) + offA;
            const unsigned short* __restrict__ px = reinterpret_cast<const unsigned short*>(sx0 + buf * SX) + offB;
#pragma unroll
            for (int st = ws; st < STEPS; st += MS) {
                s4 pa[WA], pb[WB];
#pragma unroll
                for (int a = 0; a < WA; ++a) pa[a] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(pdy + (st * TN + a) * BST));
#pragma unroll
                for (int b = 0; b < WB; ++b) pb[b] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(px + (st * TK + b) * BST));
#pragma unroll
                for (int a = 0; a < WA; ++a)
#pragma unroll
                    for (int b = 0; b < WB; ++b) acc[a][b] = mfma16_bf16(pa[a], pb[b], acc[a][b]);
            }
        } else {
        const float* __restrict__ pdy = sdy0 + buf * SDY + offA;
        const float* __restrict__ px = sx0 + buf * SX + offB;
#pragma unroll
        for (int st = ws; st < STEPS; st += MS) {
            f4 av[WA], bv[WB];
#pragma unroll
            for (int a = 0; a < WA; ++a) {
                const float* t = pdy + (16 * st) * LDN + 16 * a;
                av[a].x = t[0]; av[a].y = t[LDN]; av[a].z = t[2 * LDN]; av[a].w = t[3 * LDN];
            }
#pragma unroll
            for (int b = 0; b < WB; ++b) {
                const float* t = px + (16 * st) * LDK + 16 * b;
                bv[b].x = t[0]; bv[b].y = t[LDK]; bv[b].z = t[2 * LDK]; bv[b].w = t[3 * LDK];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < WA; ++a)
#pragma unroll
                    for (int b = 0; b < WB; ++b) acc[a][b] = mfma16(av[a][j], bv[b][j], acc[a][b]);
        }
        }
    };
    fetch(mbeg, S0);
    finish(mbeg, S0);
    stash(0, S0);
    // the trip count is that of the workgroup's first row group, so that every group meets every barrier (a group whose chunk lies
    // past the last row stages zeros: clamped loads, masked in finish)
    const long mb0 = (long)blockIdx.x * NG * RC;
    if constexpr (PD == 1) {
        __syncthreads();
        for (long mb = mb0; mb < mend; mb += mstride) {
            const long m0 = mb + grp * RC;
            const bool more = mb + mstride < mend;
            if (more) fetch(m0 + mstride, S0);              // next chunk's global loads fly under this chunk's MFMAs
            mfma_chunk(buf);
            if (more) { finish(m0 + mstride, S0); stash(buf ^ 1, S0); }
            __syncthreads();
            buf ^= 1;
        }
    } else {
        // chunk c sits in LDS buffer c & 1, chunk c + 1 in register set (c + 1) & 1 (issued one iteration ago), and the loads of
        // chunk c + 2 go to set c & 1 (stashed one iteration ago) before the MFMAs of chunk c: unrolled by two so that the set is static
        if (mb0 + mstride < mend) fetch(mbeg + mstride, S1);
        __syncthreads();
        for (long mb = mb0; mb < mend; mb += 2 * mstride) {
            const long m0 = mb + grp * RC;
            {   // even chunk c: LDS buffer 0, next chunk in set 1, chunk after that into set 0
                const bool more1 = mb + mstride < mend, more2 = mb + 2 * mstride < mend;
                if (more2) fetch(m0 + 2 * mstride, S0);
                mfma_chunk(0);
                if (more1) { finish(m0 + mstride, S1); stash(1, S1); }
                __syncthreads();
                if (!more1) break;
            }
            {   // odd chunk c + 1: LDS buffer 1, next chunk in set 0, chunk after that into set 1
                const bool more2 = mb + 2 * mstride < mend, more3 = mb + 3 * mstride < mend;
                if (more3) fetch(m0 + 3 * mstride, S1);
                mfma_chunk(1);
                if (more2) { finish(m0 + 2 * mstride, S0); stash(0, S0); }
                __syncthreads();
            }
        }
    }
    if constexpr (NG > 1) {
        // accumulator exchange (the staging bytes are dead after the loop's last barrier): tile ab belongs to group ab % NG, every
        // other group parks its copy in LDS, the owner adds them up and issues the tile's atomics
        f4* sred = reinterpret_cast<f4*>(slds);
#pragma unroll
        for (int a = 0; a < WA; ++a)
#pragma unroll
            for (int b = 0; b < WB; ++b) {
                const int ab = a * WB + b, owner = ab % NG;
                if (grp != owner) sred[(ab * (NG - 1) + (grp - (grp > owner ? 1 : 0))) * 256 + tid] = acc[a][b];
            }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < WA; ++a)
#pragma unroll
            for (int b = 0; b < WB; ++b) {
                const int ab = a * WB + b, owner = ab % NG;
                if (grp == owner) {
#pragma unroll
                    for (int o = 0; o < NG - 1; ++o) acc[a][b] += sred[(ab * (NG - 1) + o) * 256 + tid];
                }
            }
    }
#pragma unroll
    for (int a = 0; a < WA; ++a)
#pragma unroll
        for (int b = 0; b < WB; ++b) {
            if (NG > 1 && grp != (a * WB + b) % NG) continue;
            const int k = k0 + 16 * (wb * WB + b) + i;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 16 * (wa * WA + a) + 4 * q + r;
                if (n < N && k < K) atomicAdd(dW + xl.waddr(n, k, ldw), acc[a][b][r]);
            }
        }
    if (do_bias) {
#pragma unroll
        for (int e = 0; e < RN; ++e)
            if (tid + 256 * e < NV) {
                const int c = ((tid + 256 * e) - nr[e] * C4N) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) atomicAdd(&sbias[c + j], bacc[e][j]);
            }
        __syncthreads();
        if (grp == 0 && tid < 16 * TN && n0 + tid < N) atomicAdd(dbias + n0 + tid, sbias[tid]);
    }
}

template <int TN, int TK, int WA, int WB, int RC, class XL>
static inline int launch_wgradw_cfg(const float* dy, long lddy, const XL& xl, float* dW, long ldw, float* dbias,
                                    int M, int N, int K, hipStream_t s, int dyfmt = 0) {
    const int tiles = cdiv(N, TN * 16) * cdiv(K, TK * 16);
    static const int tune_blocks = 1024;   // 4 workgroups per CU resident
    static const int tune_ng = 2;               // row groups per workgroup
    static const int tune_align = 1;
    const int chunks = cdiv(M, RC);
    auto go = [&](auto bfc, auto dyfc, auto xmc, auto ngc) {
        constexpr bool BFV = decltype(bfc)::value;
        constexpr int DYFV = decltype(dyfc)::value, XMV = decltype(xmc)::value, NGR = decltype(ngc)::value;
        // the row groups' staging (or their accumulator exchange) must fit the 64 KB of static LDS next to sbias / sln
        constexpr int NGE = (wgradw_lds_floats<TN, TK, WA, WB, RC, BFV, NGR>() + 16 * TN + 32 * TK + 64) * 4 <= 64 * 1024 ? NGR : 1;
        // >= 4 chunks per row group (one atomic per dW element and workgroup); a multiple of 8 workgroups per output tile puts the
        // workgroups that stream the same rows for different tiles on one XCD (dispatch slot = x + gx * tile): its L2 serves the re-reads
        int gx = max(1, min(chunks / (4 * NGE), tune_blocks / (tiles * NGE)));
        if (tune_align && gx >= 16) gx &= ~7;
        dim3 grid(gx, cdiv(N, TN * 16), cdiv(K, TK * 16));
        // PDT = 2 (loads two chunks ahead, a second raw register set) was measured and is NOT instantiated: the plain / fp16 variants are
        // unchanged (stage 1 fc2 147 vs 149 us: they are not latency-bound) and the LayerNorm variants spill at 128 VGPRs (130 -> 213 us)
        hipLaunchKernelGGL((wgradw_kernel<TN, TK, WA, WB, RC, BFV, XL, DYFV, XMV, NGE, 1>), grid, dim3(256 * NGE), 0, s, dy, lddy, xl, dW, ldw,
                           dbias, M, N, K, dyfmt);
    };
    using std::integral_constant;
#define LEOD_WGRADW_GO(BFV, DYFV, XMV)                                                                                                  \
    do {                                                                                                                                \
        if (tune_ng == 2) go(integral_constant<bool, BFV>{}, integral_constant<int, DYFV>{}, integral_constant<int, XMV>{}, integral_constant<int, 2>{}); \
        else go(integral_constant<bool, BFV>{}, integral_constant<int, DYFV>{}, integral_constant<int, XMV>{}, integral_constant<int, 1>{});             \
    } while (0)
    if constexpr (x_two_phase<XL>::value) {                  // one instantiation per (dY format, X mode): no mode branches in the loop
        const int xm = xl.x_mode();
        if (leod_precision() == 1) {
            if (dyfmt) { if (xm == 2) LEOD_WGRADW_GO(true, 1, 2); else if (xm == 1) LEOD_WGRADW_GO(true, 1, 1); else LEOD_WGRADW_GO(true, 1, 0); }
            else { if (xm == 2) LEOD_WGRADW_GO(true, 0, 2); else if (xm == 1) LEOD_WGRADW_GO(true, 0, 1); else LEOD_WGRADW_GO(true, 0, 0); }
        } else {
            if (dyfmt || xm == 2) return LEOD_ERR_ARG;       // 16-bit tensors exist in precision mode bf16 only
            if (xm == 1) LEOD_WGRADW_GO(false, 0, 1); else LEOD_WGRADW_GO(false, 0, 0);
        }
    } else {                                                 // gather loaders (conv / stem): one row group
        if (leod_precision() == 1) go(integral_constant<bool, true>{}, integral_constant<int, -1>{}, integral_constant<int, 0>{}, integral_constant<int, 1>{});
        else go(integral_constant<bool, false>{}, integral_constant<int, -1>{}, integral_constant<int, 0>{}, integral_constant<int, 1>{});
    }
#undef LEOD_WGRADW_GO
    return leod_launch_status();
}

// shape-driven choice of the workgroup tile (see the kernel header)
template <class XL>
static inline int launch_wgradw(const float* dy, long lddy, const XL& xl, float* dW, long ldw, float* dbias,
                                int M, int N, int K, hipStream_t s, int dyfmt = 0) {
    if (M <= 0) return LEOD_OK;
    if ((N & 3) || (K & 3) || (lddy & 3)) return LEOD_ERR_ARG;      // 16-byte row loads
    if (K <= 48 && N <= 48) return launch_wgradw_cfg<3, 3, 3, 3, 64>(dy, lddy, xl, dW, ldw, dbias, M, N, K, s, dyfmt);
    // 32-row chunks (half the barriers, twice the bytes in flight per round) pay for the short row ranges of stages 3-4 with plain or
    // fp16 X rows: 72 -> 63 us (fc2, stage 4), 118 -> 94 us (LSTM); the LayerNorm variant spills at 128 VGPRs and the long ranges of
    // stage 2 are indifferent or slower
    static const int rc32_on = 1;
    bool rc32 = false;
    if constexpr (x_two_phase<XL>::value) rc32 = rc32_on && M <= 65536 && xl.x_mode() != 1 && leod_precision() == 1;
    if (K <= 48) return launch_wgradw_cfg<12, 3, 3, 3, 16>(dy, lddy, xl, dW, ldw, dbias, M, N, K, s, dyfmt);
    if (N <= 48) return launch_wgradw_cfg<3, 12, 3, 3, 16>(dy, lddy, xl, dW, ldw, dbias, M, N, K, s, dyfmt);
    if (rc32) return launch_wgradw_cfg<6, 6, 3, 3, 32>(dy, lddy, xl, dW, ldw, dbias, M, N, K, s, dyfmt);
    return launch_wgradw_cfg<6, 6, 3, 3, 16>(dy, lddy, xl, dW, ldw, dbias, M, N, K, s, dyfmt);
}
// large row counts only: small problems keep the round-robin kernel (more workgroups per output tile)
static inline bool use_wgradw(int M) {
    static const int mode = 1;
    return mode != 0 && M >= 8192;
}
