// MaxViT partition attention core (window = contiguous tiles, grid = dilated tiles) for the RVT
// backbone: softmax(q k^T * d^-0.5) v per (partition, head), forward and backward.
// Reference: models/layers/maxvit/maxvit.py:343-354 (SelfAttentionCl), :273-304 (partitions).
//
// Layout: qkv is the row-major [M, 3C] output of the qkv Linear in *image token order*
// (M = B*H*W rows of an NHWC map); head h owns columns [h*3d, (h+1)*3d) = (q | k | v)
// (maxvit.py:347 view(B,-1,heads,3d)).  Partitioning is pure index math on the row number, so no
// partition/reverse copies exist.  Output O is [M, C] with channel = h*d + c.
//
// Mapping to CDNA4: one wave owns (partition, head, 16-token tile).  The score tile is computed
// *transposed* (S^T = K Q^T, keys along MFMA rows) so that the softmax'ed accumulator registers are
// already laid out as the A operand of the P.V MFMA (k index = lane>>4): no LDS, no shuffles apart
// from the 2-step cross-row-group reductions.  P = 80 tokens -> 5 key tiles, 20 accumulator VGPRs.
#include "common.hpp"
#include <type_traits>

struct AttnGeom {
    int B, H, W, C, heads, d, ph, pw, window;   // window: 1 = window partition, 0 = grid partition
    int fmt;                                    // precision mode bf16, LDS kernels only: bit 0 = qkv holds bf16, bit 1 = dqkv is written as
                                                // bf16 (q, k, v only ever enter bf16 MFMAs; dqkv's consumers feed bf16 MFMAs)
};

// Row (token) addressing without integer division in the inner loops: every lane computes, ONCE per wave, the row of
// the tokens 16*mt + (lane & 15) of its partition (PT values in registers); the row of any other token of the
// partition is fetched from the lane that owns it with a wave shuffle.  (The first version re-derived
// (b, py, px, ty, tx) with 4 integer divisions per access: 3600 VALU instructions per wave for 80 MFMAs.)
template <int PT>
struct TokRows {
    long r[PT];                                   // row of token 16*mt + i, or -1 beyond the partition
    __device__ __forceinline__ void init(const AttnGeom& g, int p, int i) {
        const int nH = g.H / g.ph, nW = g.W / g.pw;
        const int per = nH * nW;
        const int b = p / per, rem = p - b * per;
        const int py = rem / nW, px = rem - py * nW;
        const int P = g.ph * g.pw;
        const long base = g.window ? ((long)b * g.H + (long)py * g.ph) * g.W + (long)px * g.pw
                                   : ((long)b * g.H + py) * g.W + px;
        const int sy = g.window ? g.W : nH * g.W, sx = g.window ? 1 : nW;
#pragma unroll
        for (int mt = 0; mt < PT; ++mt) {
            const int t = 16 * mt + i;
            const int ty = t / g.pw, tx = t - ty * g.pw;
            r[mt] = t < P ? base + (long)ty * sy + (long)tx * sx : -1;
        }
    }
    // row of token 16*mt + j, j in [0,16) possibly different per lane (wave shuffle from lane j)
    __device__ __forceinline__ long at(int mt, int j) const {
        long v = r[0];
#pragma unroll
        for (int m = 1; m < PT; ++m) v = (m == mt) ? r[m] : v;      // mt is a compile-time constant at every call site
        const int lo = __shfl((int)(v & 0xffffffffu), j, 64), hi = __shfl((int)(v >> 32), j, 64);
        return ((long)hi << 32) | (unsigned int)lo;
    }
};

// float4 of a head slice (part: 0 q, 1 k, 2 v) at channel c..c+3, zero outside [0,d) or for row < 0
__device__ __forceinline__ f4 ld_head4(const float* base, long row, long ld, int off, int c, int d) {
    if (row < 0 || c >= d) return zero4();
    return ld4(base + row * ld + off + c);
}
__device__ __forceinline__ float ld_head1(const float* base, long row, long ld, int off, int c, int d) {
    if (row < 0 || c >= d) return 0.f;
    return base[row * ld + off + c];
}

// ---------------------------------------------------------------------------------------------------
// forward: one wave = (partition p, head h, query tile qt)
// ---------------------------------------------------------------------------------------------------
template <int PT, int DCH>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                       float* __restrict__ lse, AttnGeom g, float scale, int ntasks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int task = blockIdx.x * 4 + wave;
    if (task >= ntasks) return;
    const int i = lane & 15, rg = lane >> 4;
    const int qt = task % PT, h = (task / PT) % g.heads, p = task / (PT * g.heads);
    const int P = g.ph * g.pw, d = g.d;
    const long ld = 3L * g.C;
    const int hoff = h * 3 * d;

    TokRows<PT> tr;
    tr.init(g, p, i);
    long rowq = tr.r[0];
#pragma unroll
    for (int m = 1; m < PT; ++m) rowq = (m == qt) ? tr.r[m] : rowq;          // qt is wave-uniform
    f4 qf[DCH];
#pragma unroll
    for (int ch = 0; ch < DCH; ++ch) qf[ch] = ld_head4(qkv, rowq, ld, hoff, 16 * ch + 4 * rg, d);

    f4 s[PT];
#pragma unroll
    for (int mt = 0; mt < PT; ++mt) {
        s[mt] = zero4();
        const long rowk = tr.r[mt];
#pragma unroll
        for (int ch = 0; ch < DCH; ++ch) {
            const f4 kf = ld_head4(qkv, rowk, ld, hoff + d, 16 * ch + 4 * rg, d);
#pragma unroll
            for (int j = 0; j < 4; ++j) s[mt] = mfma16(kf[j], qf[ch][j], s[mt]);
        }
    }
    // softmax over keys (rows of S^T) for query column i
    float mx = -INFINITY;
#pragma unroll
    for (int mt = 0; mt < PT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * mt + 4 * rg + r;
            const float v = key < P ? s[mt][r] * scale : -INFINITY;
            s[mt][r] = v;
            mx = fmaxf(mx, v);
        }
    mx = quad16_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int mt = 0; mt < PT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = expf(s[mt][r] - mx);      // exp(-inf) = 0 for padded keys
            s[mt][r] = e;
            sum += e;
        }
    sum = quad16_sum(sum);
    const float inv = 1.0f / sum;
    if (lse && rg == 0 && rowq >= 0) lse[rowq * g.heads + h] = mx + logf(sum);

    // O = P V : A = P (s registers as they are), B = V rows gathered per (key = 16mt + 4rg + r)
    f4 o[DCH];
#pragma unroll
    for (int ct = 0; ct < DCH; ++ct) o[ct] = zero4();
#pragma unroll
    for (int mt = 0; mt < PT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long rowv = tr.at(mt, 4 * rg + r);
#pragma unroll
            for (int ct = 0; ct < DCH; ++ct) {
                const float vv = ld_head1(qkv, rowv, ld, hoff + 2 * d, 16 * ct + i, d);
                o[ct] = mfma16(s[mt][r] * inv, vv, o[ct]);
            }
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long row = tr.at(qt, 4 * rg + r);
        if (row < 0) continue;
#pragma unroll
        for (int ct = 0; ct < DCH; ++ct) {
            const int c = 16 * ct + i;
            if (c < d) out[row * g.C + h * d + c] = o[ct][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// backward, query-owned part: dQ and D[query] = sum_key P * dP
// ---------------------------------------------------------------------------------------------------
template <int PT, int DCH>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                         const float* __restrict__ lse, float* __restrict__ dqkv,
                                                         float* __restrict__ dsum, AttnGeom g, float scale, int ntasks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int task = blockIdx.x * 4 + wave;
    if (task >= ntasks) return;
    const int i = lane & 15, rg = lane >> 4;
    const int qt = task % PT, h = (task / PT) % g.heads, p = task / (PT * g.heads);
    const int P = g.ph * g.pw, d = g.d;
    const long ld = 3L * g.C;
    const int hoff = h * 3 * d;

    TokRows<PT> tr;
    tr.init(g, p, i);
    long rowq = tr.r[0];
#pragma unroll
    for (int m = 1; m < PT; ++m) rowq = (m == qt) ? tr.r[m] : rowq;
    f4 qf[DCH], dof[DCH];
#pragma unroll
    for (int ch = 0; ch < DCH; ++ch) {
        qf[ch] = ld_head4(qkv, rowq, ld, hoff, 16 * ch + 4 * rg, d);
        dof[ch] = ld_head4(dout, rowq, g.C, h * d, 16 * ch + 4 * rg, d);
    }
    const float l = rowq >= 0 ? lse[rowq * g.heads + h] : 0.f;
    f4 s[PT], dp[PT];
#pragma unroll
    for (int mt = 0; mt < PT; ++mt) {
        s[mt] = zero4(); dp[mt] = zero4();
        const long rowk = tr.r[mt];
#pragma unroll
        for (int ch = 0; ch < DCH; ++ch) {
            const f4 kf = ld_head4(qkv, rowk, ld, hoff + d, 16 * ch + 4 * rg, d);
            const f4 vf = ld_head4(qkv, rowk, ld, hoff + 2 * d, 16 * ch + 4 * rg, d);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[mt] = mfma16(kf[j], qf[ch][j], s[mt]);
                dp[mt] = mfma16(vf[j], dof[ch][j], dp[mt]);
            }
        }
    }
    float D = 0.f;
#pragma unroll
    for (int mt = 0; mt < PT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * mt + 4 * rg + r;
            const float pr = (key < P && rowq >= 0) ? expf(s[mt][r] * scale - l) : 0.f;
            s[mt][r] = pr;
            D += pr * dp[mt][r];
        }
    D = quad16_sum(D);
    if (rg == 0 && rowq >= 0) dsum[rowq * g.heads + h] = D;
    f4 dq[DCH];
#pragma unroll
    for (int ct = 0; ct < DCH; ++ct) dq[ct] = zero4();
#pragma unroll
    for (int mt = 0; mt < PT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long rowk = tr.at(mt, 4 * rg + r);
            const float ds = s[mt][r] * (dp[mt][r] - D) * scale;
#pragma unroll
            for (int ct = 0; ct < DCH; ++ct) {
                const float kk = ld_head1(qkv, rowk, ld, hoff + d, 16 * ct + i, d);
                dq[ct] = mfma16(ds, kk, dq[ct]);
            }
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long row = tr.at(qt, 4 * rg + r);
        if (row < 0) continue;
#pragma unroll
        for (int ct = 0; ct < DCH; ++ct) {
            const int c = 16 * ct + i;
            if (c < d) dqkv[row * ld + hoff + c] = dq[ct][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// backward, key-owned part: dK and dV for 16 keys, looping over all queries of the partition
// ---------------------------------------------------------------------------------------------------
template <int PT, int DCH>
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                          const float* __restrict__ lse, const float* __restrict__ dsum,
                                                          float* __restrict__ dqkv, AttnGeom g, float scale, int ntasks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int task = blockIdx.x * 4 + wave;
    if (task >= ntasks) return;
    const int i = lane & 15, rg = lane >> 4;
    const int kt = task % PT, h = (task / PT) % g.heads, p = task / (PT * g.heads);
    const int P = g.ph * g.pw, d = g.d;
    const long ld = 3L * g.C;
    const int hoff = h * 3 * d;

    TokRows<PT> tr;
    tr.init(g, p, i);
    long rowk = tr.r[0];
#pragma unroll
    for (int m = 1; m < PT; ++m) rowk = (m == kt) ? tr.r[m] : rowk;
    f4 kf[DCH], vf[DCH];
#pragma unroll
    for (int ch = 0; ch < DCH; ++ch) {
        kf[ch] = ld_head4(qkv, rowk, ld, hoff + d, 16 * ch + 4 * rg, d);
        vf[ch] = ld_head4(qkv, rowk, ld, hoff + 2 * d, 16 * ch + 4 * rg, d);
    }
    f4 dk[DCH], dv[DCH];
#pragma unroll
    for (int ct = 0; ct < DCH; ++ct) { dk[ct] = zero4(); dv[ct] = zero4(); }
#pragma unroll
    for (int qm = 0; qm < PT; ++qm) {
        // S[query][key] (not transposed): A = Q rows, B = K^T
        const long rowq = tr.r[qm];
        f4 s = zero4(), dp = zero4();
#pragma unroll
        for (int ch = 0; ch < DCH; ++ch) {
            const f4 qf = ld_head4(qkv, rowq, ld, hoff, 16 * ch + 4 * rg, d);
            const f4 dof = ld_head4(dout, rowq, g.C, h * d, 16 * ch + 4 * rg, d);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s = mfma16(qf[j], kf[ch][j], s);
                dp = mfma16(dof[j], vf[ch][j], dp);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long rq = tr.at(qm, 4 * rg + r);          // query of accumulator row r; key = column i
            float pr = 0.f, ds = 0.f;
            if (rq >= 0 && rowk >= 0) {
                pr = expf(s[r] * scale - lse[rq * g.heads + h]);
                ds = pr * (dp[r] - dsum[rq * g.heads + h]) * scale;
            }
#pragma unroll
            for (int ct = 0; ct < DCH; ++ct) {
                const float dov = ld_head1(dout, rq, g.C, h * d, 16 * ct + i, d);
                const float qv = ld_head1(qkv, rq, ld, hoff, 16 * ct + i, d);
                dv[ct] = mfma16(pr, dov, dv[ct]);
                dk[ct] = mfma16(ds, qv, dk[ct]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long row = tr.at(kt, 4 * rg + r);
        if (row < 0) continue;
#pragma unroll
        for (int ct = 0; ct < DCH; ++ct) {
            const int c = 16 * ct + i;
            if (c < d) {
                dqkv[row * ld + hoff + d + c] = dk[ct][r];
                dqkv[row * ld + hoff + 2 * d + c] = dv[ct][r];
            }
        }
    }
}

// =====================================================================================================================
// LDS-staged variants (default).  One workgroup = (partition, group of HG heads), one wave = (head, 16-token tile).
// The qkv rows of the partition (per token HG*3d contiguous floats: head h owns columns [3dh, 3d(h+1)) = q|k|v) are
// fetched ONCE with coalesced 16-byte loads into LDS [token][HG*3d + 4]; every MFMA operand then comes from LDS:
//   A/B fragments along the head dim : one ds_read_b128 per lane (row stride == 4 mod 16 floats: conflict-free),
//   B fragments along the token dim  : ds_read_b32 (key rows 4q+r land in distinct 16-bank groups).
// The register-direct kernels above re-read K/V of a partition from global memory in every one of its PT waves and in
// operand layout (16 rows x 64 B per instruction, 4-byte V loads); they stay as the LEOD_ATTN_LDS=0 reference.
// Outputs go back through LDS so that a wave instruction stores whole per-token segments (O: d floats per head,
// dqkv: the full 3d*HG segment with dq, dk, dv of both passes of the fused backward).
// =====================================================================================================================
__device__ __forceinline__ long token_row(const AttnGeom& g, int p, int t) {
    const int nH = g.H / g.ph, nW = g.W / g.pw, per = nH * nW;
    if (t >= g.ph * g.pw) return -1;
    const int b = p / per, rem = p - b * per;
    const int py = rem / nW, px = rem - py * nW;
    const int ty = t / g.pw, tx = t - ty * g.pw;
    return g.window ? ((long)b * g.H + (long)py * g.ph + ty) * g.W + (long)px * g.pw + tx
                    : ((long)b * g.H + py + (long)ty * nH) * g.W + px + (long)tx * nW;
}
__device__ __forceinline__ f4 lds4(const float* p, bool ok) { return ok ? *reinterpret_cast<const f4*>(p) : zero4(); }

// Fragment of one token row along the head dimension for the contractions over d (Q K^T, dO V^T): full 16-wide chunks in the
// K-permuted layout (lane (i, rg) holds k = 16c + 4rg .. +3, four MFMAs per chunk) and, for d = 24, an 8-wide tail with
// k = 16 + 2rg .. +1 (TWO MFMAs, every lane valid) instead of a half-empty third chunk: 6 instead of 8 MFMAs per product
// and no select per operand element.
template <int D>
struct KFrag {
    static constexpr int NC = D / 16, TAIL = D % 16;
    static_assert(TAIL == 0 || TAIL == 8, "head dimension must be 16c or 16c + 8");
    f4 c[NC > 0 ? NC : 1];
    float t0, t1;
};
template <int D>
__device__ __forceinline__ KFrag<D> kfrag_load(const float* row, int rg) {
    KFrag<D> f;
#pragma unroll
    for (int c = 0; c < KFrag<D>::NC; ++c) f.c[c] = *reinterpret_cast<const f4*>(row + 16 * c + 4 * rg);
    f.t0 = f.t1 = 0.f;
    if (KFrag<D>::TAIL) {
        const float2 t = *reinterpret_cast<const float2*>(row + 16 * KFrag<D>::NC + 2 * rg);
        f.t0 = t.x; f.t1 = t.y;
    }
    return f;
}
// BF (precision mode bf16): operands rounded to bf16 in registers, one v_mfma_f32_16x16x16_bf16 per 16-wide chunk and one for the
// 8-wide tail (both operands carry their two tail values in slots 0, 1 and zeros in 2, 3: a contraction only needs the two
// operands to agree on which slot holds which k).
template <int D, int BF = 0>
__device__ __forceinline__ f4 kfrag_mfma(const KFrag<D>& a, const KFrag<D>& b, f4 acc) {
    if constexpr (BF != 0) {
#pragma unroll
        for (int c = 0; c < KFrag<D>::NC; ++c) acc = mfma16_16<BF>(pack16<BF>(a.c[c]), pack16<BF>(b.c[c]), acc);
        if (KFrag<D>::TAIL) acc = mfma16_16<BF>(pack16<BF>(f4{a.t0, a.t1, 0.f, 0.f}), pack16<BF>(f4{b.t0, b.t1, 0.f, 0.f}), acc);
        return acc;
    } else {
#pragma unroll
    for (int c = 0; c < KFrag<D>::NC; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = mfma16(a.c[c][j], b.c[c][j], acc);
    if (KFrag<D>::TAIL) { acc = mfma16(a.t0, b.t0, acc); acc = mfma16(a.t1, b.t1, acc); }
    return acc;
    }
}

template <int PT, int D, int HG, int BF = 0>
__global__ __launch_bounds__(64 * PT * HG) void attn_fwd_lds_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                      float* __restrict__ lse, AttnGeom g, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // head dim D is a template parameter: every LDS offset below is a compile-time immediate off one base register
    constexpr int NTHR = 64 * PT * HG, TOK = 16 * PT, DCH = (D + 15) / 16, d = D;
    constexpr int S = HG * 3 * D + 4, F = HG * 3 * D / 4;
    __shared__ long srow[TOK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, rg = lane >> 4;
    const int hl = wave / PT, qt = wave - hl * PT;
    const int ngrp = g.heads / HG;
    const int p = blockIdx.x / ngrp, h0 = (blockIdx.x - p * ngrp) * HG;
    const int P = g.ph * g.pw;
    const long ld = 3L * g.C;
    if (tid < TOK) srow[tid] = token_row(g, p, tid);
    __syncthreads();
    {   // all global loads of the thread are issued before the first LDS store: a load -> store loop waits for every load in
        // turn (4-5 dependent HBM round trips per workgroup)
        // (the qkv format is tested ONCE, around the whole staging block, and the loads are unconditional on clamped rows: with
        // the test and the bf16 unpack behind each load the compiler waited for every load in turn)
        constexpr int NL = (TOK * F + NTHR - 1) / NTHR;
        auto stage_qkv = [&](auto q16) {
            constexpr bool Q16 = decltype(q16)::value;
            f4 stage[Q16 ? 1 : NL]; u2_ stageh[Q16 ? NL : 1];
            unsigned ok = 0;
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const int e = tid + j * NTHR, tok = min(e / F, TOK - 1), f = e - (e / F) * F;
                const long row = srow[tok];
                ok |= (unsigned)(e < TOK * F && row >= 0) << j;
                const long off = max(row, 0L) * ld + h0 * 3 * d + 4 * f;
                if constexpr (Q16) stageh[j] = *reinterpret_cast<const u2_*>(reinterpret_cast<const unsigned short*>(qkv) + off);
                else stage[j] = ld4(qkv + off);
            }
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const int e = tid + j * NTHR, tok = e / F, f = e - tok * F;
                f4 v;
                if constexpr (Q16) v = unpack_bf16(__builtin_bit_cast(s4, stageh[j])); else v = stage[j];
                if (e < TOK * F) *reinterpret_cast<f4*>(smem + tok * S + 4 * f) = (ok >> j) & 1u ? v : zero4();
            }
        };
        if (g.fmt & 1) stage_qkv(std::true_type{}); else stage_qkv(std::false_type{});
    }
    __syncthreads();
    const float* hb = smem + hl * 3 * d;                     // this wave's head inside a staged token row
    const bool qvalid = 16 * qt + i < P;
    const KFrag<D> qf = kfrag_load<D>(hb + (16 * qt + i) * S, rg);
    f4 s[PT];
#pragma unroll
    for (int mt = 0; mt < PT; ++mt) s[mt] = kfrag_mfma<D, BF>(kfrag_load<D>(hb + (16 * mt + i) * S + d, rg), qf, zero4());
    float mx = -INFINITY;
#pragma unroll
    for (int mt = 0; mt < PT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * mt + 4 * rg + r;
            const float v = key < P ? s[mt][r] * scale : -INFINITY;
            s[mt][r] = v;
            mx = fmaxf(mx, v);
        }
    mx = quad16_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int mt = 0; mt < PT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = fast_exp(s[mt][r] - mx);          // v_exp_f32; exp(-inf) = 0 for padded keys
            s[mt][r] = e;
            sum += e;
        }
    sum = quad16_sum(sum);
    const float inv = 1.0f / sum;
    if (lse && rg == 0 && qvalid) lse[srow[16 * qt + i] * g.heads + h0 + hl] = mx + logf(sum);
    f4 o[DCH];
#pragma unroll
    for (int ct = 0; ct < DCH; ++ct) o[ct] = zero4();
    if constexpr (BF) {
        // P V: lane (i, rg) holds P[query i][keys 4rg .. +3] (its four accumulator rows) and gathers V[keys 4rg .. +3][col i]:
        // ONE bf16 MFMA per (key tile, column tile) instead of four fp32 ones
#pragma unroll
        for (int mt = 0; mt < PT; ++mt) {
            const s4 pa = pack16_raw<BF>(s[mt] * inv);
            const float* vrow = hb + (16 * mt + 4 * rg) * S + 2 * d;
#pragma unroll
            for (int ct = 0; ct < DCH; ++ct) {
                const bool ok = 16 * ct + i < d;
                const f4 vv = ok ? f4{vrow[16 * ct + i], vrow[S + 16 * ct + i], vrow[2 * S + 16 * ct + i], vrow[3 * S + 16 * ct + i]} : zero4();
                o[ct] = mfma16_16<BF>(pa, pack16<BF>(vv), o[ct]);
            }
        }
    } else {
#pragma unroll
    for (int mt = 0; mt < PT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* vrow = hb + (16 * mt + 4 * rg + r) * S + 2 * d;
#pragma unroll
            for (int ct = 0; ct < DCH; ++ct) {
                const float vv = 16 * ct + i < d ? vrow[16 * ct + i] : 0.f;
                o[ct] = mfma16(s[mt][r] * inv, vv, o[ct]);
            }
        }
    }
    // O tile -> this wave's own q slots (nobody else reads them) -> per-token d-float segments
    float* ob = smem + hl * 3 * d + (16 * qt) * S;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ct = 0; ct < DCH; ++ct)
            if (16 * ct + i < d) ob[(4 * rg + r) * S + 16 * ct + i] = o[ct][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int d4 = D / 4;
    for (int e = lane; e < 16 * d4; e += 64) {
        const int tok = e / d4, c4 = e - tok * d4;
        const long row = srow[16 * qt + tok];
        if (row >= 0) *reinterpret_cast<f4*>(out + row * g.C + (h0 + hl) * d + 4 * c4) = *reinterpret_cast<const f4*>(ob + tok * S + 4 * c4);
    }
}

// fused backward: phase 1 = query-owned (dQ, D), phase 2 = key-owned (dK, dV), then one coalesced store of the whole
// [dq | dk | dv] segment of every token.  Probabilities are recomputed from lse in both phases.
template <int PT, int D, int HG, bool BF = false>
__global__ __launch_bounds__(64 * PT * HG) void attn_bwd_lds_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                      const float* __restrict__ lse, float* __restrict__ dqkv,
                                                                      AttnGeom g, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NTHR = 64 * PT * HG, TOK = 16 * PT, DCH = (D + 15) / 16, d = D;
    constexpr int S = HG * 3 * D + 4, F = HG * 3 * D / 4;
    constexpr int Sd = HG * D + 4, Fd = HG * D / 4;
    __shared__ long srow[TOK];
    __shared__ float sL[HG * TOK], sD[HG * TOK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, rg = lane >> 4;
    const int hl = wave / PT, qt = wave - hl * PT;            // also the key tile of phase 2
    const int ngrp = g.heads / HG;
    const int p = blockIdx.x / ngrp, h0 = (blockIdx.x - p * ngrp) * HG;
    const int P = g.ph * g.pw;
    const long ld = 3L * g.C;
    float* sdo = smem + TOK * S;
    if (tid < TOK) srow[tid] = token_row(g, p, tid);
    __syncthreads();
    {   // every global load of the thread first, then the LDS stores (see the forward kernel)
        constexpr int NL = (TOK * F + NTHR - 1) / NTHR, NLd = (TOK * Fd + NTHR - 1) / NTHR, NLl = (HG * TOK + NTHR - 1) / NTHR;
        auto stage_all = [&](auto q16) {                      // format test hoisted around the block, unconditional clamped loads (see fwd)
            constexpr bool Q16 = decltype(q16)::value;
            f4 stage[Q16 ? 1 : NL], staged[NLd]; u2_ stageh[Q16 ? NL : 1];
            float stagel[NLl];
            unsigned ok = 0, okd = 0, okl = 0;
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const int e = tid + j * NTHR, tok = min(e / F, TOK - 1), f = e - (e / F) * F;
                const long row = srow[tok];
                ok |= (unsigned)(e < TOK * F && row >= 0) << j;
                const long off = max(row, 0L) * ld + h0 * 3 * d + 4 * f;
                if constexpr (Q16) stageh[j] = *reinterpret_cast<const u2_*>(reinterpret_cast<const unsigned short*>(qkv) + off);
                else stage[j] = ld4(qkv + off);
            }
#pragma unroll
            for (int j = 0; j < NLd; ++j) {
                const int e = tid + j * NTHR, tok = min(e / Fd, TOK - 1), f = e - (e / Fd) * Fd;
                const long row = srow[tok];
                okd |= (unsigned)(e < TOK * Fd && row >= 0) << j;
                staged[j] = ld4(dout + max(row, 0L) * g.C + h0 * d + 4 * f);
            }
#pragma unroll
            for (int j = 0; j < NLl; ++j) {
                const int e = tid + j * NTHR, hh = min(e / TOK, HG - 1), tok = e - (e / TOK) * TOK;
                const long row = srow[tok];
                okl |= (unsigned)(e < HG * TOK && row >= 0) << j;
                stagel[j] = lse[max(row, 0L) * g.heads + h0 + hh];
            }
#pragma unroll
            for (int j = 0; j < NL; ++j) {
                const int e = tid + j * NTHR, tok = e / F, f = e - tok * F;
                f4 v;
                if constexpr (Q16) v = unpack_bf16(__builtin_bit_cast(s4, stageh[j])); else v = stage[j];
                if (e < TOK * F) *reinterpret_cast<f4*>(smem + tok * S + 4 * f) = (ok >> j) & 1u ? v : zero4();
            }
#pragma unroll
            for (int j = 0; j < NLd; ++j) {
                const int e = tid + j * NTHR, tok = e / Fd, f = e - tok * Fd;
                if (e < TOK * Fd) *reinterpret_cast<f4*>(sdo + tok * Sd + 4 * f) = (okd >> j) & 1u ? staged[j] : zero4();
            }
#pragma unroll
            for (int j = 0; j < NLl; ++j) {
                const int e = tid + j * NTHR;
                if (e < HG * TOK) sL[e] = (okl >> j) & 1u ? stagel[j] : 0.f;
            }
        };
        if (g.fmt & 1) stage_all(std::true_type{}); else stage_all(std::false_type{});
    }
    __syncthreads();
    const float* hb = smem + hl * 3 * d;
    const float* db = sdo + hl * d;
    // ---- phase 1: dQ and D of query tile qt ------------------------------------------------------------------------------
    f4 dq[DCH];
    {
        const bool qvalid = 16 * qt + i < P;
        const KFrag<D> qf = kfrag_load<D>(hb + (16 * qt + i) * S, rg), dof = kfrag_load<D>(db + (16 * qt + i) * Sd, rg);
        const float l = sL[hl * TOK + 16 * qt + i];
        f4 s[PT], dp[PT];
#pragma unroll
        for (int mt = 0; mt < PT; ++mt) {
            s[mt] = kfrag_mfma<D, BF>(kfrag_load<D>(hb + (16 * mt + i) * S + d, rg), qf, zero4());
            dp[mt] = kfrag_mfma<D, BF>(kfrag_load<D>(hb + (16 * mt + i) * S + 2 * d, rg), dof, zero4());
        }
        float Dq = 0.f;
#pragma unroll
        for (int mt = 0; mt < PT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * mt + 4 * rg + r;
                const float pr = (key < P && qvalid) ? fast_exp(s[mt][r] * scale - l) : 0.f;
                s[mt][r] = pr;
                Dq += pr * dp[mt][r];
            }
        Dq = quad16_sum(Dq);
        if (rg == 0) sD[hl * TOK + 16 * qt + i] = Dq;
#pragma unroll
        for (int ct = 0; ct < DCH; ++ct) dq[ct] = zero4();
        if constexpr (BF) {
#pragma unroll
            for (int mt = 0; mt < PT; ++mt) {
                f4 ds;
#pragma unroll
                for (int r = 0; r < 4; ++r) ds[r] = s[mt][r] * (dp[mt][r] - Dq) * scale;
                const s4 pa = pack_bf16(ds);
                const float* krow = hb + (16 * mt + 4 * rg) * S + d;
#pragma unroll
                for (int ct = 0; ct < DCH; ++ct) {
                    const bool ok = 16 * ct + i < d;
                    const f4 kk = ok ? f4{krow[16 * ct + i], krow[S + 16 * ct + i], krow[2 * S + 16 * ct + i], krow[3 * S + 16 * ct + i]} : zero4();
                    dq[ct] = mfma16_bf16(pa, pack_bf16(kk), dq[ct]);
                }
            }
        } else {
#pragma unroll
        for (int mt = 0; mt < PT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* krow = hb + (16 * mt + 4 * rg + r) * S + d;
                const float ds = s[mt][r] * (dp[mt][r] - Dq) * scale;
#pragma unroll
                for (int ct = 0; ct < DCH; ++ct) {
                    const float kk = 16 * ct + i < d ? krow[16 * ct + i] : 0.f;
                    dq[ct] = mfma16(ds, kk, dq[ct]);
                }
            }
        }
    }
    __syncthreads();                                          // D of every query of the partition is in LDS
    // ---- phase 2: dK, dV of key tile kt = qt --------------------------------------------------------------------------------
    f4 dk[DCH], dv[DCH];
    {
        const bool kvalid = 16 * qt + i < P;
        const KFrag<D> kf = kfrag_load<D>(hb + (16 * qt + i) * S + d, rg), vf = kfrag_load<D>(hb + (16 * qt + i) * S + 2 * d, rg);
#pragma unroll
        for (int ct = 0; ct < DCH; ++ct) { dk[ct] = zero4(); dv[ct] = zero4(); }
#pragma unroll
        for (int qm = 0; qm < PT; ++qm) {
            const f4 s = kfrag_mfma<D, BF>(kfrag_load<D>(hb + (16 * qm + i) * S, rg), kf, zero4());
            const f4 dp = kfrag_mfma<D, BF>(kfrag_load<D>(db + (16 * qm + i) * Sd, rg), vf, zero4());
            if constexpr (BF) {
                f4 pr4 = zero4(), ds4 = zero4();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int query = 16 * qm + 4 * rg + r;
                    if (query < P && kvalid) {
                        pr4[r] = fast_exp(s[r] * scale - sL[hl * TOK + query]);
                        ds4[r] = pr4[r] * (dp[r] - sD[hl * TOK + query]) * scale;
                    }
                }
                const s4 ppr = pack_bf16(pr4), pds = pack_bf16(ds4);
                const float* qrow = hb + (16 * qm + 4 * rg) * S;
                const float* dorow = db + (16 * qm + 4 * rg) * Sd;
#pragma unroll
                for (int ct = 0; ct < DCH; ++ct) {
                    const bool ok = 16 * ct + i < d;
                    const int cc = 16 * ct + i;
                    const f4 dov = ok ? f4{dorow[cc], dorow[Sd + cc], dorow[2 * Sd + cc], dorow[3 * Sd + cc]} : zero4();
                    const f4 qv = ok ? f4{qrow[cc], qrow[S + cc], qrow[2 * S + cc], qrow[3 * S + cc]} : zero4();
                    dv[ct] = mfma16_bf16(ppr, pack_bf16(dov), dv[ct]);
                    dk[ct] = mfma16_bf16(pds, pack_bf16(qv), dk[ct]);
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int query = 16 * qm + 4 * rg + r;       // accumulator row r; key = column i
                float pr = 0.f, ds = 0.f;
                if (query < P && kvalid) {
                    pr = fast_exp(s[r] * scale - sL[hl * TOK + query]);
                    ds = pr * (dp[r] - sD[hl * TOK + query]) * scale;
                }
                const float* qrow = hb + query * S;
                const float* dorow = db + query * Sd;
#pragma unroll
                for (int ct = 0; ct < DCH; ++ct) {
                    const bool ok = 16 * ct + i < d;
                    const float dov = ok ? dorow[16 * ct + i] : 0.f;
                    const float qv = ok ? qrow[16 * ct + i] : 0.f;
                    dv[ct] = mfma16(pr, dov, dv[ct]);
                    dk[ct] = mfma16(ds, qv, dk[ct]);
                }
            }
        }
    }
    __syncthreads();                                          // every wave is done reading the staged q/k/v
    float* gb = smem + hl * 3 * d + (16 * qt) * S;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ct = 0; ct < DCH; ++ct)
            if (16 * ct + i < d) {
                float* t = gb + (4 * rg + r) * S + 16 * ct + i;
                t[0] = dq[ct][r]; t[d] = dk[ct][r]; t[2 * d] = dv[ct][r];
                // keeps the dq stores of the two column tiles apart: hipcc (ROCm 7.2) merged them into ds_write2_b32 and, for the
                // one-head d = 32 instantiation (row stride 100 dwords), emitted offset0:25 / :50 instead of :100 / :200 for accumulator
                // rows 2 and 3 (tools/attn_pad_diag.py: dq of tokens 2, 3 (mod 4) landed a quarter row into the previous rows) --
                // the "padded one-head partitions" defect of round 1 was this instantiation, not the padding
                asm volatile("" ::: "memory");
            }
    __syncthreads();
    for (int e = tid; e < TOK * F; e += NTHR) {
        const int tok = e / F, f = e - tok * F;
        const long row = srow[tok];
        if (row >= 0) {
            const f4 v = *reinterpret_cast<const f4*>(smem + tok * S + 4 * f);
            if (g.fmt & 2) *reinterpret_cast<s4*>(reinterpret_cast<unsigned short*>(dqkv) + row * ld + h0 * 3 * d + 4 * f) = pack_bf16(v);
            else *reinterpret_cast<f4*>(dqkv + row * ld + h0 * 3 * d + 4 * f) = v;
        }
    }
}

// =====================================================================================================================
// bf16-tile variants (round 3, precision mode bf16 with bf16 qkv rows in HBM): q / k / v (and dO) are staged ONCE as bf16 -- the fp32
// tiles above were re-packed to bf16 for every MFMA operand (2 v_cvt_pk per fragment, 4 ds_read_b32 + 2 v_cvt_pk per transposed
// operand) and kept 15-19 % of the LDS cycles in bank conflicts.
//   * rows of SD dwords with SD % 16 == 8 (72 for two heads of d = 24, no padding): the 16-byte head-dimension fragments of 16 rows
//     (ds_read_b128, 8 bf16 = k 8rg .. 8rg+7 of ONE v_mfma_f32_16x16x32_bf16: d = 24 / 32 is a single MFMA instead of chunk + tail)
//     fall on 16 distinct 4-bank slots of their service group, and the 8 rows x 32 bytes of a half-wave of ds_read_b64_tr_b16 (the
//     operands that contract over TOKENS: V in P V, K in dS K, dO and Q in the key-owned pass) on 8 distinct 8-bank bins: both read
//     patterns are conflict-free in the same un-swizzled layout;
//   * d = 24: lanes rg = 3 of a head-dimension fragment hold k = 24 .. 31, i.e. the first 8 values of the NEXT part of the row --
//     zeroed on ONE operand of every product (the wave's own q / dO / k / v fragment, loaded once), the streamed operand stays raw;
//   * 32 KB of LDS per (partition, two heads) instead of 64: three 10-wave workgroups per CU.
// =====================================================================================================================
template <int D, int HG>
struct A16L {
    static constexpr int RW = HG * 3 * D, RWD = HG * D;                     // bf16 elements of a staged qkv / dO row
    static constexpr int SD = (RW / 2 + 7) / 16 * 16 + 8;                   // row strides in dwords, == 8 (mod 16)
    static constexpr int SDD = (RWD / 2 + 7) / 16 * 16 + 8;
    static_assert(RW % 8 == 0 && RWD % 8 == 0 && SD >= RW / 2 && SDD >= RWD / 2, "rows are staged in units of 8 elements");
};
typedef unsigned u4v_ __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s4 lds_s4_;
// head-dimension fragment: 8 bf16 of row `row` starting at dword `dw` (+ 4 rg)
__device__ __forceinline__ s8v a16_frag(const unsigned* base, int row, int sd, int dw, int rg) {
    return __builtin_bit_cast(s8v, *reinterpret_cast<const u4v_*>(base + row * sd + dw + 4 * rg));
}
template <int D> __device__ __forceinline__ s8v a16_mask(s8v v, int rg) {   // d = 24: k = 24 .. 31 belongs to the next part of the row
    if (D == 24 && rg == 3) v = s8v{0, 0, 0, 0, 0, 0, 0, 0};
    return v;
}
// token-dimension fragment: element [4 rg + j][16-column tile at dword dw] of the 16-row tile starting at row0, for j = 0 .. 3
__device__ __forceinline__ s4 a16_tfrag(const unsigned* base, int row0, int sd, int dw, int i, int rg) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_*)(base + (row0 + 4 * rg + (i >> 2)) * sd + dw + 2 * (i & 3)));
}

// stage rows of 8-element units: src rows are bf16 (SRC16) or fp32; all global loads first, then the LDS stores.  The padding units of a
// row (SDW / 4 > UNITS) and the 8 slack dwords behind the last row are ZEROED: the fragments of the last part of a row read 8 elements
// past it, and although those values only ever meet a zeroed operand or feed discarded output columns, 0 * NaN is NaN.
// CVT (SRC16 only): the source rows are fp16 (forward-stored tensors of precision mode 16f) and are re-rounded to the bf16 the gradient
// MFMAs take while they are stored to LDS
template <int NTHR, int TOK, int UNITS, int SDW, bool SRC16, bool CVT = false>
struct A16Raw {                                                 // the global loads of one staged operand, still in registers
    static constexpr int UR = SDW / 4;                          // units per staged row, padding included
    static constexpr int NL = (TOK * UR + NTHR - 1) / NTHR;
    u4v_ raw16[SRC16 ? NL : 1]; f4 raw32[SRC16 ? 1 : 2 * NL];
    unsigned ok;
    __device__ __forceinline__ void load(const void* src, const long* srow, long ld, long col0, int tid) {
        ok = 0;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int e = tid + j * NTHR, tok = min(e / UR, TOK - 1), f = e - (e / UR) * UR;
            const long row = srow[tok];
            ok |= (unsigned)(e < TOK * UR && f < UNITS && row >= 0) << j;
            const long off = max(row, 0L) * ld + col0 + 8 * min(f, UNITS - 1);
            if constexpr (SRC16) raw16[j] = *reinterpret_cast<const u4v_*>(reinterpret_cast<const unsigned short*>(src) + off);
            else { raw32[2 * j] = ld4(reinterpret_cast<const float*>(src) + off); raw32[2 * j + 1] = ld4(reinterpret_cast<const float*>(src) + off + 4); }
        }
    }
    __device__ __forceinline__ void store(unsigned* dst, int tid) const {
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int e = tid + j * NTHR, tok = e / UR, f = e - tok * UR;
            u4v_ v;
            if constexpr (SRC16 && CVT) {
                const u2_ lo = __builtin_bit_cast(u2_, h16_to_bf16(__builtin_bit_cast(s4, u2_{raw16[j].x, raw16[j].y})));
                const u2_ hi = __builtin_bit_cast(u2_, h16_to_bf16(__builtin_bit_cast(s4, u2_{raw16[j].z, raw16[j].w})));
                v = u4v_{lo.x, lo.y, hi.x, hi.y};
            } else if constexpr (SRC16) v = raw16[j];
            else {
                const s4 lo = pack_bf16(raw32[2 * j]), hi = pack_bf16(raw32[2 * j + 1]);
                const u2_ a = __builtin_bit_cast(u2_, lo), b = __builtin_bit_cast(u2_, hi);
                v = u4v_{a.x, a.y, b.x, b.y};
            }
            if (e < TOK * UR) *reinterpret_cast<u4v_*>(dst + tok * SDW + 4 * f) = (ok >> j) & 1u ? v : u4v_{0u, 0u, 0u, 0u};
        }
        if (tid < 8) dst[TOK * SDW + tid] = 0u;
    }
};
template <int NTHR, int TOK, int UNITS, int SDW, bool SRC16>
__device__ __forceinline__ void a16_stage(unsigned* dst, const void* src, const long* srow, long ld, long col0, int tid) {
    A16Raw<NTHR, TOK, UNITS, SDW, SRC16> r;
    r.load(src, srow, ld, col0, tid);
    r.store(dst, tid);
}

// OF: 1 = bf16 rows and operands, 2 = fp16 rows and operands (precision mode 16f: q, k, v, P and the stored O in the reference's autocast dtype)
template <int PT, int D, int HG, int OF = 1>
__global__ __launch_bounds__(64 * PT * HG) void attn_fwd_lds16_kernel(const void* __restrict__ qkv, float* __restrict__ out,
                                                                        float* __restrict__ lse, AttnGeom g, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned smem16[];
    constexpr int NTHR = 64 * PT * HG, TOK = 16 * PT, DCH = (D + 15) / 16, d = D;
    constexpr int SD = A16L<D, HG>::SD, SO = D + 4;
    __shared__ long srow[TOK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, rg = lane >> 4;
    const int hl = wave / PT, qt = wave - hl * PT;
    const int ngrp = g.heads / HG;
    const int p = blockIdx.x / ngrp, h0 = (blockIdx.x - p * ngrp) * HG;
    const int P = g.ph * g.pw;
    unsigned* sq = smem16;
    float* sout = reinterpret_cast<float*>(smem16 + TOK * SD + 8) + wave * 16 * SO;      // wave-private O tile
    if (tid < TOK) srow[tid] = token_row(g, p, tid);
    __syncthreads();
    a16_stage<NTHR, TOK, A16L<D, HG>::RW / 8, SD, true>(sq, qkv, srow, 3L * g.C, (long)h0 * 3 * d, tid);
    __syncthreads();
    const int hq = hl * 3 * d / 2, hk = hq + d / 2, hv = hq + d;                          // dword offsets of this head's q | k | v
    const bool qvalid = 16 * qt + i < P;
    const s8v qf = a16_mask<D>(a16_frag(sq, 16 * qt + i, SD, hq, rg), rg);
    f4 s[PT];
#pragma unroll
    for (int mt = 0; mt < PT; ++mt) s[mt] = mfma32_16<OF>(a16_frag(sq, 16 * mt + i, SD, hk, rg), qf, zero4());
    float mx = -INFINITY;
#pragma unroll
    for (int mt = 0; mt < PT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * mt + 4 * rg + r;
            const float v = key < P ? s[mt][r] * scale : -INFINITY;
            s[mt][r] = v;
            mx = fmaxf(mx, v);
        }
    mx = quad16_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int mt = 0; mt < PT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = fast_exp(s[mt][r] - mx);
            s[mt][r] = e;
            sum += e;
        }
    sum = quad16_sum(sum);
    const float inv = 1.0f / sum;
    if (lse && rg == 0 && qvalid) lse[srow[16 * qt + i] * g.heads + h0 + hl] = mx + logf(sum);
    f4 o[DCH];
#pragma unroll
    for (int ct = 0; ct < DCH; ++ct) o[ct] = zero4();
#pragma unroll
    for (int mt = 0; mt < PT; ++mt) {
        const s4 pa = pack16_raw<OF>(s[mt] * inv);
#pragma unroll
        for (int ct = 0; ct < DCH; ++ct) o[ct] = mfma16_16<OF>(pa, a16_tfrag(sq, 16 * mt, SD, hv + 8 * ct, i, rg), o[ct]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ct = 0; ct < DCH; ++ct)
            if (16 * ct + i < d) sout[(4 * rg + r) * SO + 16 * ct + i] = o[ct][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int d4 = D / 4;
    for (int e = lane; e < 16 * d4; e += 64) {
        const int tok = e / d4, c4 = e - tok * d4;
        const long row = srow[16 * qt + tok];
        if (row < 0) continue;
        const f4 v = *reinterpret_cast<const f4*>(sout + tok * SO + 4 * c4);
        if (g.fmt & 4) *reinterpret_cast<s4*>(reinterpret_cast<unsigned short*>(out) + row * g.C + (h0 + hl) * d + 4 * c4) = pack16<OF>(v);
        else *reinterpret_cast<f4*>(out + row * g.C + (h0 + hl) * d + 4 * c4) = v;
    }
}

// QH: the qkv rows are fp16 (precision mode 16f) -- re-rounded to bf16 while staged; every contraction of the backward pass takes bf16 operands
template <int PT, int D, int HG, bool QH = false>
__global__ __launch_bounds__(64 * PT * HG) void attn_bwd_lds16_kernel(const void* __restrict__ qkv, const float* __restrict__ dout,
                                                                        const float* __restrict__ lse, void* __restrict__ dqkv,
                                                                        AttnGeom g, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned smem16[];
    constexpr int NTHR = 64 * PT * HG, TOK = 16 * PT, DCH = (D + 15) / 16, d = D;
    constexpr int SD = A16L<D, HG>::SD, SDD = A16L<D, HG>::SDD, F8 = A16L<D, HG>::RW / 8;
    __shared__ long srow[TOK];
    __shared__ float sL[HG * TOK], sD[HG * TOK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, rg = lane >> 4;
    const int hl = wave / PT, qt = wave - hl * PT;            // also the key tile of phase 2
    const int ngrp = g.heads / HG;
    const int p = blockIdx.x / ngrp, h0 = (blockIdx.x - p * ngrp) * HG;
    const int P = g.ph * g.pw;
    const long ld = 3L * g.C;
    unsigned* sq = smem16;
    unsigned* sdo = smem16 + TOK * SD + 8;
    if (tid < TOK) srow[tid] = token_row(g, p, tid);
    __syncthreads();
    {   // lse first (its loads then fly with the rest), then qkv and dO rows
        constexpr int NLl = (HG * TOK + NTHR - 1) / NTHR;
        float stagel[NLl]; unsigned okl = 0;
#pragma unroll
        for (int j = 0; j < NLl; ++j) {
            const int e = tid + j * NTHR, hh = min(e / TOK, HG - 1), tok = e - (e / TOK) * TOK;
            const long row = srow[tok];
            okl |= (unsigned)(e < HG * TOK && row >= 0) << j;
            stagel[j] = lse[max(row, 0L) * g.heads + h0 + hh];
        }
        // every global load of the workgroup's operands before the first LDS store (qkv staged, THEN dO loaded cost a second round trip)
        A16Raw<NTHR, TOK, F8, SD, true, QH> rq;
        rq.load(qkv, srow, ld, (long)h0 * 3 * d, tid);
        if (g.fmt & 8) {
            A16Raw<NTHR, TOK, A16L<D, HG>::RWD / 8, SDD, true> rd;
            rd.load(dout, srow, (long)g.C, (long)h0 * d, tid);
            rq.store(sq, tid); rd.store(sdo, tid);
        } else {
            A16Raw<NTHR, TOK, A16L<D, HG>::RWD / 8, SDD, false> rd;
            rd.load(dout, srow, (long)g.C, (long)h0 * d, tid);
            rq.store(sq, tid); rd.store(sdo, tid);
        }
#pragma unroll
        for (int j = 0; j < NLl; ++j) {
            const int e = tid + j * NTHR;
            if (e < HG * TOK) sL[e] = (okl >> j) & 1u ? stagel[j] : 0.f;
        }
    }
    __syncthreads();
    const int hq = hl * 3 * d / 2, hk = hq + d / 2, hv = hq + d, hd = hl * d / 2;       // dword offsets: q | k | v of this head, its dO slice
    // ---- phase 1: dQ and D of query tile qt ------------------------------------------------------------------------------
    f4 dq[DCH];
    {
        const bool qvalid = 16 * qt + i < P;
        const s8v qf = a16_mask<D>(a16_frag(sq, 16 * qt + i, SD, hq, rg), rg);
        const s8v dof = a16_mask<D>(a16_frag(sdo, 16 * qt + i, SDD, hd, rg), rg);
        const float l = sL[hl * TOK + 16 * qt + i];
        f4 s[PT], dp[PT];
#pragma unroll
        for (int mt = 0; mt < PT; ++mt) {
            s[mt] = mfma32_bf16(a16_frag(sq, 16 * mt + i, SD, hk, rg), qf, zero4());
            dp[mt] = mfma32_bf16(a16_frag(sq, 16 * mt + i, SD, hv, rg), dof, zero4());
        }
        float Dq = 0.f;
#pragma unroll
        for (int mt = 0; mt < PT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * mt + 4 * rg + r;
                const float pr = (key < P && qvalid) ? fast_exp(s[mt][r] * scale - l) : 0.f;
                s[mt][r] = pr;
                Dq += pr * dp[mt][r];
            }
        Dq = quad16_sum(Dq);
        if (rg == 0) sD[hl * TOK + 16 * qt + i] = Dq;
#pragma unroll
        for (int ct = 0; ct < DCH; ++ct) dq[ct] = zero4();
#pragma unroll
        for (int mt = 0; mt < PT; ++mt) {
            f4 ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) ds[r] = s[mt][r] * (dp[mt][r] - Dq) * scale;
            const s4 pa = pack_bf16(ds);
#pragma unroll
            for (int ct = 0; ct < DCH; ++ct) dq[ct] = mfma16_bf16(pa, a16_tfrag(sq, 16 * mt, SD, hk + 8 * ct, i, rg), dq[ct]);
        }
    }
    __syncthreads();                                          // D of every query of the partition is in LDS
    // ---- phase 2: dK, dV of key tile kt = qt --------------------------------------------------------------------------------
    f4 dk[DCH], dv[DCH];
    {
        const bool kvalid = 16 * qt + i < P;
        const s8v kf = a16_mask<D>(a16_frag(sq, 16 * qt + i, SD, hk, rg), rg);
        const s8v vf = a16_mask<D>(a16_frag(sq, 16 * qt + i, SD, hv, rg), rg);
#pragma unroll
        for (int ct = 0; ct < DCH; ++ct) { dk[ct] = zero4(); dv[ct] = zero4(); }
#pragma unroll
        for (int qm = 0; qm < PT; ++qm) {
            const f4 s = mfma32_bf16(a16_frag(sq, 16 * qm + i, SD, hq, rg), kf, zero4());
            const f4 dp = mfma32_bf16(a16_frag(sdo, 16 * qm + i, SDD, hd, rg), vf, zero4());
            f4 pr4 = zero4(), ds4 = zero4();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int query = 16 * qm + 4 * rg + r;       // accumulator row r; key = column i
                if (query < P && kvalid) {
                    pr4[r] = fast_exp(s[r] * scale - sL[hl * TOK + query]);
                    ds4[r] = pr4[r] * (dp[r] - sD[hl * TOK + query]) * scale;
                }
            }
            const s4 ppr = pack_bf16(pr4), pds = pack_bf16(ds4);
#pragma unroll
            for (int ct = 0; ct < DCH; ++ct) {
                dv[ct] = mfma16_bf16(ppr, a16_tfrag(sdo, 16 * qm, SDD, hd + 8 * ct, i, rg), dv[ct]);
                dk[ct] = mfma16_bf16(pds, a16_tfrag(sq, 16 * qm, SD, hq + 8 * ct, i, rg), dk[ct]);
            }
        }
    }
    __syncthreads();                                          // every wave is done reading the staged q / k / v
    // [dq | dk | dv] of this wave's 16 tokens as bf16 into its own slots of the qkv tile, then one coalesced copy-out of whole rows
    unsigned short* gb = reinterpret_cast<unsigned short*>(sq) + (16 * qt) * (2 * SD) + hl * 3 * d;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ct = 0; ct < DCH; ++ct)
            if (16 * ct + i < d) {
                unsigned short* t = gb + (4 * rg + r) * (2 * SD) + 16 * ct + i;
                const s4 pk = pack_bf16(f4{dq[ct][r], dk[ct][r], dv[ct][r], 0.f});
                t[0] = (unsigned short)pk[0]; t[d] = (unsigned short)pk[1]; t[2 * d] = (unsigned short)pk[2];
                asm volatile("" ::: "memory");                // keep the 16-bit stores apart (see the ds_write2 defect above)
            }
    __syncthreads();
    for (int e = tid; e < TOK * F8; e += NTHR) {
        const int tok = e / F8, f = e - tok * F8;
        const long row = srow[tok];
        if (row >= 0)
            *reinterpret_cast<u4v_*>(reinterpret_cast<unsigned short*>(dqkv) + row * ld + h0 * 3 * d + 8 * f) =
                *reinterpret_cast<const u4v_*>(sq + tok * SD + 4 * f);
    }
}

template <int PT, int D, int HG>
static int run_attn_lds(int which, const float* qkv, const float* dout, float* out, float* lse, float* dqkv,
                        const AttnGeom& g, float scale, hipStream_t s) {
    const int NP = g.B * (g.H / g.ph) * (g.W / g.pw);
    const int nblk = NP * (g.heads / HG);
    if (nblk == 0) return LEOD_OK;
    const int TOK = 16 * PT, S = HG * 3 * D + 4, Sd = HG * D + 4;
    // precision mode bf16 with bf16 qkv rows (and bf16 dqkv): the bf16-tile kernels
    const bool h16 = leod_precision_mode() == 2;               // 16-bit qkv / O rows are fp16
    if (leod_precision() == 1 && which == 0 && (g.fmt & 1)) {
        const size_t lds = ((size_t)TOK * A16L<D, HG>::SD + 8) * 4 + (size_t)PT * HG * 16 * (D + 4) * 4;
        if (h16) hipLaunchKernelGGL((attn_fwd_lds16_kernel<PT, D, HG, 2>), dim3(nblk), dim3(64 * PT * HG), lds, s, qkv, out, lse, g, scale);
        else hipLaunchKernelGGL((attn_fwd_lds16_kernel<PT, D, HG, 1>), dim3(nblk), dim3(64 * PT * HG), lds, s, qkv, out, lse, g, scale);
        return leod_launch_status();
    }
    if (leod_precision() == 1 && which == 1 && (g.fmt & 3) == 3) {
        const size_t lds = ((size_t)TOK * (A16L<D, HG>::SD + A16L<D, HG>::SDD) + 16) * 4;
        if (h16) hipLaunchKernelGGL((attn_bwd_lds16_kernel<PT, D, HG, true>), dim3(nblk), dim3(64 * PT * HG), lds, s, qkv, dout, lse, dqkv, g, scale);
        else hipLaunchKernelGGL((attn_bwd_lds16_kernel<PT, D, HG, false>), dim3(nblk), dim3(64 * PT * HG), lds, s, qkv, dout, lse, dqkv, g, scale);
        return leod_launch_status();
    }
    if ((g.fmt & 12) || (g.fmt && h16)) return LEOD_ERR_UNSUPPORTED;   // 16-bit O / dO rows, fp16 rows: the 16-bit-tile kernels only
    if (which == 0) {
        const size_t lds = (size_t)TOK * S * sizeof(float);
        LEOD_BY_OPFMT(hipLaunchKernelGGL((attn_fwd_lds_kernel<PT, D, HG, OF>), dim3(nblk), dim3(64 * PT * HG), lds, s, qkv, out, lse, g, scale));
    } else {
        const size_t lds = (size_t)TOK * (S + Sd) * sizeof(float);
        if (leod_precision() == 1) hipLaunchKernelGGL((attn_bwd_lds_kernel<PT, D, HG, true>), dim3(nblk), dim3(64 * PT * HG), lds, s, qkv, dout, lse, dqkv, g, scale);
        else hipLaunchKernelGGL((attn_bwd_lds_kernel<PT, D, HG>), dim3(nblk), dim3(64 * PT * HG), lds, s, qkv, dout, lse, dqkv, g, scale);
    }
    return leod_launch_status();
}

// ---------------------------------------------------------------------------------------------------
template <int PT, int DCH>
static int run_attn(int which, const float* qkv, const float* dout, float* out, float* lse, float* dsum, float* dqkv,
                    const AttnGeom& g, float scale, hipStream_t s) {
    const int NP = g.B * (g.H / g.ph) * (g.W / g.pw);
    const int ntasks = NP * g.heads * PT;
    if (ntasks == 0) return LEOD_OK;
    const dim3 grid(cdiv(ntasks, 4)), blk(256);
    if (which == 0) hipLaunchKernelGGL((attn_fwd_kernel<PT, DCH>), grid, blk, 0, s, qkv, out, lse, g, scale, ntasks);
    else if (which == 1) hipLaunchKernelGGL((attn_bwd_q_kernel<PT, DCH>), grid, blk, 0, s, qkv, dout, lse, dqkv, dsum, g, scale, ntasks);
    else hipLaunchKernelGGL((attn_bwd_kv_kernel<PT, DCH>), grid, blk, 0, s, qkv, dout, lse, dsum, dqkv, g, scale, ntasks);
    return leod_launch_status();
}

static int dispatch_attn(int which, const float* qkv, const float* dout, float* out, float* lse, float* dsum,
                         float* dqkv, const AttnGeom& g, hipStream_t s) {
    if (g.C != g.heads * g.d || (g.d & 3) || g.d > 32 || g.H % g.ph || g.W % g.pw) return LEOD_ERR_ARG;
    const int P = g.ph * g.pw, PT = (P + 15) / 16, DCH = (g.d + 15) / 16;
    const float scale = 1.0f / sqrtf((float)g.d);
    constexpr int use_lds = 1;
    // One head per workgroup (5 waves for an 80-token partition) instead of two (10 waves, the CU's wave limit at three workgroups): six
    // workgroups per CU overlap their load / MFMA / store phases better -- backward 984 -> 889 us per step over the four stages, forward of
    // stage 1 126 -> 116 us, the rest equal (tools/kbench.py attn, profiles/r04_z_attn_hg_kbench.txt).  (hg1: bit 0 forward, bit 1 backward.)
    static const int hg1 = 3;
    const int HG = (g.heads % 2 == 0 && 2 * PT <= 16 && !((hg1 & 1) && which == 0) && !((hg1 & 2) && which != 0)) ? 2 : 1;
    static const int force_pad1 = 1;
    // (round 1 routed one-head workgroups with padded partitions to the register-direct
    // backward; the defect behind it was a mis-merged ds_write2_b32 in the <4, 32, 1> instantiation, see the kernel's epilogue)
    const bool lds_shape = (g.d == 24 || g.d == 32) && (PT <= 5 || PT == 8 || (HG == 1 && (PT == 10 || PT == 15))) &&
                           !(HG == 1 && P < 16 * PT && which != 0 && !force_pad1);
    if (use_lds && lds_shape) {
        // which: 0 forward, 1 fused backward (the register-direct path runs 1 = q pass, then 2 = kv pass)
        if (which == 2) return LEOD_OK;                       // the fused LDS backward already produced dK / dV
#define ATTL(PTV, DV, HGV) if (PT == PTV && g.d == DV && HG == HGV) return run_attn_lds<PTV, DV, HGV>(which, qkv, dout, out, lse, dqkv, g, scale, s);
#define ATTD(PTV, HGV) ATTL(PTV, 24, HGV) ATTL(PTV, 32, HGV)
        ATTD(1, 1) ATTD(1, 2) ATTD(2, 1) ATTD(2, 2) ATTD(3, 1) ATTD(3, 2) ATTD(4, 1) ATTD(4, 2) ATTD(5, 1) ATTD(5, 2)
        ATTD(8, 1) ATTD(8, 2) ATTD(10, 1) ATTD(15, 1)
#undef ATTD
#undef ATTL
        return LEOD_ERR_UNSUPPORTED;
    }
    if (g.fmt) return LEOD_ERR_UNSUPPORTED;                    // the register-direct kernels read / write fp32 only
#define ATT(PTV, DV) if (PT == PTV && DCH == DV) return run_attn<PTV, DV>(which, qkv, dout, out, lse, dsum, dqkv, g, scale, s);
    ATT(1, 1) ATT(1, 2) ATT(4, 1) ATT(4, 2) ATT(5, 1) ATT(5, 2) ATT(15, 2) ATT(2, 1) ATT(2, 2) ATT(3, 2) ATT(8, 2) ATT(10, 2)
#undef ATT
    return LEOD_ERR_UNSUPPORTED;
}

// out[M,C] = softmax(q k^T / sqrt(d)) v per partition/head ; lse[M,heads] (optional) = log-sum-exp of the scaled scores
// 1: forward and fused backward of this geometry run on the LDS kernels in the current precision mode bf16, i.e. qkv may be handed
// over as bf16 (qkv_bf16) and dqkv produced as bf16 (dqkv_bf16); 0: fp32 tensors only
LEOD_API int leod_partition_attn_16bit_ok(int B, int H, int W, int C, int heads, int ph, int pw) {
    if (leod_precision() != 1 || heads <= 0 || C % heads || H % ph || W % pw) return 0;
    constexpr int use_lds = 1;
    static const int on = 1;
    const int d = C / heads, P = ph * pw, PT = (P + 15) / 16;
    const int HG = (heads % 2 == 0 && 2 * PT <= 16) ? 2 : 1;
    const bool inst = PT <= 5 || PT == 8 || (HG == 1 && (PT == 10 || PT == 15));
    static const int force_pad1 = 1;
    return on && use_lds && (d == 24 || d == 32) && inst && !(HG == 1 && P < 16 * PT && !force_pad1);
}
// 1: on top of leod_partition_attn_16bit_ok, the attention output O may be written as bf16 (forward, bit 1 of qkv_bf16) and its gradient
// dO read as bf16 (backward, bit 1 of qkv_bf16): the bf16-tile kernels stage both as the bf16 MFMA operands they are anyway
LEOD_API int leod_partition_attn_o16_ok(int B, int H, int W, int C, int heads, int ph, int pw) {
    static const int on = 1;
    return on && leod_partition_attn_16bit_ok(B, H, W, C, heads, ph, pw);
}

LEOD_API int leod_partition_attn_fwd(const float* qkv, float* out, float* lse, int B, int H, int W, int C, int heads,
                                     int ph, int pw, int window, int qkv_bf16, hipStream_t stream) {
    LeodFwdScope fwd_scope;
    if (!qkv || !out || heads <= 0) return LEOD_ERR_ARG;
    AttnGeom g{B, H, W, C, heads, C / heads, ph, pw, window, ((qkv_bf16 & 1) ? 1 : 0) | ((qkv_bf16 & 2) ? 4 : 0)};
    return dispatch_attn(0, qkv, nullptr, out, lse, nullptr, nullptr, g, stream);
}

// dqkv[M,3C] = gradient of the attention core wrt the qkv rows; dsum [M,heads] is scratch
LEOD_API int leod_partition_attn_bwd(const float* qkv, const float* dout, const float* lse, float* dsum, float* dqkv,
                                     int B, int H, int W, int C, int heads, int ph, int pw, int window,
                                     int qkv_bf16, int dqkv_bf16, hipStream_t stream) {
    if (!qkv || !dout || !lse || !dsum || !dqkv || heads <= 0) return LEOD_ERR_ARG;
    AttnGeom g{B, H, W, C, heads, C / heads, ph, pw, window, ((qkv_bf16 & 1) ? 1 : 0) | (dqkv_bf16 ? 2 : 0) | ((qkv_bf16 & 2) ? 8 : 0)};
    int rc = dispatch_attn(1, qkv, dout, nullptr, const_cast<float*>(lse), dsum, dqkv, g, stream);
    if (rc != LEOD_OK) return rc;
    return dispatch_attn(2, qkv, dout, nullptr, const_cast<float*>(lse), dsum, dqkv, g, stream);
}
