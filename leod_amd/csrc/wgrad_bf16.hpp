// Weight-gradient GEMM of precision mode bf16 (round 3):  dW[n][k] += sum_m dY(m,n) * X(m,k)  (+ dbias[n] += sum_m dY(m,n)) for the
// Linear layers of the time-batched step (13 k - 860 k rows, N and K in 48 .. 1536) -- the shapes wgradw_kernel<.., BF = true> served.
//
// What bound wgradw_kernel there (tools/kbench.py wgrad, bf16 mode: 100-210 TFLOP/s and 0.7-4.2 TB/s on EVERY shape, i.e. neither
// roofline; 69 % of its wave cycles in s_waitcnt / barriers, profiles/r03_d_wgradw_pmc.txt):
//   * 16-row chunks: one barrier and one exposed HBM round trip per 9 MFMAs of a wave, 6 KB of loads in flight per row group
//     (24 KB per CU: Little's law at ~2 us of loaded latency gives ~4 TB/s, which is what the 16-bit variants reached);
//   * the 16-k bf16 MFMA (half rate) because a 16-row step offers 16 contraction values;
//   * VALU work behind every load: bf16 -> fp32 -> bf16 round trips of dY rows, fp32 column sums for the bias gradient, the
//     LayerNorm scale / shift (with their LDS reads) on every element, 8-byte loads of 16-bit rows.
// Here:
//   * 64-row chunks (32 for the 240-column tiles, 128 for 48 x 48), double-buffered in LDS with ONE barrier per chunk and the next
//     chunk's global loads (16 bytes per lane, fixed column per thread: no per-slot address arithmetic) in flight under the MFMAs:
//     2-4 workgroups per CU keep 60-75 KB of loads in flight;
//   * v_mfma_f32_16x16x32_bf16: an operand fragment is TWO ds_read_b64_tr_b16 of consecutive [16 row][16 col] blocks (rows 4q..4q+3
//     and 16+4q..+3 of a 32-row step: the same row permutation for dY and X, so the contraction is unchanged);
//   * bf16 dY rows go to LDS as loaded (no conversion); the bias gradient is an MFMA against a fragment of ones;
//   * LayerNorm operands are staged as xhat = (x - mean) * rstd only; scale and shift are applied to the finished tile:
//       dW[n][k] = ln_w[k] * sum_m dY(m,n) xhat(m,k) + ln_b[k] * sum_m dY(m,n)
//   * GELU operands (fp16 pre-activations) are evaluated two at a time on the packed fp32 VALU (v_pk_fma_f32 / v_pk_mul_f32).
// dW accumulates with one fp32 atomic per element and workgroup, as before.
#pragma once

typedef unsigned u4_ __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f2_ gelu_erf2(f2_ x) {              // gelu_erf (common.hpp) on two values: packed fp32 arithmetic
    const f2_ z = __builtin_elementwise_abs(x) * 0.70710678118654752440f;
    const f2_ d = z * 0.3275911f + 1.0f;
    const f2_ t = {fast_rcp(d.x), fast_rcp(d.y)};
    const f2_ poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const f2_ a = (z * z) * -1.4426950408889634f;
    const f2_ e = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
    const f2_ ht = (0.5f * poly) * e;
    const f2_ cdf = {x.x >= 0.f ? 1.0f - ht.x : ht.x, x.y >= 0.f ? 1.0f - ht.y : ht.y};
    return x * cdf;
}
__device__ __forceinline__ unsigned pack_bf16x2(f2_ v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2_)); }

// row groups of a staged operand: the largest power of two <= 256 / (slots per row); always 192 threads = three whole waves,
// because every tile is 3 * 2^j slots wide
constexpr int wgw_row_groups(int slots_per_row) {
    int g = 1;
    while (2 * g * slots_per_row <= 256) g *= 2;
    return g;
}
// LDS element offset of staging slot e relative to slot 0 of the thread (rows rg + RG * e of the [16 row][16 col] block layout)
template <int RG, int T, int BST>
__device__ __forceinline__ constexpr int wgw_slot_off(int e) {
    return RG >= 16 ? e * (RG / 16) * T * BST : (e % (16 / (RG < 16 ? RG : 16))) * RG * 16 + (e / (16 / (RG < 16 ? RG : 16))) * T * BST;
}

// dynamic LDS of a workgroup: the double-buffered staging, or the accumulator exchange of waves that split the row steps (MS of them)
template <int TN, int TK, int RC, int MS = 1>
constexpr int wgw_lds_bytes() {
    return 2 * ((RC / 16) * (TN + TK) * 272) * 2 > (MS - 1) * 12 * 1024 ? 2 * ((RC / 16) * (TN + TK) * 272) * 2 : (MS - 1) * 12 * 1024;   // (MS > 1 only with 3 x 3 tiles per wave)
}

// TN x TK: 16 x 16 tiles of the workgroup's dW tile; NWN x NWK: wave grid over it (each wave 3 x 3 tiles; waves left over split the
// 32-row steps of a chunk); RC rows per chunk; DYF: dY fp32 (0) / bf16 (1); XM: see XRows (0 rows, 1 LayerNorm, 2 gelu(fp16), 3 bf16 rows, 4 fp16 rows)
template <int TN, int TK, int NWN, int NWK, int RC, int DYF, int XM, int OCC>
__global__ __launch_bounds__(256, OCC) void wgrad_wide_bf16_kernel(const void* __restrict__ dyv, long lddy, XRows xl, float* dW, long ldw,
                                                                    float* dbias, f4* __restrict__ part, int M, int N, int K, int dbg) {
    constexpr int WA = TN / NWN, WB = TK / NWK, MS = 4 / (NWN * NWK), KS = RC / 32;
    constexpr bool K16 = KS % MS != 0;              // the waves split 16-row steps instead (16-k MFMA): the 48 x 48 tile
    // a wave holds WA x WB tiles: 3 x 3, or (stages 2-4) 6 x 3 / 3 x 6 -- 192 x 96 / 96 x 192 outputs per workgroup, so that the fp32 operand
    // is fetched by half as many workgroups.  Measured (tools/wgrad_big_ab.sh): -2 % on the 54 k / 13 k-row launches, 0.1-0.2 ms per step;
    // those launches are bound by the latency of ONE chunk of loads in flight per workgroup (~3 us per 32-row chunk at 2-3 workgroups per CU:
    // cutting their L2 traffic by a third changed little, and a second chunk in flight spilled at this register budget: 79 -> 103-134 us).
    constexpr int NSL = WA * WB + WA;                   // 64-lane f4 slots of a wave's partial: its tiles, then its column-sum tiles
    static_assert((WA == 3 || WA == 6) && (WB == 3 || WB == 6) && WA * WB <= 18 && NWN * NWK * MS == 4 && (RC / 16) % MS == 0 &&
                  (MS == 1 || (WA == 3 && WB == 3)), "4 waves of 3 x 3, 6 x 3 or 3 x 6 tiles");
    constexpr int BST = 16 * 16 + 16;
    constexpr int DCW = DYF ? 8 : 4, XCW = XM >= 2 ? 8 : 4;                 // columns per 16-byte load
    constexpr int DSPR = 16 * TN / DCW, XSPR = 16 * TK / XCW;
    constexpr int DRG = wgw_row_groups(DSPR) < RC ? wgw_row_groups(DSPR) : RC, XRG = wgw_row_groups(XSPR) < RC ? wgw_row_groups(XSPR) : RC;
    // XG (gelu operands): the evaluation is VALU-bound, so ALL four waves stage X -- slot s = tid + 256 e of the chunk's RC x XSPR slots
    // (row s / XSPR, column slot s % XSPR; offsets per slot in registers) instead of three waves with a fixed column
    constexpr bool XG = XM == 2;
    static_assert(!XG || (RC * XSPR) % 256 == 0, "whole slots per thread");
    constexpr int RN = RC / DRG, RK = XG ? RC * XSPR / 256 : RC / XRG;
    constexpr int SDY = (RC / 16) * TN * BST, SX = (RC / 16) * TK * BST;    // bf16 elements of one buffer
    static_assert(DSPR * DRG == 192 && (XG || XSPR * XRG == 192), "three waves stage an operand");
    static_assert(2 * (SDY + SX) * 2 <= wgw_lds_bytes<TN, TK, RC, MS>(), "LDS size");
    extern __shared__ __attribute__((aligned(16))) unsigned short wg_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, q = lane >> 4;
    const int wn = wave % NWN, wk = (wave / NWN) % NWK, ws = wave / (NWN * NWK);
    const int n0 = blockIdx.y * TN * 16, k0 = blockIdx.z * TK * 16;
    // ---- staging: dY by waves 0-2, X by waves 1-3; a thread keeps ONE column slot and takes rows rg + RG * e ------------------------
    const bool dact = wave < 3, xact = XG || wave >= 1;
    const int xt = tid - 64;
    const int dcs = tid % DSPR, drg = tid / DSPR, xcs = (xact ? xt : 0) % XSPR, xrg = (xact ? xt : 0) / XSPR;
    const int dcol = n0 + dcs * DCW, xcol = k0 + xcs * XCW;
    const bool dok = dcol < N, xok = xcol < K;
    const char* dbase; long dstride;            // bytes
    {
        constexpr int ES = DYF ? 2 : 4;
        dbase = reinterpret_cast<const char*>(dyv) + (long)(dok ? dcol : 0) * ES;
        dstride = lddy * ES;
    }
    const char* xbase; long xstride;
    {
        constexpr int ES = XM >= 2 ? 2 : 4;
        const int c = xok ? xcol : 0;
        if (XM == 0 && xl.x2 && c >= xl.K1) { xbase = reinterpret_cast<const char*>(xl.x2) + (long)(c - xl.K1) * ES; xstride = xl.ld2 * ES; }
        else { xbase = reinterpret_cast<const char*>(xl.x) + (long)c * ES; xstride = xl.ld * ES; }
    }
    const int dl0 = ((drg >> 4) * TN + ((dcs * DCW) >> 4)) * BST + (drg & 15) * 16 + ((dcs * DCW) & 15);
    const int xl0 = ((xrg >> 4) * TK + ((xcs * XCW) >> 4)) * BST + (xrg & 15) * 16 + ((xcs * XCW) & 15);
    int gxr[XG ? RK : 1], gxo[XG ? RK : 1], gxc[XG ? RK : 1];      // XG: row in the chunk, LDS element offset, byte offset of the column
    bool gxk[XG ? RK : 1];
    if constexpr (XG) {
#pragma unroll
        for (int e = 0; e < RK; ++e) {
            const int s = tid + 256 * e, r = s / XSPR, c = (s - r * XSPR) * XCW;
            gxr[e] = r; gxk[e] = k0 + c < K; gxc[e] = (gxk[e] ? k0 + c : 0) * 2;
            gxo[e] = ((r >> 4) * TK + (c >> 4)) * BST + (r & 15) * 16 + (c & 15);
        }
    }
    u4_ rd[RN], rx[RK];
    f2_ rst[XM == 1 ? RK : 1];
    auto fetch = [&](long m0, auto tailc) {
        constexpr bool TAIL = decltype(tailc)::value;
        if (dact) {
            const char* p = dbase + (m0 + drg) * dstride;
#pragma unroll
            for (int e = 0; e < RN; ++e) {
                if constexpr (TAIL) rd[e] = *reinterpret_cast<const u4_*>(dbase + min(m0 + drg + DRG * e, (long)M - 1) * dstride);
                else rd[e] = *reinterpret_cast<const u4_*>(p + (long)(DRG * e) * dstride);
            }
        }
        if constexpr (XG) {
            const char* xb = reinterpret_cast<const char*>(xl.x);
            const long xs = xl.ld * 2;
#pragma unroll
            for (int e = 0; e < RK; ++e) {
                const long r = TAIL ? min(m0 + gxr[e], (long)M - 1) : m0 + gxr[e];
                rx[e] = *reinterpret_cast<const u4_*>(xb + r * xs + gxc[e]);
            }
        } else if (xact) {
            const char* p = xbase + (m0 + xrg) * xstride;
#pragma unroll
            for (int e = 0; e < RK; ++e) {
                if constexpr (TAIL) {
                    const long r = min(m0 + xrg + XRG * e, (long)M - 1);
                    rx[e] = *reinterpret_cast<const u4_*>(xbase + r * xstride);
                    if constexpr (XM == 1) rst[e] = *reinterpret_cast<const f2_*>(xl.stats + 2 * r);
                } else {
                    rx[e] = *reinterpret_cast<const u4_*>(p + (long)(XRG * e) * xstride);
                    if constexpr (XM == 1) rst[e] = *reinterpret_cast<const f2_*>(xl.stats + 2 * (m0 + xrg + XRG * e));
                }
            }
        }
    };
    auto stash = [&](long m0, int buf) {
        const int lim = (int)min((long)RC, (long)M - m0);             // valid rows of this chunk
        if (dact) {
            unsigned short* d = wg_smem + buf * (SDY + SX) + dl0;
#pragma unroll
            for (int e = 0; e < RN; ++e) {
                const bool ok = dok && drg + DRG * e < lim;
                if constexpr (DYF) {
                    u4_ v = rd[e];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = ok ? v[j] : 0u;
                    *reinterpret_cast<u4_*>(d + wgw_slot_off<DRG, TN, BST>(e)) = v;
                } else {
                    const f4 v = __builtin_bit_cast(f4, rd[e]);
                    u2_ o = {pack_bf16x2(f2_{v.x, v.y}), pack_bf16x2(f2_{v.z, v.w})};
                    o.x = ok ? o.x : 0u; o.y = ok ? o.y : 0u;
                    *reinterpret_cast<u2_*>(d + wgw_slot_off<DRG, TN, BST>(e)) = o;
                }
            }
        }
        if (xact) {
            unsigned short* d = wg_smem + buf * (SDY + SX) + SDY + (XG ? 0 : xl0);
#pragma unroll
            for (int e = 0; e < RK; ++e) {
                const bool ok = XG ? (gxk[e] && gxr[e] < lim) : (xok && xrg + XRG * e < lim);
                if constexpr (XM == 2) {
                    auto cv = [&](unsigned w) -> unsigned {          // two fp16 pre-activations -> two bf16 gelu values
                        const h2_ h = __builtin_bit_cast(h2_, w);
                        const f2_ u = {(float)h.x, (float)h.y};
                        return ok ? pack_bf16x2((dbg & 4) ? u : gelu_erf2(u)) : 0u;
                    };
                    const u4_ o = {cv(rx[e].x), cv(rx[e].y), cv(rx[e].z), cv(rx[e].w)};
                    *reinterpret_cast<u4_*>(d + gxo[e]) = o;
                } else if constexpr (XM == 3) {                     // bf16 rows: staged as loaded
                    u4_ v = rx[e];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = ok ? v[j] : 0u;
                    *reinterpret_cast<u4_*>(d + wgw_slot_off<XRG, TK, BST>(e)) = v;
                } else if constexpr (XM == 4) {                     // fp16 rows (precision mode 16f: the attention output): re-rounded to bf16
                    auto cv = [&](unsigned w) -> unsigned {          // two fp16 values -> two bf16 values
                        const h2_ h = __builtin_bit_cast(h2_, w);
                        return ok ? pack_bf16x2(f2_{(float)h.x, (float)h.y}) : 0u;
                    };
                    const u4_ o = {cv(rx[e].x), cv(rx[e].y), cv(rx[e].z), cv(rx[e].w)};
                    *reinterpret_cast<u4_*>(d + wgw_slot_off<XRG, TK, BST>(e)) = o;
                } else {
                    f4 v = __builtin_bit_cast(f4, rx[e]);
                    if constexpr (XM == 1) v = (v - rst[e].x) * rst[e].y;
                    u2_ o = {pack_bf16x2(f2_{v.x, v.y}), pack_bf16x2(f2_{v.z, v.w})};
                    o.x = ok ? o.x : 0u; o.y = ok ? o.y : 0u;
                    *reinterpret_cast<u2_*>(d + wgw_slot_off<XRG, TK, BST>(e)) = o;
                }
            }
        }
    };
    // ---- accumulators: 3 x 3 tiles of dW and the column sums of dY (bias gradient; LayerNorm shift term) ----------------------------
    // bias tile a of a wave's three: every wave needs its own in LayerNorm mode; otherwise the waves of a row of the grid share them out
    const bool bias_any = XM == 1 || (dbias != nullptr && blockIdx.z == 0);
    f4 acc[WA][WB], bacc[WA];
#pragma unroll
    for (int a = 0; a < WA; ++a) {
        bacc[a] = zero4();
#pragma unroll
        for (int b = 0; b < WB; ++b) acc[a][b] = zero4();
    }
    const s8v ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    typedef __attribute__((address_space(3))) s4 lds_s4;
    const int loff = (4 * q + (i >> 2)) * 16 + 4 * (i & 3);
    auto mfma_chunk = [&](int buf) {
        const unsigned short* pdy = wg_smem + buf * (SDY + SX) + (wn * WA) * BST + loff;
        const unsigned short* px = wg_smem + buf * (SDY + SX) + SDY + (wk * WB) * BST + loff;
        if constexpr (K16) {
            const s4 ones4 = {0x3F80, 0x3F80, 0x3F80, 0x3F80};
#pragma unroll
            for (int st = ws; st < RC / 16; st += MS) {
                s4 pa[WA], pb[WB];
#pragma unroll
                for (int a = 0; a < WA; ++a) pa[a] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(pdy + (st * TN + a) * BST));
#pragma unroll
                for (int b = 0; b < WB; ++b) pb[b] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(px + (st * TK + b) * BST));
#pragma unroll
                for (int a = 0; a < WA; ++a)
#pragma unroll
                    for (int b = 0; b < WB; ++b) acc[a][b] = mfma16_bf16(pa[a], pb[b], acc[a][b]);
                if (bias_any) {
#pragma unroll
                    for (int a = 0; a < WA; ++a)
                        if (XM == 1 || a % NWK == wk) bacc[a] = mfma16_bf16(pa[a], ones4, bacc[a]);
                }
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < KS / MS; ++ks) {
            const int st = 2 * (ws + MS * ks);
            s8v pa[WA], pb[WB];
#pragma unroll
            for (int a = 0; a < WA; ++a) {
                const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(pdy + (st * TN + a) * BST));
                const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(pdy + ((st + 1) * TN + a) * BST));
                pa[a] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int b = 0; b < WB; ++b) {
                const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(px + (st * TK + b) * BST));
                const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(px + ((st + 1) * TK + b) * BST));
                pb[b] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int a = 0; a < WA; ++a)
#pragma unroll
                for (int b = 0; b < WB; ++b) acc[a][b] = mfma32_bf16(pa[a], pb[b], acc[a][b]);
            if (bias_any) {
#pragma unroll
                for (int a = 0; a < WA; ++a)
                    if (XM == 1 || a % NWK == wk) bacc[a] = mfma32_bf16(pa[a], ones, bacc[a]);
            }
        }
        }
    };
    // ---- chunk stream: chunk c of this workgroup = rows (blockIdx.x + c * gridDim.x) * RC .. (the resident workgroups stream one
    // contiguous window of dY / X) ------------------------------------------------------------------------------------------------------
    const std::integral_constant<bool, false> FULL{};
    const std::integral_constant<bool, true> TAILC{};
    const long mstride = (long)gridDim.x * RC;
    long m0 = (long)blockIdx.x * RC;
    if (m0 < M) {
        if (m0 + RC <= M) fetch(m0, FULL); else fetch(m0, TAILC);
        stash(m0, 0);
    }
    __syncthreads();
    int buf = 0;
    for (; m0 < M; m0 += mstride) {
        const long mn = m0 + mstride;
        const bool more = mn < M;
        if (more) { if (mn + RC <= M) fetch(mn, FULL); else fetch(mn, TAILC); }
        if (!(dbg & 2)) mfma_chunk(buf);
        if (more) stash(mn, buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // ---- waves that split the row steps add their tiles through LDS (the staging bytes are dead after the last barrier) --------------
    if constexpr (MS > 1) {
        f4* sred = reinterpret_cast<f4*>(wg_smem);
        if (ws > 0) {
#pragma unroll
            for (int a = 0; a < WA; ++a) {
#pragma unroll
                for (int b = 0; b < WB; ++b) sred[((ws - 1) * NSL + a * WB + b) * 64 + lane] = acc[a][b];
                sred[((ws - 1) * NSL + WA * WB + a) * 64 + lane] = bacc[a];
            }
        }
        __syncthreads();
        if (ws > 0) return;
#pragma unroll
        for (int o = 0; o < MS - 1; ++o)
#pragma unroll
            for (int a = 0; a < WA; ++a) {
#pragma unroll
                for (int b = 0; b < WB; ++b) acc[a][b] += sred[(o * NSL + a * WB + b) * 64 + lane];
                bacc[a] += sred[(o * NSL + WA * WB + a) * 64 + lane];
            }
    }
    if (dbg & 1) return;
    // ---- epilogue: acc[a][b][r] = tile element (n = 4q + r, k = i); bacc[a][r] = column sum of dY column n (the same in every lane i) --
    // LayerNorm scale / shift of the finished tile (see the header)
    if constexpr (XM == 1) {
#pragma unroll
        for (int b = 0; b < WB; ++b) {
            const int k = k0 + 16 * (wk * WB + b) + i;
            const float g = k < K ? xl.ln_w[k] : 0.f, sh = k < K ? xl.ln_b[k] : 0.f;
#pragma unroll
            for (int a = 0; a < WA; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[a][b][r] = fmaf(g, acc[a][b][r], sh * bacc[a][r]);
        }
    }
    const bool bias_out = dbias != nullptr && blockIdx.z == 0;
    if (part != nullptr) {
        // partial tile of this workgroup as the waves hold it: 12 slots (9 tiles, 3 column-sum tiles) of 64 lanes x 16 bytes per wave --
        // 1 KB stores; wgrad_wide_reduce_kernel adds the partials of a tile and scatters the sums to dW / dbias.  (Atomics straight
        // from here: 1024 workgroups x 9216 elements on the SAME addresses took 130 us of a 200 us launch, see the header of the reduce.)
        constexpr int NW = NWN * NWK;
        const long tile = (long)blockIdx.y * gridDim.z + blockIdx.z, ntiles = (long)gridDim.y * gridDim.z;
        f4* dst = part + (((long)blockIdx.x * ntiles + tile) * NW + (wk * NWN + wn)) * (NSL * 64) + lane;
#pragma unroll
        for (int a = 0; a < WA; ++a) {
#pragma unroll
            for (int b = 0; b < WB; ++b) dst[(a * WB + b) * 64] = acc[a][b];
            if (bias_out && a % NWK == wk) dst[(WA * WB + a) * 64] = bacc[a];
        }
        return;
    }
#pragma unroll
    for (int b = 0; b < WB; ++b) {
        const int k = k0 + 16 * (wk * WB + b) + i;
#pragma unroll
        for (int a = 0; a < WA; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + 16 * (wn * WA + a) + 4 * q + r;
                if (n < N && k < K) atomicAdd(dW + (long)n * ldw + k, acc[a][b][r]);
            }
    }
    if (bias_out && i == 0) {
#pragma unroll
        for (int a = 0; a < WA; ++a)
            if (a % NWK == wk) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + 16 * (wn * WA + a) + 4 * q + r;
                    if (n < N) atomicAdd(dbias + n, bacc[a][r]);
                }
            }
    }
}

// Sum of the per-workgroup partial tiles of wgrad_wide_bf16_kernel (gx partials of ntiles tiles, each NW waves x 12 slots x 64 lanes x f4)
// into dW / dbias.  One thread per f4 of a partial; gridDim.y groups of `per` partials each (one group: plain read-modify-write of
// dW, otherwise one atomic per element and group -- 16-42 per element instead of 1024).
template <int TN, int TK, int NWN, int NWK>
__global__ __launch_bounds__(256) void wgrad_wide_reduce_kernel(const f4* __restrict__ part, int gx, int ny, int nz, int per, float* dW, long ldw,
                                                                float* dbias, int N, int K) {
    constexpr int NW = NWN * NWK, WA = TN / NWN, WB = TK / NWK, NT9 = WA * WB, NSL = NT9 + WA;
    const long O = (long)ny * nz * NW * NSL * 64;
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    if (o >= O) return;
    const int lane = (int)(o & 63), t = (int)((o >> 6) % NSL), w = (int)((o / (64 * NSL)) % NW), tile = (int)(o / (64 * NSL * NW));
    const int ty = tile / nz, tz = tile - ty * nz, wn = w % NWN, wk = w / NWN, i = lane & 15, q = lane >> 4;
    if (t >= NT9 && !(dbias != nullptr && tz == 0 && (t - NT9) % NWK == wk && i == 0)) return;
    const int g0 = blockIdx.y * per, g1 = min(gx, g0 + per);
    f4 s0 = zero4(), s1 = zero4(), s2 = zero4(), s3 = zero4();
    const f4* p = part + (long)g0 * O + o;
    int g = g0;
    f4 s4 = zero4(), s5 = zero4(), s6 = zero4(), s7 = zero4();
    for (; g + 8 <= g1; g += 8, p += 8 * O) {          // eight independent 16-byte loads in flight per thread
        s0 += p[0]; s1 += p[O]; s2 += p[2 * O]; s3 += p[3 * O]; s4 += p[4 * O]; s5 += p[5 * O]; s6 += p[6 * O]; s7 += p[7 * O];
    }
    for (; g + 4 <= g1; g += 4, p += 4 * O) { s0 += p[0]; s1 += p[O]; s2 += p[2 * O]; s3 += p[3 * O]; }
    for (; g < g1; ++g, p += O) s0 += p[0];
    const f4 v = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
    const bool single = gridDim.y == 1;
    if (t < NT9) {
        const int a = t / WB, b = t - a * WB;
        const int k = tz * TK * 16 + 16 * (wk * WB + b) + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = ty * TN * 16 + 16 * (wn * WA + a) + 4 * q + r;
            if (n < N && k < K) {
                float* d = dW + (long)n * ldw + k;
                if (single) *d += v[r]; else atomicAdd(d, v[r]);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = ty * TN * 16 + 16 * (wn * WA + (t - NT9)) + 4 * q + r;
            if (n < N) { if (single) dbias[n] += v[r]; else atomicAdd(dbias + n, v[r]); }
        }
    }
}

// Scratch of the partial tiles: one buffer per stream (launches of a stream are ordered, so they share it).  The caller registers it
// (leod_set_workspace: the host side allocates it with its own allocator, which also works under stream capture); without one the
// library allocates on first use, and if that is refused (stream capture in progress) the launch takes the atomic epilogue.
#include <mutex>
#include <unordered_map>
struct WgwScratch { void* p = nullptr; size_t bytes = 0; };
static inline std::unordered_map<hipStream_t, WgwScratch>& wgw_scratch_map() { static std::unordered_map<hipStream_t, WgwScratch> m; return m; }
static inline std::mutex& wgw_scratch_mutex() { static std::mutex m; return m; }
constexpr size_t kWgwScratchBytes = (size_t)768 << 20;      // partial tiles (1024 workgroups x 4 waves x 12 KB) + the bf16 operand copies of wgrad_dma.hpp
static inline void wgrad_wide_register_scratch(hipStream_t s, void* p, size_t bytes) {
    std::lock_guard<std::mutex> lock(wgw_scratch_mutex());
    WgwScratch& e = wgw_scratch_map()[s];
    e.p = p; e.bytes = p ? bytes : 0;
}
static inline f4* wgrad_wide_scratch(hipStream_t s, size_t bytes) {
    std::lock_guard<std::mutex> lock(wgw_scratch_mutex());
    WgwScratch& e = wgw_scratch_map()[s];
    if (e.bytes >= bytes) return reinterpret_cast<f4*>(e.p);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
    void* nb = nullptr;
    const size_t want = bytes < kWgwScratchBytes ? kWgwScratchBytes : bytes;
    if (hipMalloc(&nb, want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    e.p = nb; e.bytes = want;        // (a smaller buffer it replaces may still be read by launches in flight: it is not freed)
    return reinterpret_cast<f4*>(nb);
}

template <int TN, int TK, int NWN, int NWK, int RC, int DYF, int XM, int OCC>
static inline int launch_wgrad_wide_cfg(const void* dy, long lddy, const XRows& xl, float* dW, long ldw, float* dbias,
                                        int M, int N, int K, hipStream_t s) {
    constexpr int dbg = 0;          // ablation bits of the kernel (skip epilogue / MFMAs / loads): compile-time, for experiments
    static const int tune_wgs = OCC * 256;
    constexpr int LDS = wgw_lds_bytes<TN, TK, RC, 4 / (NWN * NWK)>();
    const int tiles = cdiv(N, TN * 16) * cdiv(K, TK * 16), chunks = cdiv(M, RC);
    // all workgroups resident (OCC per CU); >= 2 chunks each; a multiple of 8 per output tile keeps the workgroups that stream the same
    // rows for different tiles on one XCD (see launch_wgradw_cfg)
    // launches of <= 60 k rows (stages 3-4) run beside the main lane's kernels and are latency-bound: 384 workgroups instead of OCC * 256
    // leave wave slots to the other lane and halve the partial tiles (15.84 -> 15.76-15.79 ms per step, profiles/r04_a_graph_ab.txt)
    static const int tune_small = 384;
    static const int small_rows = 60000;
    const int wgs = (tune_small > 0 && M <= small_rows) ? tune_small : tune_wgs;
    int gx = max(1, min(chunks / 2, wgs / tiles));
    if (gx >= 16) gx &= ~7;
    dim3 grid(gx, cdiv(N, TN * 16), cdiv(K, TK * 16));
    auto kern = wgrad_wide_bf16_kernel<TN, TK, NWN, NWK, RC, DYF, XM, OCC>;
    static bool attr_set = false;                             // dynamic LDS opt-in, once per instantiation
    if (!attr_set) { hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr_set = true; }
    static const int use_part = 1;
    constexpr int NW = NWN * NWK, NSL = (TN / NWN) * (TK / NWK) + TN / NWN;
    const long O = (long)tiles * NW * NSL * 64;               // f4 per partial
    f4* part = (use_part && gx > 1) ? wgrad_wide_scratch(s, (size_t)gx * O * sizeof(f4)) : nullptr;
    hipLaunchKernelGGL(kern, grid, dim3(256), LDS, s, dy, lddy, xl, dW, ldw, dbias, part, M, N, K, dbg);
    if (part != nullptr && !(dbg & 1)) {
        // ~2048 waves of reduce threads: groups of `per` partials
        const int ob = (int)cdiv(O, 256);
        // (capping the groups -- fewer atomics per element, more partials per thread -- did not shorten this kernel: its 100-300 us at the tail of
        // the weight-gradient lane are contention with the other lane's kernels, 17 us when it runs alone; profiles/r04_a_graph_ab.txt)
        int groups = max(1, min(gx, 512 / ob));
        const int per = cdiv(gx, groups);
        groups = cdiv(gx, per);
        hipLaunchKernelGGL((wgrad_wide_reduce_kernel<TN, TK, NWN, NWK>), dim3(ob, groups), dim3(256), 0, s, part, gx, (int)grid.y, (int)grid.z, per,
                           dW, ldw, dbias, N, K);
    }
    return leod_launch_status();
}

// The (tile, dY format, X mode) combinations of the RVT step in precision mode bf16 -- anything else stays on wgradw_kernel:
//   48 x 48   proj (stage 1): fp32 dY, fp32 or bf16 rows (the attention output O)
//   192 x 48  qkv / fc1 (stage 1): bf16 dY, LayerNorm(fp32 rows)
//   48 x 192  fc2 (stage 1): fp32 dY, gelu(fp16 rows)
//   96 x 96   stages 2-4 and the ConvLSTM 1x1: the four pairs above and bf16 dY with fp32 [x | h] rows
// chunk rows: ~18-25 KB of loads per chunk (fp32 rows carry twice the bytes of 16-bit rows, and twice the staging registers)
static inline int wgrad_wide_combo(const XRows& xl, int N, int K, int dyfmt) {
    const int xm = xl.x_mode();
    if (xm == 4) {                                                   // fp32 dY, fp16 rows (mode 16f): the tilings of the bf16-row cases
        if (dyfmt) return 0;
        if (N <= 48 && K <= 48) return 14;
        if (K <= 48 || N <= 48) return 0;
        static const int big4 = 1;
        return (big4 && K % 192 == 0 && N >= 96) ? 16 : 15;
    }
    const int c = (dyfmt ? 1 : 0) * 4 + xm;        // 0: f32/rows 1: f32/LN 2: f32/gelu16 3: f32/bf16 rows 4: bf16/rows 5: bf16/LN
    if (c > 5) return 0;
    if (N <= 48 && K <= 48) return c == 0 ? 1 : c == 3 ? 8 : 0;
    if (K <= 48) return c == 5 ? 2 : 0;
    if (N <= 48) return c == 2 ? 3 : 0;
    // 192-wide tiles along the 16-bit operand's side: 192 x 96 outputs for bf16 dY with fp32 X (qkv, fc1, ConvLSTM), 96 x 192 for fp32 dY
    // with 16-bit X (fc2 on the fp16 hidden, proj on bf16 O)
    static const int big = 1;
    if (big && N % 192 == 0 && K >= 96 && (c == 5 || c == 4)) return c == 5 ? 10 : 11;
    if (big && K % 192 == 0 && N >= 96 && (c == 2 || c == 3)) return c == 2 ? 12 : 13;
    return c == 0 ? 4 : c == 5 ? 5 : c == 2 ? 6 : c == 4 ? 7 : c == 3 ? 9 : 0;
}
static inline bool use_wgrad_wide(const XRows& xl, long lddy, int M, int N, int K, int dyfmt) {
    if (leod_precision() != 1 || M < 8192) return false;
    const int xm = xl.x_mode();
    const int dcw = dyfmt ? 8 : 4, xcw = xm >= 2 ? 8 : 4;
    if ((N % dcw) || (lddy % dcw) || (K % xcw) || (xl.ld % xcw)) return false;
    if (xl.x2 && (xm != 0 || (xl.K1 % 4) || (xl.ld2 % 4))) return false;
    return wgrad_wide_combo(xl, N, K, dyfmt) != 0;
}
static inline int launch_wgrad_wide(const void* dy, long lddy, const XRows& xl, float* dW, long ldw, float* dbias,
                                    int M, int N, int K, hipStream_t s, int dyfmt) {
#define LEOD_WGW(TN, TK, NWN, NWK, RC, DYF, XM, OCC) \
    return launch_wgrad_wide_cfg<TN, TK, NWN, NWK, RC, DYF, XM, OCC>(dy, lddy, xl, dW, ldw, dbias, M, N, K, s)
    switch (wgrad_wide_combo(xl, N, K, dyfmt)) {
        case 1: LEOD_WGW(3, 3, 1, 1, 64, 0, 0, 2);
        case 2: LEOD_WGW(12, 3, 4, 1, 32, 1, 1, 4);
        case 3: LEOD_WGW(3, 12, 1, 4, 32, 0, 2, 3);
        case 4: LEOD_WGW(6, 6, 2, 2, 32, 0, 0, 3);
        case 5: LEOD_WGW(6, 6, 2, 2, 32, 1, 1, 3);
        case 6: LEOD_WGW(6, 6, 2, 2, 64, 0, 2, 3);
        case 7: LEOD_WGW(6, 6, 2, 2, 32, 1, 0, 3);
        case 8: LEOD_WGW(3, 3, 1, 1, 64, 0, 3, 2);
        case 9: LEOD_WGW(6, 6, 2, 2, 32, 0, 3, 3);
        case 10: LEOD_WGW(12, 6, 2, 2, 32, 1, 1, 2);
        case 11: LEOD_WGW(12, 6, 2, 2, 32, 1, 0, 2);
        case 12: LEOD_WGW(6, 12, 2, 2, 64, 0, 2, 2);
        case 13: LEOD_WGW(6, 12, 2, 2, 32, 0, 3, 2);
        case 14: LEOD_WGW(3, 3, 1, 1, 64, 0, 4, 2);
        case 15: LEOD_WGW(6, 6, 2, 2, 32, 0, 4, 3);
        case 16: LEOD_WGW(6, 12, 2, 2, 32, 0, 4, 2);
    }
#undef LEOD_WGW
    return LEOD_ERR_UNSUPPORTED;
}
