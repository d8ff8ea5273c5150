// Weight-gradient GEMM for the SHORT row ranges (stages 3-4 of the backbone and their ConvLSTMs: 13 k - 54 k rows, N and K multiples of
// 96 in 192 .. 1536; for the power-of-two widths of RVT-B, 128 x 128 or 64 x 64 tiles and up to 400 k rows):  dW[n][k] += sum_m dY(m,n) * X(m,k), dbias[n] += sum_m dY(m,n)   (round 6).
//
// What bound wgrad_wide_bf16_kernel there (profiles/r06_a_bench_line_default.json, roofline.by_rows: 87-90 us per launch at 13 k AND at 54 k
// rows = 160 TFLOP/s, 0.5-1.0 TB/s): a workgroup has ONE 32-row chunk of loads in flight (register staging: a second register set spills),
// so a launch is `chunks per workgroup` dependent memory round trips of ~2 us with 18 MFMAs each in between; and the ablation of
// profiles/r06_c_ablation_headroom.txt shows that these launches are NOT hidden on the side lane (removing them takes 0.8 ms off the step).
// Here the operands are plain bf16 rows -- the 16-bit gradient rows as they are, anything else (fp32 gradient rows, LayerNorm inputs,
// fp16 pre-activations through GELU, fp16 attention outputs, [x | h] of the ConvLSTM) rounded ONCE by wgrad_prep16_kernel, which at
// these sizes costs 3-35 us -- and stream HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no conversion VALU,
// no ds_write pass) into a ring of NS slots of RC rows: NS - 1 chunks of loads in flight per workgroup, counted vmcnt waits, ONE raw
// s_barrier per chunk.  The LDS image of a slot is the [16 row][16 column] block layout of wgrad_bf16.hpp without padding: a DMA
// instruction (64 lanes x 16 bytes) fills two consecutive blocks, the MFMA operand fragments are the same pairs of ds_read_b64_tr_b16.
// Accumulators, LayerNorm scale / shift on the finished tile, per-workgroup partial tiles and wgrad_wide_reduce_kernel<6, 6, 2, 2> are
// those of the 96 x 96 configuration of wgrad_bf16.hpp (same lane -> element map), so the results are bit-identical to it up to the
// summation order over row chunks.
#pragma once

struct Prep16 {                 // one operand of wgrad_prep16_kernel
    const void* src; long ld;   // source rows
    const float* src2; long ld2; int C1;       // mode 4: columns >= C1 come from src2 (fp32)
    const float* stats;         // mode 1: (mean, rstd) per row
    unsigned short* dst;        // bf16 [M][C]
    int C, mode;                // mode 0 fp32, 1 LayerNorm xhat of fp32 rows, 2 gelu(fp16), 3 fp16, 4 fp32 [x | x2], -1: nothing to do
};

// 8 columns per thread (one 16-byte store); grid (ceil(M * maxC8 / 256), 2, problems): y picks the operand, z the problem of a grouped launch
struct Prep16Group { Prep16 a[4]; Prep16 b[4]; };
__global__ __launch_bounds__(256) void wgrad_prep16_kernel(Prep16Group grp, int M) {
    const Prep16& p = blockIdx.y ? grp.b[blockIdx.z] : grp.a[blockIdx.z];
    if (p.mode < 0) return;
    const int C8 = p.C >> 3;
    const long it = (long)blockIdx.x * 256 + threadIdx.x;
    if (it >= (long)M * C8) return;
    const long m = it / C8;
    const int c = (int)(it - m * C8) << 3;
    f4 lo, hi;
    if (p.mode == 2 || p.mode == 3) {
        const u4_ raw = *reinterpret_cast<const u4_*>(reinterpret_cast<const unsigned short*>(p.src) + m * p.ld + c);
        auto up = [](unsigned w) -> f2_ { const h2_ h = __builtin_bit_cast(h2_, w); return f2_{(float)h.x, (float)h.y}; };
        f2_ v0 = up(raw.x), v1 = up(raw.y), v2 = up(raw.z), v3 = up(raw.w);
        if (p.mode == 2) { v0 = gelu_erf2(v0); v1 = gelu_erf2(v1); v2 = gelu_erf2(v2); v3 = gelu_erf2(v3); }
        lo = f4{v0.x, v0.y, v1.x, v1.y}; hi = f4{v2.x, v2.y, v3.x, v3.y};
    } else {
        const float* s = (p.mode == 4 && c >= p.C1) ? p.src2 + m * p.ld2 + (c - p.C1) : reinterpret_cast<const float*>(p.src) + m * p.ld + c;
        lo = ld4(s); hi = ld4(s + 4);
        if (p.mode == 1) {
            const f2_ st = *reinterpret_cast<const f2_*>(p.stats + 2 * m);
            lo = (lo - st.x) * st.y; hi = (hi - st.x) * st.y;
        }
    }
    const u4_ o = {pack_bf16x2(f2_{lo.x, lo.y}), pack_bf16x2(f2_{lo.z, lo.w}), pack_bf16x2(f2_{hi.x, hi.y}), pack_bf16x2(f2_{hi.z, hi.w})};
    *reinterpret_cast<u4_*>(p.dst + m * p.C + c) = o;
}

// one LDS-DMA instruction: 64 lanes x 16 bytes from per-lane global addresses to LDS bytes [dst, dst + 1024) in lane order.  M0 (the
// destination base) is written in the same statement that reads it (cdna_hip_programming.md 5.7); hipcc does not count this load: the
// loop below waits with its own vmcnt.
__device__ __forceinline__ void wgd_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wgd_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// 96 x 96 outputs per workgroup: 4 waves as 2 x 2, 3 x 3 MFMA tiles (16x16x32 bf16) each.  RC rows per chunk, NS ring slots.
// LN: X holds xhat; the tile is scaled / shifted by ln_w / ln_b in the epilogue (needs the column sums of dY in every wave).
template <int T, int RC, int NS> constexpr int wgd_lds_bytes() { return NS * 2 * (RC / 16) * T * 256 * 2; }
template <int T, int RC, int NS> constexpr int wgd_occupancy() { return wgd_lds_bytes<T, RC, NS>() <= 52 * 1024 ? 3 : wgd_lds_bytes<T, RC, NS>() <= 80 * 1024 ? 2 : 1; }

// One problem of a launch; a launch carries up to 4 of one row count (the four Linear weight gradients of an attention block: qkv, proj,
// fc1, fc2 -- issued together at the end of the block's backward) in the 1-D grid, problem k owning workgroups [blk0, blk0 + gx * ny * nz).
struct WgdProb {
    const unsigned short* A; long lda; const unsigned short* B; long ldb; const float* ln_w; const float* ln_b; float* dbias_flag; f4* part;
    int cpw, gx, ny, nz, N, K, blk0;
};
struct WgdGroup { WgdProb p[4]; int n; };
// LN: some problem of the launch has X = xhat (ln_w != NULL for that problem): its tiles are scaled / shifted in the epilogue
template <int T, int RC, int NS, bool LN>
__global__ __launch_bounds__(256, ((wgd_occupancy<T, RC, NS>()) >= 2 ? 2 : 1)) void wgrad_dma_kernel(WgdGroup grp, int chunks, int dbg) {
    int pk = 0;
    for (int k = 1; k < grp.n; ++k) if ((int)blockIdx.x >= grp.p[k].blk0) pk = k;
    const WgdProb& pr = grp.p[pk];
    const unsigned short* __restrict__ A = pr.A; const unsigned short* __restrict__ B = pr.B; const long lda = pr.lda, ldb = pr.ldb;
    const float* __restrict__ ln_w = pr.ln_w; const float* __restrict__ ln_b = pr.ln_b; float* dbias_flag = pr.dbias_flag; f4* __restrict__ part = pr.part;
    const int cpw = pr.cpw, gx = pr.gx, ny = pr.ny, nz = pr.nz, K = pr.K;
    const int bid = (int)blockIdx.x - pr.blk0;
    const bool ln = LN && ln_w != nullptr;
    constexpr int WA = T / 2, WB = T / 2, NWN = 2, NWK = 2, RB = RC / 16, KS = RC / 32;
    constexpr int BLK = 256;                                   // bf16 elements of a [16][16] block (512 bytes, no padding)
    constexpr int SA = RB * T * BLK, SLOT = 2 * SA;            // elements: dY part, then X part
    constexpr int NI = SLOT * 2 / 1024, IPW = NI / 4;          // DMA instructions per chunk / per wave
    static_assert(NI % 4 == 0 && RC % 32 == 0, "whole instructions per wave");
    constexpr int NSL = WA * WB + WA;
    extern __shared__ __attribute__((aligned(1024))) unsigned short wd_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, q = lane >> 4;
    const int wn = wave % NWN, wk = wave / NWN;
    // 1-D grid, XCD-aware (workgroup id -> XCD id % 8 in dispatch order): the ny * nz tiles that stream the SAME row range sit on one XCD
    // next to each other in time, so that one of them misses in that XCD's L2 and the others hit
    const int ntl = ny * nz;
    int tl, bx;
    if ((gx & 7) == 0) { const int xcd = bid & 7, sj = bid >> 3; tl = sj % ntl; bx = (sj / ntl) * 8 + xcd; }
    else { bx = bid % gx; tl = bid / gx; }
    const int by = tl / nz, bz = tl - by * nz;
    const int n0 = by * T * 16, k0 = bz * T * 16;
    const int c0 = bx * cpw, c1 = min(chunks, c0 + cpw);
    const int nch = c1 - c0;
    // ---- DMA sources: instruction t = wave + 4 j fills blocks 2t, 2t + 1 of a slot; lane -> (block, row, half) ---------------------------
    static_assert(IPW % 2 == 0, "the first half of a wave's instructions fills the dY part, the second half the X part");
    const char* src[IPW];
    const long stepA = (long)RC * lda * 2, stepB = (long)RC * ldb * 2;
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
        const int t = wave + 4 * j;
        const int blk = 2 * t + (lane >> 5), r = (lane & 31) >> 1, half = lane & 1;
        const bool isb = j >= IPW / 2;                         // (blk >= RB * T for every wave: RB * T / 2 instructions per part, 4 waves)
        const int b = isb ? blk - RB * T : blk;
        const int rb = b / T, cb = b - rb * T;
        const long ld = isb ? ldb : lda;
        const int col = (isb ? k0 : n0) + cb * 16 + half * 8;
        src[j] = reinterpret_cast<const char*>(isb ? B : A) + (((long)c0 * RC + rb * 16 + r) * ld + col) * 2;
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)wd_smem);
    auto issue = [&](int slot) {
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            wgd_dma16(src[j], __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(slot * SLOT * 2 + (wave + 4 * j) * 1024)));
            src[j] += j >= IPW / 2 ? stepB : stepA;
        }
    };
    // ---- accumulators -----------------------------------------------------------------------------------------------------------
    const bool bias_out = dbias_flag != nullptr && bz == 0;
    const bool bias_any = ln || bias_out;
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
    auto mfma_chunk = [&](int slot) {
        const unsigned short* pdy = wd_smem + slot * SLOT + (wn * WA) * BLK + loff;
        const unsigned short* px = wd_smem + slot * SLOT + SA + (wk * WB) * BLK + loff;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int st = 2 * ks;
            s8v pa[WA], pb[WB];
#pragma unroll
            for (int a = 0; a < WA; ++a) {
                const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(pdy + (st * T + a) * BLK));
                const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(pdy + ((st + 1) * T + a) * BLK));
                pa[a] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int b = 0; b < WB; ++b) {
                const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(px + (st * T + b) * BLK));
                const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(px + ((st + 1) * T + b) * BLK));
                pb[b] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int a = 0; a < WA; ++a)
#pragma unroll
                for (int b = 0; b < WB; ++b) acc[a][b] = mfma32_bf16(pa[a], pb[b], acc[a][b]);
            if (bias_any) {
#pragma unroll
                for (int a = 0; a < WA; ++a)
                    if (ln || a % NWK == wk) bacc[a] = mfma32_bf16(pa[a], ones, bacc[a]);
            }
        }
    };
    // ---- chunk stream: NS - 1 chunks in flight; iteration c: own DMA of chunk c landed (counted vmcnt) -> barrier (everyone's landed, and
    // everyone has finished reading slot (c - 1) % NS) -> refill that slot with chunk c + NS - 1 -> MFMAs on slot c % NS -----------------
#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (p < nch) issue(p);
    for (int c = 0; c < nch; ++c) {
        const int ahead = min(nch - 1 - c, NS - 2);            // chunks issued after chunk c that may stay in flight
        if (ahead >= 2) wgd_wait_vm<2 * IPW>(); else if (ahead == 1) wgd_wait_vm<IPW>(); else wgd_wait_vm<0>();
        if (!(dbg & 4)) __builtin_amdgcn_s_barrier();
        if (c + NS - 1 < nch && !(dbg & 2)) issue((c + NS - 1) % NS);
        if (!(dbg & 1)) mfma_chunk(c % NS);
    }
    // ---- epilogue: as wgrad_wide_bf16_kernel (acc[a][b][r] = element (n = 4q + r, k = i) of tile (a, b)) ---------------------------------
    if (ln) {
#pragma unroll
        for (int b = 0; b < WB; ++b) {
            const int k = k0 + 16 * (wk * WB + b) + i;
            const float g = k < K ? ln_w[k] : 0.f, sh = k < K ? ln_b[k] : 0.f;
#pragma unroll
            for (int a = 0; a < WA; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[a][b][r] = fmaf(g, acc[a][b][r], sh * bacc[a][r]);
        }
    }
    constexpr int NW = NWN * NWK;
    const long tile = (long)by * nz + bz, ntiles = ntl;
    f4* dst = part + (((long)bx * ntiles + tile) * NW + (wk * NWN + wn)) * (NSL * 64) + lane;
#pragma unroll
    for (int a = 0; a < WA; ++a) {
#pragma unroll
        for (int b = 0; b < WB; ++b) dst[(a * WB + b) * 64] = acc[a][b];
        if (bias_out && a % NWK == wk) dst[(WA * WB + a) * 64] = bacc[a];
    }
}

constexpr int kWgdRC = 64;
constexpr size_t kWgdOperandBytes = (size_t)600 << 20;       // bf16 copies of both operands of the largest covered launch (338 k x (512 + 128))

// tile edge (in 16-column blocks) for an N x K weight: 96 x 96 where the dimensions are multiples of 96 (RVT-T / -S: 32 / 48 channels x 2^s),
// else 128 x 128 (RVT-B: 64 x 2^s), else 64 x 64; 0: not covered
static inline int wgd_tile(int N, int K) {
    if (N % 96 == 0 && K % 96 == 0) return 6;
    if (N % 128 == 0 && K % 128 == 0) return 8;
    if (N % 64 == 0 && K % 64 == 0) return 4;
    return 0;
}
static inline bool wgrad_dma_ok(const XRows& xl, long lddy, int M, int N, int K, int dyfmt) {
    if (leod_precision() != 1 || M < 8192 || (M % kWgdRC) || N < 64 || K < 64) return false;
    const int T = wgd_tile(N, K);
    if (!T) return false;
    const int xm = xl.x_mode();
    if ((lddy & 7) || (xl.ld & 7)) return false;
    if (xl.x2 && (xm != 0 || (xl.K1 & 7) || (xl.ld2 & 7))) return false;
    const size_t need = (size_t)M * ((dyfmt ? 0 : N) + (xm == 3 ? 0 : K)) * 2;
    if (need > kWgdOperandBytes) return false;
    // row ranges above these stay on the register-staged wide kernel, which streams at 2-2.9 TB/s there and needs no preparation pass
    // (profiles/r06_f_wgrad_dma_1mpx_kbench.txt: RVT-B 1 Mpx stages 2-4, 338 k / 84 k / 21 k rows, 1262 / 1161 / 1719 us -> 948 / 607 / 419;
    // RVT-S Gen1 stage 2, 215 k rows x 96-multiples: 375 us wide vs 462 us here)
    if (M > (T == 6 ? 60000 : 400000)) return false;
    // the square projections (fp32 dY AND 16-bit attention rows to convert, 4-16 output tiles) stay on the register-staged kernel: the
    // preparation pass is as long as the contraction there (30 vs 36 us at 53 760 x 192 x 192, 27 vs 28 us at 13 440 x 384 x 384)
    if (!dyfmt && xm != 3 && xm != 2 && (long)N * K < 200000) return false;
    return true;
}

// the partial tiles of up to 4 problems -> dW / dbias (wgrad_wide_reduce_kernel<T, T, 2, 2> with a problem table: blockIdx.z)
struct WgrProb { const f4* part; float* dW; float* dbias; long ldw; int gx, ny, nz, per, N, K, ob, groups; };
struct WgrGroup { WgrProb p[4]; };
template <int T>
__global__ __launch_bounds__(256) void wgrad_dma_reduce_kernel(WgrGroup grp) {
    const WgrProb& pr = grp.p[blockIdx.z];
    if ((int)blockIdx.x >= pr.ob || (int)blockIdx.y >= pr.groups) return;
    constexpr int NWN = 2, NWK = 2, NW = 4, WA = T / 2, WB = T / 2, NT9 = WA * WB, NSL = NT9 + WA;
    const int gx = pr.gx, nz = pr.nz, per = pr.per, N = pr.N, K = pr.K;
    float* dW = pr.dW; float* dbias = pr.dbias; const long ldw = pr.ldw;
    const long O = (long)pr.ny * nz * NW * NSL * 64;
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    if (o >= O) return;
    const int lane = (int)(o & 63), t = (int)((o >> 6) % NSL), w = (int)((o / (64 * NSL)) % NW), tile = (int)(o / (64 * NSL * NW));
    const int ty = tile / nz, tz = tile - ty * nz, wn = w % NWN, wk = w / NWN, i = lane & 15, q = lane >> 4;
    if (t >= NT9 && !(dbias != nullptr && tz == 0 && (t - NT9) % NWK == wk && i == 0)) return;
    const int g0 = blockIdx.y * per, g1 = min(gx, g0 + per);
    f4 s0 = zero4(), s1 = zero4(), s2 = zero4(), s3 = zero4(), s4 = zero4(), s5 = zero4(), s6 = zero4(), s7 = zero4();
    const f4* p = pr.part + (long)g0 * O + o;
    int g = g0;
    for (; g + 8 <= g1; g += 8, p += 8 * O) {          // eight independent 16-byte loads in flight per thread
        s0 += p[0]; s1 += p[O]; s2 += p[2 * O]; s3 += p[3 * O]; s4 += p[4 * O]; s5 += p[5 * O]; s6 += p[6 * O]; s7 += p[7 * O];
    }
    for (; g + 4 <= g1; g += 4, p += 4 * O) { s0 += p[0]; s1 += p[O]; s2 += p[2 * O]; s3 += p[3 * O]; }
    for (; g < g1; ++g, p += O) s0 += p[0];
    const f4 v = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
    const bool single = pr.groups == 1;
    if (t < NT9) {
        const int a = t / WB, b = t - a * WB;
        const int k = tz * T * 16 + 16 * (wk * WB + b) + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = ty * T * 16 + 16 * (wn * WA + a) + 4 * q + r;
            if (n < N && k < K) {
                float* d = dW + (long)n * ldw + k;
                if (single) *d += v[r]; else atomicAdd(d, v[r]);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = ty * T * 16 + 16 * (wn * WA + (t - NT9)) + 4 * q + r;
            if (n < N) { if (single) dbias[n] += v[r]; else atomicAdd(dbias + n, v[r]); }
        }
    }
}

struct WgdHostProb { const void* dy; long lddy; XRows xl; float* dW; long ldw; float* dbias; int N, K, dyfmt; };

// n <= 4 problems of ONE row count M and one tile edge T: one preparation launch, one contraction launch, one reduce launch
template <int T, int RC, int NS>
static inline int launch_wgrad_dma_group_t(int n, const WgdHostProb* hp, int M, hipStream_t s) {
    constexpr int TW = T * 16;
    constexpr int LDS = wgd_lds_bytes<T, RC, NS>();
    constexpr int dbg = 0;          // ablation bits (1: no MFMAs, 2: no refills, 4: no barrier), compile-time, for experiments
    constexpr int NW = 4, NSL = (T / 2) * (T / 2) + T / 2;
    const int chunks = M / RC;
    const int target = 256 * wgd_occupancy<T, RC, NS>();       // resident workgroups
    int total_tiles = 0;
    for (int k = 0; k < n; ++k) total_tiles += (hp[k].N / TW) * (hp[k].K / TW);
    // ---- geometry + scratch layout --------------------------------------------------------------------------------------------------
    WgdGroup g{};
    WgrGroup rg{};
    Prep16Group pg{};
    size_t off = 0, offs_part[4], offs_a[4], offs_b[4];
    int blk = 0, obmax = 0, grmax = 0, c8max = 0;
    bool any_ln = false, any_prep = false;
    for (int k = 0; k < n; ++k) {
        const WgdHostProb& h = hp[k];
        const int xm = h.xl.x_mode();
        const int tiles = (h.N / TW) * (h.K / TW);
        // row ranges a multiple of 8: the tiles of one row range share an XCD (53 760 x 576 x 192: 62 -> 44 us)
        int gx = max(1, min(chunks / NS, target / total_tiles));
        if (gx >= 8) gx &= ~7;
        const int cpw = cdiv(chunks, gx);
        if (gx < 8) gx = cdiv(chunks, cpw);
        const long O = (long)tiles * NW * NSL * 64;            // f4 per partial
        offs_part[k] = off; off += ((size_t)gx * O * sizeof(f4) + 1023) & ~(size_t)1023;
        offs_a[k] = off; off += h.dyfmt ? 0 : (((size_t)M * h.N * 2 + 1023) & ~(size_t)1023);
        offs_b[k] = off; off += xm == 3 ? 0 : (((size_t)M * h.K * 2 + 1023) & ~(size_t)1023);
        g.p[k].cpw = cpw; g.p[k].gx = gx; g.p[k].ny = h.N / TW; g.p[k].nz = h.K / TW; g.p[k].N = h.N; g.p[k].K = h.K; g.p[k].blk0 = blk;
        blk += gx * tiles;
        const int ob = (int)cdiv(O, 256);
        int groups = max(1, min(gx, 512 / ob));
        const int per = cdiv(gx, groups);
        groups = cdiv(gx, per);
        rg.p[k] = WgrProb{nullptr, h.dW, h.dbias, h.ldw, gx, h.N / TW, h.K / TW, per, h.N, h.K, ob, groups};
        obmax = max(obmax, ob); grmax = max(grmax, groups);
        any_ln = any_ln || xm == 1;
    }
    g.n = n;
    char* ws = reinterpret_cast<char*>(wgrad_wide_scratch(s, off + 4096));
    if (!ws) return LEOD_ERR_UNSUPPORTED;
    for (int k = 0; k < n; ++k) {
        const WgdHostProb& h = hp[k];
        const int xm = h.xl.x_mode();
        f4* part = reinterpret_cast<f4*>(ws + offs_part[k]);
        unsigned short* a16 = reinterpret_cast<unsigned short*>(ws + offs_a[k]);
        unsigned short* b16 = reinterpret_cast<unsigned short*>(ws + offs_b[k]);
        Prep16 pa{}, pb{};
        pa.mode = pb.mode = -1;
        if (!h.dyfmt) pa = Prep16{h.dy, h.lddy, nullptr, 0, 0, nullptr, a16, h.N, 0};
        if (xm != 3) pb = Prep16{h.xl.x, h.xl.ld, h.xl.x2, h.xl.ld2, h.xl.x2 ? h.xl.K1 : h.K, h.xl.stats, b16, h.K,
                                 xm == 1 ? 1 : xm == 2 ? 2 : xm == 4 ? 3 : (h.xl.x2 ? 4 : 0)};
        pg.a[k] = pa; pg.b[k] = pb;
        if (pa.mode >= 0) { any_prep = true; c8max = max(c8max, h.N / 8); }
        if (pb.mode >= 0) { any_prep = true; c8max = max(c8max, h.K / 8); }
        g.p[k].A = h.dyfmt ? reinterpret_cast<const unsigned short*>(h.dy) : a16;
        g.p[k].lda = h.dyfmt ? h.lddy : h.N;
        g.p[k].B = xm == 3 ? reinterpret_cast<const unsigned short*>(h.xl.x) : b16;
        g.p[k].ldb = xm == 3 ? h.xl.ld : h.K;
        g.p[k].ln_w = xm == 1 ? h.xl.ln_w : nullptr; g.p[k].ln_b = xm == 1 ? h.xl.ln_b : nullptr;
        g.p[k].dbias_flag = h.dbias; g.p[k].part = part;
        rg.p[k].part = part;
    }
    for (int k = n; k < 4; ++k) { pg.a[k].mode = pg.b[k].mode = -1; }
    if (any_prep) hipLaunchKernelGGL(wgrad_prep16_kernel, dim3(cdiv((long)M * c8max, 256), 2, n), dim3(256), 0, s, pg, M);
    if (any_ln) {
        auto kern = wgrad_dma_kernel<T, RC, NS, true>;
        static bool attr_set = false;
        if (!attr_set) { hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr_set = true; }
        hipLaunchKernelGGL(kern, dim3(blk), dim3(256), LDS, s, g, chunks, dbg);
    } else {
        auto kern = wgrad_dma_kernel<T, RC, NS, false>;
        static bool attr_set = false;
        if (!attr_set) { hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr_set = true; }
        hipLaunchKernelGGL(kern, dim3(blk), dim3(256), LDS, s, g, chunks, dbg);
    }
    hipLaunchKernelGGL(wgrad_dma_reduce_kernel<T>, dim3(obmax, grmax, n), dim3(256), 0, s, rg);
    return leod_launch_status();
}
template <int T, int RC, int NS>
static inline int launch_wgrad_dma_t(const void* dy, long lddy, const XRows& xl, float* dW, long ldw, float* dbias,
                                     int M, int N, int K, hipStream_t s, int dyfmt) {
    const WgdHostProb h{dy, lddy, xl, dW, ldw, dbias, N, K, dyfmt};
    return launch_wgrad_dma_group_t<T, RC, NS>(1, &h, M, s);
}
static inline int launch_wgrad_dma(const void* dy, long lddy, const XRows& xl, float* dW, long ldw, float* dbias,
                                   int M, int N, int K, hipStream_t s, int dyfmt) {
    // (192 x 192 tiles -- T = 12, one workgroup per CU, half the LDS-DMA bytes per output -- measured slower on 9 of the 10 shapes of
    // tools/kbench_wgrad_small.py: 593 vs 549 us summed; the chunk stream is not bound by DMA bytes, profiles/r06_d_wgrad_dma_kbench.txt)
    // 128 x 128 tiles: two ring slots of 64 rows (2 workgroups per CU) beat three slots of 32 or 64 rows and 64 x 64 tiles on every shape of
    // RVT-B (3597 vs 3703 / 4054 / 4392 us over the twenty launches of profiles/r06_f_wgrad_dma_1mpx_kbench.txt)
    switch (wgd_tile(N, K)) {
        case 6: return launch_wgrad_dma_t<6, 64, 3>(dy, lddy, xl, dW, ldw, dbias, M, N, K, s, dyfmt);
        case 8: return launch_wgrad_dma_t<8, 64, 2>(dy, lddy, xl, dW, ldw, dbias, M, N, K, s, dyfmt);
        default: return launch_wgrad_dma_t<4, 64, 3>(dy, lddy, xl, dW, ldw, dbias, M, N, K, s, dyfmt);
    }
}
