// Direct 3x3 / stride-1 / pad-1 convolution on NHWC maps for the PAFPN and the YOLOX head (yolo_pafpn.py:109-140,
// yolo_head.py:208-222; BaseConv = conv -> BatchNorm -> SiLU, network_blocks.py:29-51), precision mode bf16.
//
// The implicit GEMM of k_conv.hip rebuilds every im2col operand chunk from global memory: 18 load -> stash -> barrier rounds of
// 9 MFMAs per wave for a 96 -> 96 conv, 55 / 77 us for 6.8 GFLOP (forward / dgrad of a 32 x 40 head level on 32 frames).  Here a
// workgroup owns RH full output rows of one image (<= 160 pixels = 10 MFMA row tiles):
//   * the (RH + 2) x (W + 2) x Cin input halo is converted to bf16 and copied into LDS ONCE; the nine taps are nine shifted
//     views of it (per-lane base address + a tap offset), no im2col;
//   * the weights of one tap ([<= 96 out channels][Cin] bf16, packed once per call) stream through a double-buffered LDS tile with
//     the next tap's loads in flight under the current tap's MFMAs: one barrier per tap;
//   * every wave keeps up to 3 row tiles x 6 column tiles of accumulators; a B fragment read feeds up to 3 MFMAs;
//   * epilogue: fp32 rows through a wave-private LDS tile (16-byte stores), BatchNorm (sum, sumsq) per channel from the accumulator
//     layout with one atomic per channel and wave into the replicated statistic block.
// The same kernel is the dgrad (dx = conv(dy, W flipped and transposed)): only the weight packing differs.
#include <stdlib.h>

#include "common.hpp"
#include "conv3.hpp"

typedef unsigned short bf16_t;

// w [N][Cin][3][3] fp32 -> bf16, K-contiguous per tap.
//   mode 0 (forward): wp[tap][n][c]                      B rows = output channels, k = input channel
//   mode 1 (dgrad)  : wp[8 - tap][c][n]                  B rows = input channels (the dgrad's outputs), k = output channel
__global__ __launch_bounds__(256) void conv3_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int N, int Cin, int mode) {
    const long total = (long)N * Cin * 9;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int tap = (int)(e % 9); const long r = e / 9; const int c = (int)(r % Cin), n = (int)(r / Cin);
        const long o = mode == 0 ? ((long)tap * N + n) * Cin + c : ((long)(8 - tap) * Cin + c) * N + n;
        const f2_ p = {w[e], 0.f};
        out[o] = (bf16_t)(__builtin_bit_cast(unsigned, __builtin_convertvector(p, bf2_)) & 0xffffu);
    }
}

// KC = input channels / 16, NTO = output column tiles (16 channels each) per workgroup, TW = row tiles per wave (ceil(10 / 4))
// K32 (Cin % 32 == 0): v_mfma_f32_16x16x32_bf16 (8 bf16 per lane and operand, 16-byte fragment reads) -- on gfx950 the 16x16x16 form
// issues at the same 16 cycles per instruction, i.e. at half the bf16 MFMA rate

template <int KC, int NTO, int TW>
__global__ __launch_bounds__(256) void conv3s1_kernel(const float* __restrict__ x, const bf16_t* __restrict__ wp, float* __restrict__ y,
                                                       double* __restrict__ colstats, int stat_rep, int accumulate,
                                                       int B, int H, int W, int Cout, int RH) {
    constexpr bool K32 = KC % 2 == 0;
    // operand rows (halo pixels, weight rows): bf16 rows whose dword stride is 4 * odd for the 8-byte fragment reads of the 16-k MFMA
    // and == 8 (mod 16) for the 16-byte reads of the 32-k MFMA (conflict-free ds_read_b64 / ds_read_b128, MI355X_MICROARCH.md LDS)
    constexpr int CI = 16 * KC, LDP = CI + (K32 ? 16 : 8), LDB = LDP, BN = NTO * 16, LDO = BN + 4;
    constexpr int QK = K32 ? 8 : 4;                               // k elements per lane and fragment
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int rblocks = (H + RH - 1) / RH;
    const int b = blockIdx.x / rblocks, y0 = (blockIdx.x - b * rblocks) * RH;
    const int rows = min(RH, H - y0);
    const int P = rows * W, ntiles = (P + 15) >> 4;
    const int co0 = blockIdx.y * BN;
    const int WH = W + 2;
    const int halo_elems = (RH + 2) * WH * LDP;
    bf16_t* halo = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* sB = halo + ((halo_elems + 7) & ~7);                 // [2][BN * LDB]
    // ---- weights of tap 0 start flying first ------------------------------------------------------------------------------
    constexpr int BSLOT = BN * (CI / 8);                          // 16-byte slots of one tap's weight tile
    constexpr int RB = (BSLOT + 255) / 256;
    typedef int i4 __attribute__((ext_vector_type(4)));
    // three register sets: the weights of taps t + 1 .. t + 3 are in flight / waiting while tap t computes (one tap's MFMAs are
    // ~0.4 us per wave, an L2 round trip is > 1 us: with one tap of look-ahead every tap waited for its weights)
    constexpr int PD = 3;
    i4 rb[PD][RB];
    auto fetch_b = [&](int tap, i4 (&r)[RB]) {
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            const int e = tid + 256 * p;
            if (e < BSLOT) {
                const int n = e / (CI / 8), c8 = (e - n * (CI / 8)) * 8;
                r[p] = *reinterpret_cast<const i4*>(wp + ((long)tap * Cout + co0 + n) * CI + c8);
            }
        }
    };
    auto stash_b = [&](int buf, const i4 (&r)[RB]) {
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            const int e = tid + 256 * p;
            if (e < BSLOT) {
                const int n = e / (CI / 8), c8 = (e - n * (CI / 8)) * 8;
                *reinterpret_cast<i4*>(sB + buf * (BN * LDB) + n * LDB + c8) = r[p];
            }
        }
    };
    fetch_b(0, rb[0]);
    fetch_b(1, rb[1]);
    fetch_b(2, rb[2]);
    // ---- input halo: rows y0 - 1 .. y0 + RH, columns -1 .. W, zero outside the image -------------------------------------------
    // (loads of a batch are all issued before the first LDS store: a load -> store loop pays one memory round trip per iteration)
    const int hslots = (rows + 2) * WH * (CI / 4);
    const float* xb = x + (long)b * H * W * CI;
    constexpr int HB = 12;
    for (int e0 = tid; e0 < hslots; e0 += 256 * HB) {
        f4 hv[HB]; int ho[HB];
#pragma unroll
        for (int j = 0; j < HB; ++j) {
            const int e = e0 + 256 * j;
            const int hp = e / (CI / 4), c4 = (e - hp * (CI / 4)) * 4;
            const int hy = hp / WH, hx = hp - hy * WH;
            const int iy = y0 - 1 + hy, ix = hx - 1;
            ho[j] = e < hslots ? hp * LDP + c4 : -1;
            hv[j] = zero4();
            if (e < hslots && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) hv[j] = ld4(xb + ((long)iy * W + ix) * CI + c4);
        }
#pragma unroll
        for (int j = 0; j < HB; ++j)
            if (ho[j] >= 0) *reinterpret_cast<s4*>(halo + ho[j]) = pack_bf16(hv[j]);
    }
    stash_b(0, rb[0]);
    fetch_b(3, rb[0]);
    // ---- this wave's row tiles: lane i of tile t is output pixel p = 16 t + i of the region ---------------------------------
    int hbase[TW]; bool tok[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        const int tile = wave + 4 * t;
        tok[t] = tile < ntiles;
        int p = tile * 16 + i;
        if (p >= P) p = 0;
        const int py = p / W, px = p - py * W;
        hbase[t] = ((py + 1) * WH + px + 1) * LDP + QK * q;
    }
    f4 acc[TW][NTO];
#pragma unroll
    for (int t = 0; t < TW; ++t)
#pragma unroll
        for (int n = 0; n < NTO; ++n) acc[t][n] = zero4();
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int buf = tap & 1;
        const int dy = tap / 3, dx = tap - 3 * dy;
        const int toff = ((dy - 1) * WH + (dx - 1)) * LDP;
        const bf16_t* pb = sB + buf * (BN * LDB) + i * LDB + QK * q;
        if constexpr (K32) {
#pragma unroll
            for (int kc = 0; kc < KC / 2; ++kc) {
                s8v a[TW];
#pragma unroll
                for (int t = 0; t < TW; ++t) a[t] = *reinterpret_cast<const s8v*>(halo + hbase[t] + toff + 32 * kc);
#pragma unroll
                for (int n = 0; n < NTO; ++n) {
                    const s8v bv = *reinterpret_cast<const s8v*>(pb + 16 * n * LDB + 32 * kc);
#pragma unroll
                    for (int t = 0; t < TW; ++t) acc[t][n] = mfma32_bf16(a[t], bv, acc[t][n]);   // absent tiles compute on pixel 0 (discarded)
                }
            }
        } else {
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                s4 a[TW];
#pragma unroll
                for (int t = 0; t < TW; ++t) a[t] = *reinterpret_cast<const s4*>(halo + hbase[t] + toff + 16 * kc);
#pragma unroll
                for (int n = 0; n < NTO; ++n) {
                    const s4 bv = *reinterpret_cast<const s4*>(pb + 16 * n * LDB + 16 * kc);
#pragma unroll
                    for (int t = 0; t < TW; ++t) acc[t][n] = mfma16_bf16(a[t], bv, acc[t][n]);
                }
            }
        }
        if (tap + 1 < 9) stash_b(buf ^ 1, rb[(tap + 1) % PD]);      // arrived two taps ago
        if (tap + 4 < 9) fetch_b(tap + 4, rb[(tap + 1) % PD]);      // its register set is free again
        __syncthreads();
    }
    // ---- epilogue ----------------------------------------------------------------------------------------------------------
    float* so = reinterpret_cast<float*>(smem_raw) + wave * 16 * LDO;     // wave-private 16 x BN tile (the halo is done with)
    float cs[NTO], cq[NTO];
#pragma unroll
    for (int n = 0; n < NTO; ++n) { cs[n] = 0.f; cq[n] = 0.f; }
    float* yb = y + ((long)(b * H + y0) * W) * Cout + co0;
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        if (!tok[t]) continue;                                             // wave-uniform
        const int tile = wave + 4 * t;
#pragma unroll
        for (int n = 0; n < NTO; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[t][n][r];
                so[(4 * q + r) * LDO + 16 * n + i] = v;
                if (tile * 16 + 4 * q + r < P) { cs[n] += v; cq[n] += v * v; }
            }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < (16 * NTO * 4 + 63) / 64; ++k) {
            const int idx = k * 64 + lane;
            const int row = idx / (NTO * 4), c4 = (idx - row * (NTO * 4)) * 4;
            const int p = tile * 16 + row;
            if (idx < 16 * NTO * 4 && p < P) {
                f4 v = *reinterpret_cast<const f4*>(so + row * LDO + c4);
                float* dst = yb + (long)p * Cout + c4;
                if (accumulate) v += ld4(dst);
                *reinterpret_cast<f4*>(dst) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    if (colstats) {
        double* cst = colstats + (stat_rep > 1 ? (size_t)((blockIdx.x * 4 + wave) & (stat_rep - 1)) * 2 * Cout : 0);
#pragma unroll
        for (int n = 0; n < NTO; ++n) {
            const float a = quad16_sum(cs[n]), c = quad16_sum(cq[n]);
            if (q == 0) { atomicAdd(cst + co0 + 16 * n + i, (double)a); atomicAdd(cst + Cout + co0 + 16 * n + i, (double)c); }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same convolution: dW[n][c][tap] += sum_p dy[p][n] x[p + off(tap)][c].
// A workgroup stages the bf16 input halo and the bf16 dy rows of a region once and contracts over the region's pixels for the three
// taps of one kernel row (blockIdx.y): both MFMA operands are COLUMNS of pixel-major LDS tiles, read with ds_read_b64_tr_b16 (the
// lane supplies the address of "its" pixel row, the hardware transposes 4 pixels x 16 channels per 16-lane group); the dy
// fragments are shared by the three taps.  Each wave keeps a 3 x 3 block of 16 x 16 tiles per tap (27 accumulator tiles), the
// workgroup walks over regions r, r + gridDim.x, ... and writes its partial sums with plain stores to part[worker][tap][n][c]
// (c contiguous); conv3_wgrad_reduce_kernel adds the workers' partials into dW[n][c][tap].  No atomics: scattered 4-byte atomics into
// the [n][c][3][3] layout ran at ~28 per ns (the atomic version of this kernel took 200 us, 94 us of it for 2.6 M atomics), and the
// result is bitwise reproducible.
// NA = N / 16, NB = Cin / 16 (both even: the 4 waves split the (n, c) tile grid 2 x 2).
// ---------------------------------------------------------------------------------------------------------------------
template <int NA, int NB>
__global__ __launch_bounds__(256) void conv3s1_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ part,
                                                             int B, int H, int W, int RH, int nregions) {
    static_assert(NA % 2 == 0 && NB % 2 == 0, "the 4 waves split the tile grid 2 x 2");
    constexpr int N = 16 * NA, CI = 16 * NB, LDX = CI + 16, LDY = N + 16, TA = NA / 2, TB = NB / 2, MAXS = 5;
    typedef __attribute__((address_space(3))) s4 lds_s4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int wa = wave & 1, wb = wave >> 1;
    const int g = blockIdx.y;                                     // kernel row: taps 3g .. 3g + 2
    const int WH = W + 2;
    bf16_t* halo = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* sdy = halo + (((RH + 2) * WH * LDX + 7) & ~7);
    f4 acc[3][TA][TB];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int b = 0; b < TB; ++b) acc[j][a][b] = zero4();
    const int rblocks = (H + RH - 1) / RH;
    for (int reg = blockIdx.x; reg < nregions; reg += gridDim.x) {
        const int b = reg / rblocks, y0 = (reg - b * rblocks) * RH;
        const int rows = min(RH, H - y0);
        const int P = rows * W, P32 = (P + 31) & ~31, steps = P32 >> 5;
        // ---- stage the input halo and the dy rows (bf16); all loads of a batch before the first LDS store -----------------------
        const int hslots = (rows + 2) * WH * (CI / 4);
        const float* xb = x + (long)b * H * W * CI;
        constexpr int HB = 12;
        for (int e0 = tid; e0 < hslots; e0 += 256 * HB) {
            f4 hv[HB]; int ho[HB];
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                const int e = e0 + 256 * j;
                const int hp = e / (CI / 4), c4 = (e - hp * (CI / 4)) * 4;
                const int hy = hp / WH, hx = hp - hy * WH;
                const int iy = y0 - 1 + hy, ix = hx - 1;
                ho[j] = e < hslots ? hp * LDX + c4 : -1;
                hv[j] = zero4();
                if (e < hslots && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) hv[j] = ld4(xb + ((long)iy * W + ix) * CI + c4);
            }
#pragma unroll
            for (int j = 0; j < HB; ++j)
                if (ho[j] >= 0) *reinterpret_cast<s4*>(halo + ho[j]) = pack_bf16(hv[j]);
        }
        const int dslots = P32 * (N / 4);
        const float* dyb = dy + ((long)(b * H + y0) * W) * N;
        for (int e0 = tid; e0 < dslots; e0 += 256 * HB) {
            f4 hv[HB];
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                const int e = e0 + 256 * j;
                const int p = e / (N / 4);
                hv[j] = (e < dslots && p < P) ? ld4(dyb + (long)e * 4) : zero4();       // rows P .. P32 - 1: zeros (contribute nothing)
            }
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                const int e = e0 + 256 * j;
                if (e < dslots) { const int p = e / (N / 4), c4 = (e - p * (N / 4)) * 4; *reinterpret_cast<s4*>(sdy + p * LDY + c4) = pack_bf16(hv[j]); }
            }
        }
        __syncthreads();
        // ---- contraction over the region's pixels, 32 per MFMA ----------------------------------------------------------------------
        for (int s = 0; s < steps; ++s) {
            // the two pixel rows this lane addresses in a transpose read: p = 32 s + 8 q + (i >> 2) (+ 4)
            int hx0[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int p = 32 * s + 8 * q + 4 * h + (i >> 2);
                if (p >= P) p = P - 1;                                               // dy is zero there: any valid pixel will do
                const int py = p / W, px = p - py * W;
                hx0[h] = ((py + g) * WH + px) * LDX + 4 * (i & 3);                   // tap (g, 0); taps (g, 1), (g, 2): + LDX, + 2 LDX
            }
            const bf16_t* pdy = sdy + (32 * s + 8 * q + (i >> 2)) * LDY + 4 * (i & 3);
            s8v av[TA];
#pragma unroll
            for (int a = 0; a < TA; ++a) {
                const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(pdy + 16 * (TA * wa + a)));
                const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(pdy + 4 * LDY + 16 * (TA * wa + a)));
                av[a] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
#pragma unroll
                for (int bb = 0; bb < TB; ++bb) {
                    const int co = j * LDX + 16 * (TB * wb + bb);
                    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(halo + hx0[0] + co));
                    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(halo + hx0[1] + co));
                    const s8v bv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                    for (int a = 0; a < TA; ++a) acc[j][a][bb] = mfma32_bf16(av[a], bv, acc[j][a][bb]);
                }
            }
        }
        __syncthreads();                                                             // the tiles are restaged for the next region
    }
    // ---- part[worker][3g + j][n][c] = acc: row 4q + r of tile (a, b) is output channel n, column i is input channel c ----------------
    float* pw = part + (long)blockIdx.x * 9 * N * CI;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int bb = 0; bb < TB; ++bb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = 16 * (TA * wa + a) + 4 * q + r, c = 16 * (TB * wb + bb) + i;
                    pw[((long)(3 * g + j) * N + n) * CI + c] = acc[j][a][bb][r];
                }
}

// dW[n][c][tap] += sum over workers of part[worker][tap][n][c]
__global__ __launch_bounds__(256) void conv3_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dW, int workers, int N, int CI) {
    const int total = 9 * N * CI;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    float s = 0.f;
#pragma unroll 8
    for (int w = 0; w < workers; ++w) s += part[(long)w * total + e];             // 8 independent loads in flight per thread
    const int c = e % CI, r = e / CI; const int n = r % N, tap = r / N;
    dW[((long)n * CI + c) * 9 + tap] += s;
}

static inline size_t conv3_smem(int RH, int W, int Cin, int nto) {
    const int LDP = Cin + (Cin % 32 == 0 ? 16 : 8);
    const size_t halo = (((size_t)(RH + 2) * (W + 2) * LDP + 7) & ~(size_t)7) * 2;
    const size_t sb = (size_t)2 * nto * 16 * LDP * 2;
    const size_t so = (size_t)4 * 16 * (nto * 16 + 4) * 4;
    return max(halo + sb, so);
}
// output rows per workgroup: as many as give <= 160 pixels (10 row tiles) and fit the 160 KB of LDS; 0 = does not fit at all
static inline int conv3_rows_per_block(int H, int W, int Cin, int nto) {
    int rh = max(1, min(H, 160 / W));
    while (rh > 0 && conv3_smem(rh, W, Cin, nto) > 160 * 1024) --rh;
    return rh;
}

bool conv3s1_supported(int H, int W, int Cin, int Cout) {
    static const int on = getenv("LEOD_CONV3_DIRECT") ? atoi(getenv("LEOD_CONV3_DIRECT")) : 1;
    if (!on || leod_precision() != 1) return false;
    if (W > 160 || W < 4 || (Cin != 48 && Cin != 96 && Cin != 192) || (Cout != 48 && Cout != 96 && Cout != 192)) return false;
    return conv3_rows_per_block(H, W, Cin, Cout == 48 ? 3 : 6) > 0;
}

size_t conv3s1_pack_bytes(int Cin, int Cout) { return (size_t)9 * Cin * Cout * sizeof(bf16_t); }

// x [B,H,W,Cin] -> y [B,H,W,Cout].  transposed = 0: y = conv3x3(x, w[Cout][Cin][3][3]); 1: the dgrad of a conv whose weight is
// w[Cin][Cout][3][3] (x = dy).  wpack: scratch of conv3s1_pack_bytes.
int conv3s1_launch(const float* x, const float* w, float* y, double* colstats, int stat_rep, int accumulate, int B, int H, int W,
                   int Cin, int Cout, int transposed, void* wpack, hipStream_t stream) {
    bf16_t* wp = reinterpret_cast<bf16_t*>(wpack);
    const long total = (long)9 * Cin * Cout;
    // the packed layout is [tap][out rows][k]; for the dgrad the stored weight is [N = Cin of this call][C = Cout of this call]
    hipLaunchKernelGGL(conv3_pack_kernel, dim3((int)min((long)1024, (total + 255) / 256)), dim3(256), 0, stream, w, wp,
                       transposed ? Cin : Cout, transposed ? Cout : Cin, transposed ? 1 : 0);
    const int nto = Cout == 48 ? 3 : 6;
    const int RH = conv3_rows_per_block(H, W, Cin, nto);
    if (RH <= 0) return LEOD_ERR_UNSUPPORTED;
    const dim3 grid(B * cdiv(H, RH), Cout / (16 * nto));
    const size_t smem = conv3_smem(RH, W, Cin, nto);
    const int tw = cdiv(cdiv(min(RH, H) * W, 16), 4) <= 2 ? 2 : 3;          // row tiles per wave
#define C3_CASE(KCV, NTOV) C3_CASE2(KCV, NTOV, 2) C3_CASE2(KCV, NTOV, 3)
#define C3_CASE2(KCV, NTOV, TWV)                                                                                                     \
    if (Cin == 16 * KCV && nto == NTOV && tw == TWV) {                                                                                            \
        static bool attr_set = false;                                                                                                \
        if (!attr_set) {                                                                                                             \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3s1_kernel<KCV, NTOV, TWV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_set = true;                                                                                                         \
        }                                                                                                                            \
        hipLaunchKernelGGL((conv3s1_kernel<KCV, NTOV, TWV>), grid, dim3(256), smem, stream, x, wp, y, colstats, stat_rep, accumulate, B, H, W, Cout, RH); \
        return leod_launch_status();                                                                                                 \
    }
    C3_CASE(3, 3) C3_CASE(3, 6) C3_CASE(6, 3) C3_CASE(6, 6) C3_CASE(12, 3) C3_CASE(12, 6)
#undef C3_CASE
#undef C3_CASE2
    return LEOD_ERR_UNSUPPORTED;
}

bool conv3s1_wgrad_supported(int H, int W, int Cin, int Cout) {
    static const int on = getenv("LEOD_CONV3_DIRECT") ? atoi(getenv("LEOD_CONV3_DIRECT")) : 1;
    if (!on || leod_precision() != 1) return false;
    return W <= 160 && W >= 4 && Cin == 96 && Cout == 96;
}

static inline int conv3_wgrad_workers(int B, int H, int W, int Cin, int Cout, int* rh_out) {
    const int LDX = Cin + 16, LDY = Cout + 16;
    int RH = max(1, min(H, 160 / W));
    while (RH > 0) {
        const size_t halo = (((size_t)(RH + 2) * (W + 2) * LDX + 7) & ~(size_t)7) * 2;
        const size_t sdy = (size_t)((RH * W + 31) & ~31) * LDY * 2;
        if (halo + sdy <= 160 * 1024) break;
        --RH;
    }
    if (rh_out) *rh_out = RH;
    if (RH <= 0) return 0;
    static const int cap = getenv("LEOD_CONV3_WORKERS") ? atoi(getenv("LEOD_CONV3_WORKERS")) : 64;      // measured: 64 -> 42 us, 85 -> 47, 128 -> 51 (level-0 head conv)
    return min(B * cdiv(H, RH), cap);      // region workers per kernel row: each walks over nregions / workers regions
}

// floats of scratch conv3s1_wgrad_launch needs (the workers' partial sums)
size_t conv3s1_wgrad_workspace_floats(int B, int H, int W, int Cin, int Cout) {
    return (size_t)conv3_wgrad_workers(B, H, W, Cin, Cout, nullptr) * 9 * Cin * Cout;
}

// dW[Cout][Cin][3][3] += wgrad of y = conv3x3(x) for dy [B,H,W,Cout], x [B,H,W,Cin]; ws: conv3s1_wgrad_workspace_floats floats
int conv3s1_wgrad_launch(const float* dy, const float* x, float* dW, float* ws, int B, int H, int W, int Cin, int Cout, hipStream_t stream) {
    int RH = 0;
    const int workers = conv3_wgrad_workers(B, H, W, Cin, Cout, &RH);
    if (workers <= 0 || !ws) return LEOD_ERR_UNSUPPORTED;
    const int LDX = Cin + 16, LDY = Cout + 16;
    const size_t smem = (((size_t)(RH + 2) * (W + 2) * LDX + 7) & ~(size_t)7) * 2 + (size_t)((RH * W + 31) & ~31) * LDY * 2;
    const int nregions = B * cdiv(H, RH);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3s1_wgrad_kernel<6, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3s1_wgrad_kernel<6, 6>), dim3(workers, 3), dim3(256), smem, stream, dy, x, ws, B, H, W, RH, nregions);
    hipLaunchKernelGGL(conv3_wgrad_reduce_kernel, dim3(cdiv(9 * Cin * Cout, 256)), dim3(256), 0, stream, ws, dW, workers, Cout, Cin);
    return leod_launch_status();
}
