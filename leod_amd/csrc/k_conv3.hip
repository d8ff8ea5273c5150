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
// h16: the packed copy holds fp16 (forward packs of precision mode 16f)
__global__ __launch_bounds__(256) void conv3_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int N, int Cin, int mode, int h16) {
    const long total = (long)N * Cin * 9;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int tap = (int)(e % 9); const long r = e / 9; const int c = (int)(r % Cin), n = (int)(r / Cin);
        const long o = mode == 0 ? ((long)tap * N + n) * Cin + c : ((long)(8 - tap) * Cin + c) * N + n;
        const f2_ p = {w[e], 0.f};
        out[o] = h16 ? __builtin_bit_cast(bf16_t, (_Float16)w[e]) : (bf16_t)(__builtin_bit_cast(unsigned, __builtin_convertvector(p, bf2_)) & 0xffffu);
    }
}

// KC = input channels / 16, NTO = output column tiles (16 channels each) per workgroup, TW = row tiles per wave (ceil(10 / 4))
// K32 (Cin % 32 == 0): v_mfma_f32_16x16x32_bf16 (8 bf16 per lane and operand, 16-byte fragment reads) -- on gfx950 the 16x16x16 form
// issues at the same 16 cycles per instruction, i.e. at half the bf16 MFMA rate

// S = 2 (forward only): the stride-2 convs of the backbone / PAFPN -- output pixel (oy, ox) reads halo pixel (2 oy + ky, 2 ox + kx);
// H, W are the INPUT size, the pixel-row stride of the halo keeps the 16 lanes of a fragment read in different banks at DOUBLE the
// pixel distance (dword stride of 2 pixels == 8 (mod 16) for the 16-byte reads, 4 * odd for the 8-byte reads)
// eval-mode BaseConv: y = silu((conv - running_mean) * w / sqrt(running_var + eps) + b) applied to the finished rows (w == NULL: plain conv)
struct BnEval { const float* w; const float* b; const float* rm; const float* rv; float eps; };

// OF: operand format, 1 = bf16 (also the dgrad form of every mode), 2 = fp16 (forward of precision mode 16f; wp packed as fp16)
// One problem of a launch.  A launch carries up to 8 independent problems of one channel geometry (blockIdx.z): the convs of equal depth in
// the cls / reg towers of the three head levels are six launches' worth of work for one launch latency, and the small levels (2560 and 10240
// pixels) no longer leave most of the chip idle while they run (round 6).  nblocks: workgroups (blockIdx.x) this problem uses.
struct C3Prob { const float* x; const bf16_t* wp; float* y; double* colstats; int stat_rep, accumulate, B, H, W, RH, WSo, nblocks; };
struct C3Group { C3Prob p[8]; };

template <int KC, int NTO, int TW, int S = 1, int OF = 1>
__global__ __launch_bounds__(256) void conv3s1_kernel(C3Group grp, int Cout, BnEval bne) {
    const C3Prob& pr = grp.p[blockIdx.z];
    if ((int)blockIdx.x >= pr.nblocks) return;
    const float* __restrict__ x = pr.x; const bf16_t* __restrict__ wp = pr.wp; float* __restrict__ y = pr.y; double* __restrict__ colstats = pr.colstats;
    const int stat_rep = pr.stat_rep, accumulate = pr.accumulate, B = pr.B, H = pr.H, W = pr.W, RH = pr.RH, WSo = pr.WSo;
    (void)B;
    // WSo: output columns per workgroup (Wo % WSo == 0).  Maps too wide for a (RH + 2)-row halo of full rows in LDS (160 columns x 128
    // channels) are cut into column segments; a segment is a map of its own width for the halo and the pixel -> lane mapping, only the
    // global addresses know the full row.
    constexpr bool K32 = KC % 2 == 0;
    // operand rows (halo pixels, weight rows): bf16 rows whose dword stride is 4 * odd for the 8-byte fragment reads of the 16-k MFMA
    // and == 8 (mod 16) for the 16-byte reads of the 32-k MFMA (conflict-free ds_read_b64 / ds_read_b128, MI355X_MICROARCH.md LDS)
    constexpr int CI = 16 * KC, LDB = CI + (K32 ? 16 : 8), LDP = S == 1 ? LDB : CI + (K32 ? 8 : 4), BN = NTO * 16, LDO = BN + 4;
    constexpr int QK = K32 ? 8 : 4;                               // k elements per lane and fragment
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int Ho = H / S, Wo = W / S;
    const int rblocks = (Ho + RH - 1) / RH, csegs = Wo / WSo;
    const int seg = blockIdx.x % csegs, rb_ = blockIdx.x / csegs;
    const int b = rb_ / rblocks, y0 = (rb_ - b * rblocks) * RH;
    const int x0o = seg * WSo, x0 = S * x0o;
    const int rows = min(RH, Ho - y0);
    const int P = rows * WSo, ntiles = (P + 15) >> 4;
    const int co0 = blockIdx.y * BN;
    const int WH = S * WSo + 2;
    const int halo_elems = (S * (RH - 1) + 3) * WH * LDP;
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
    const int hslots = (S * (rows - 1) + 3) * WH * (CI / 4);
    const float* xb = x + (long)b * H * W * CI;
    constexpr int HB = 12;
    for (int e0 = tid; e0 < hslots; e0 += 256 * HB) {
        f4 hv[HB]; int ho[HB];
#pragma unroll
        for (int j = 0; j < HB; ++j) {
            const int e = e0 + 256 * j;
            const int hp = e / (CI / 4), c4 = (e - hp * (CI / 4)) * 4;
            const int hy = hp / WH, hx = hp - hy * WH;
            const int iy = S * y0 - 1 + hy, ix = x0 + hx - 1;
            ho[j] = e < hslots ? hp * LDP + c4 : -1;
            hv[j] = zero4();
            if (e < hslots && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) hv[j] = ld4(xb + ((long)iy * W + ix) * CI + c4);
        }
#pragma unroll
        for (int j = 0; j < HB; ++j)
            if (ho[j] >= 0) *reinterpret_cast<s4*>(halo + ho[j]) = pack16<OF>(hv[j]);
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
        const int py = p / WSo, px = p - py * WSo;
        hbase[t] = ((S * py + 1) * WH + S * px + 1) * LDP + QK * q;
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
                    for (int t = 0; t < TW; ++t) acc[t][n] = mfma32_16<OF>(a[t], bv, acc[t][n]);   // absent tiles compute on pixel 0 (discarded)
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
                    for (int t = 0; t < TW; ++t) acc[t][n] = mfma16_16<OF>(a[t], bv, acc[t][n]);
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
    float* yb = y + ((long)(b * Ho + y0) * Wo + x0o) * Cout + co0;
    auto poff = [&](int p) -> long { if (csegs == 1) return p; const int py = p / WSo; return (long)py * Wo + (p - py * WSo); };   // pixel of the region -> pixel of the map
    // the whole fragment loop once per value of `accumulate`: with the read-modify-write arm inside the row loop every row group ended at
    // a join with a pending load, where hipcc drains vmcnt -- i.e. each 16-byte store waited for the previous one's acknowledgement
    auto rows_out = [&](auto accc, auto bnc) {
        constexpr bool ACC = decltype(accc)::value;
        constexpr bool BNE = decltype(bnc)::value;
        constexpr int NK = (16 * NTO * 4 + 63) / 64;
        f4 bsc[BNE ? NK : 1], bsh[BNE ? NK : 1];               // folded scale / shift of this lane's column slots (one per store slot k)
        if constexpr (BNE) {
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int idx = min(k * 64 + lane, 16 * NTO * 4 - 1);
                const int c4 = (idx % (NTO * 4)) * 4 + co0;
                const f4 w4 = ld4(bne.w + c4), b4 = ld4(bne.b + c4), m4 = ld4(bne.rm + c4), v4 = ld4(bne.rv + c4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { bsc[k][j] = w4[j] * rsqrtf(v4[j] + bne.eps); bsh[k][j] = b4[j] - m4[j] * bsc[k][j]; }
            }
        }
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
            f4 old[ACC ? NK : 1];
            if constexpr (ACC) {                                               // all read-modify-write loads of the fragment first
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int idx = min(k * 64 + lane, 16 * NTO * 4 - 1);
                    const int row = idx / (NTO * 4), c4 = (idx - row * (NTO * 4)) * 4;
                    old[k] = ld4(yb + poff(min(tile * 16 + row, P - 1)) * Cout + c4);
                }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const int idx = k * 64 + lane;
                const int row = idx / (NTO * 4), c4 = (idx - row * (NTO * 4)) * 4;
                const int p = tile * 16 + row;
                if (idx < 16 * NTO * 4 && p < P) {
                    f4 v = *reinterpret_cast<const f4*>(so + row * LDO + c4);
                    if constexpr (ACC) v += old[k];
                    if constexpr (BNE) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = siluf_(fmaf(v[j], bsc[k][j], bsh[k][j]));
                    }
                    *reinterpret_cast<f4*>(yb + poff(p) * Cout + c4) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    };
    if (bne.w) rows_out(std::false_type{}, std::true_type{});
    else if (accumulate) rows_out(std::true_type{}, std::false_type{});
    else rows_out(std::false_type{}, std::false_type{});
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
// Input gradient of a 3 x 3 / STRIDE-2 / pad-1 convolution (the backbone's downsampling convs, maxvit.py ConvDownsampling_Cf2Cl, and
// the PAFPN bottom-up convs): dx[2a + py][2b + px][c] = sum over the live taps of dy[a + oy][b + ox][:] . w[:][c][ky][kx].
// An input pixel of parity class (py, px) sees 1, 2, 2 or 4 of the nine taps (ky = 1 for even rows, ky in {0, 2} for odd rows, the
// same in x), 2.25 on average; tap (ky, kx) feeds exactly one class, from the dy pixel shifted by (oy, ox) = (ky == 0, kx == 0).
// So this is the forward kernel's loop with FOUR accumulator sets: a workgroup owns RH rows of "anchors" (a, b) = dy pixels, the bf16
// dy halo ((RH + 1) x (Wo + 1) pixels, zero past the image) sits in LDS once, the nine per-tap weight tiles [48 input channels][N]
// stream through a double-buffered LDS tile with three taps of register look-ahead, every MFMA is live work (the implicit GEMM over
// parity classes on gemm_lds_kernel reached 80-107 TFLOP/s: its im2col chunks are rebuilt from global memory for every class).
// KC = N / 16 (contraction: dy channels), 48 input channels (3 column tiles) per workgroup (blockIdx.y), TW anchor row tiles per wave.
// ---------------------------------------------------------------------------------------------------------------------
template <int KC, int TW, int NTO = 3>
__global__ __launch_bounds__(256) void conv3s2_dgrad_kernel(const float* __restrict__ dy, const bf16_t* __restrict__ wp, float* __restrict__ dx,
                                                             int accumulate, int B, int Ho, int Wo, int Cin, int RH) {
    static_assert(KC % 2 == 0, "32-k MFMAs");
    constexpr int N = 16 * KC, LDP = N + 16, LDB = LDP, BN = NTO * 16, LDO = BN + 4;        // NTO: 3 (48 input channels per workgroup) or 4 (64: RVT-B)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int rblocks = (Ho + RH - 1) / RH;
    const int b = blockIdx.x / rblocks, a0 = (blockIdx.x - b * rblocks) * RH;
    const int rows = min(RH, Ho - a0);
    const int P = rows * Wo, ntiles = (P + 15) >> 4;
    const int c0 = blockIdx.y * BN;
    const int WH = Wo + 1;
    const int halo_elems = (RH + 1) * WH * LDP;
    bf16_t* halo = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* sB = halo + ((halo_elems + 7) & ~7);                 // [2][BN * LDB]
    constexpr int BSLOT = BN * (N / 8);                           // 16-byte slots of one tap's weight tile
    constexpr int RB = (BSLOT + 255) / 256;
    typedef int i4 __attribute__((ext_vector_type(4)));
    constexpr int PD = 3;
    i4 rb[PD][RB];
    // packed weights (conv3_pack_kernel mode 1): wp[8 - tap][c][n]
    auto fetch_b = [&](int tap, i4 (&r)[RB]) {
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            const int e = tid + 256 * p;
            if (e < BSLOT) {
                const int c = e / (N / 8), n8 = (e - c * (N / 8)) * 8;
                r[p] = *reinterpret_cast<const i4*>(wp + ((long)(8 - tap) * Cin + c0 + c) * N + n8);
            }
        }
    };
    auto stash_b = [&](int buf, const i4 (&r)[RB]) {
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            const int e = tid + 256 * p;
            if (e < BSLOT) {
                const int c = e / (N / 8), n8 = (e - c * (N / 8)) * 8;
                *reinterpret_cast<i4*>(sB + buf * (BN * LDB) + c * LDB + n8) = r[p];
            }
        }
    };
    fetch_b(0, rb[0]);
    fetch_b(1, rb[1]);
    fetch_b(2, rb[2]);
    // ---- dy halo: rows a0 .. a0 + rows, columns 0 .. Wo, zero past the image ------------------------------------------------------
    const int hslots = (rows + 1) * WH * (N / 4);
    const float* dyb = dy + (long)b * Ho * Wo * N;
    constexpr int HB = 12;
    for (int e0 = tid; e0 < hslots; e0 += 256 * HB) {
        f4 hv[HB]; int ho[HB];
#pragma unroll
        for (int j = 0; j < HB; ++j) {
            const int e = e0 + 256 * j;
            const int hp = e / (N / 4), c4 = (e - hp * (N / 4)) * 4;
            const int hy = hp / WH, hx = hp - hy * WH;
            const int oy = a0 + hy;
            ho[j] = e < hslots ? hp * LDP + c4 : -1;
            hv[j] = zero4();
            if (e < hslots && oy < Ho && hx < Wo) hv[j] = ld4(dyb + ((long)oy * Wo + hx) * N + c4);
        }
#pragma unroll
        for (int j = 0; j < HB; ++j)
            if (ho[j] >= 0) *reinterpret_cast<s4*>(halo + ho[j]) = pack_bf16(hv[j]);
    }
    stash_b(0, rb[0]);
    fetch_b(3, rb[0]);
    // ---- this wave's anchor tiles: lane i of tile t is anchor p = 16 t + i of the region -------------------------------------------
    int hbase[TW]; bool tok[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) {
        const int tile = wave + 4 * t;
        tok[t] = tile < ntiles;
        int p = tile * 16 + i;
        if (p >= P) p = 0;
        const int py = p / Wo, px = p - py * Wo;
        hbase[t] = (py * WH + px) * LDP + 8 * q;
    }
    f4 acc[4][TW][NTO];
#pragma unroll
    for (int cl = 0; cl < 4; ++cl)
#pragma unroll
        for (int t = 0; t < TW; ++t)
#pragma unroll
            for (int n = 0; n < NTO; ++n) acc[cl][t][n] = zero4();
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int buf = tap & 1;
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int cl = (ky == 1 ? 0 : 2) + (kx == 1 ? 0 : 1);                          // parity class (py, px) this tap feeds
        const int toff = ((ky == 0 ? 1 : 0) * WH + (kx == 0 ? 1 : 0)) * LDP;         // dy pixel (a + oy, b + ox)
        const bf16_t* pb = sB + buf * (BN * LDB) + i * LDB + 8 * q;
#pragma unroll
        for (int kc = 0; kc < KC / 2; ++kc) {
            s8v av[TW];
#pragma unroll
            for (int t = 0; t < TW; ++t) av[t] = *reinterpret_cast<const s8v*>(halo + hbase[t] + toff + 32 * kc);
#pragma unroll
            for (int n = 0; n < NTO; ++n) {
                const s8v bv = *reinterpret_cast<const s8v*>(pb + 16 * n * LDB + 32 * kc);
#pragma unroll
                for (int t = 0; t < TW; ++t) acc[cl][t][n] = mfma32_bf16(av[t], bv, acc[cl][t][n]);   // absent tiles compute on anchor 0 (discarded)
            }
        }
        if (tap + 1 < 9) stash_b(buf ^ 1, rb[(tap + 1) % PD]);
        if (tap + 4 < 9) fetch_b(tap + 4, rb[(tap + 1) % PD]);
        __syncthreads();
    }
    // ---- epilogue: class (py, px) of anchor (a, b) is input pixel (2a + py, 2b + px); rows through a wave-private LDS tile -------------
    float* so = reinterpret_cast<float*>(smem_raw) + wave * 16 * LDO;
    const int H = 2 * Ho, W = 2 * Wo;
    float* xb = dx + (long)b * H * W * Cin + c0;
    auto rows_out = [&](auto accc) {                                          // see conv3s1_kernel: no read-modify-write arm inside the row loop
        constexpr bool ACC = decltype(accc)::value;
        constexpr int NK = (16 * NTO * 4 + 63) / 64;
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            if (!tok[t]) continue;                                             // wave-uniform
            const int tile = wave + 4 * t;
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) {
#pragma unroll
                for (int n = 0; n < NTO; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) so[(4 * q + r) * LDO + 16 * n + i] = acc[cl][t][n][r];
                long doff[NK];
                f4 old[ACC ? NK : 1];
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int idx = min(k * 64 + lane, 16 * NTO * 4 - 1);
                    const int row = idx / (NTO * 4), c4 = (idx - row * (NTO * 4)) * 4;
                    const int p = min(tile * 16 + row, P - 1);
                    const int py = p / Wo, px = p - py * Wo;
                    doff[k] = ((long)(2 * (a0 + py) + (cl >> 1)) * W + 2 * px + (cl & 1)) * Cin + c4;
                    if constexpr (ACC) old[k] = ld4(xb + doff[k]);
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int idx = k * 64 + lane;
                    const int row = idx / (NTO * 4), c4 = (idx - row * (NTO * 4)) * 4;
                    if (idx < 16 * NTO * 4 && tile * 16 + row < P) {
                        f4 v = *reinterpret_cast<const f4*>(so + row * LDO + c4);
                        if constexpr (ACC) v += old[k];
                        *reinterpret_cast<f4*>(xb + doff[k]) = v;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        }
    };
    if (accumulate) rows_out(std::true_type{}); else rows_out(std::false_type{});
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of a 3 x 3 / pad-1 convolution of stride S (1: PAFPN / head, 2: the backbone's downsampling convs and the PAFPN
// bottom-up convs): dW[n][c][tap] += sum_p dy[p][n] x[S p + off(tap)][c].
// A workgroup stages the bf16 input halo and the bf16 dy rows of a region (whole output rows of one image, <= 160 pixels) once and
// contracts over the region's pixels for the three taps of one kernel row (blockIdx.y): both MFMA operands are COLUMNS of pixel-major
// LDS tiles, read with ds_read_b64_tr_b16 (the lane supplies the address of "its" pixel row -- for stride 2 simply every other halo
// pixel -- and the hardware transposes 4 pixels x 16 channels per 16-lane group); the dy fragments are shared by the three taps.
// blockIdx.z picks a 16 NA x 16 NB slice of the (n, c) plane (wide layers: 192 -> 384 is 4 x 2 slices of 96 x 96).  Each wave keeps
// a 3 x 3 block of 16 x 16 tiles per tap (27 accumulator tiles): the 4 waves split the slice 2 x 2, or -- 48 input channels -- 2 x 1
// with the pixel steps dealt to the two halves.  The workgroup walks over regions r, r + gridDim.x, ... and writes its partial sums
// with plain stores to part[worker][tap][n][c] (c contiguous); conv3_wgrad_reduce_kernel adds the workers' partials into
// dW[n][c][tap].  No atomics: scattered 4-byte atomics into the [n][c][3][3] layout ran at ~28 per ns (the atomic version of this
// kernel took 200 us, 94 us of it for 2.6 M atomics), and the result is bitwise reproducible.
// The pixel-row stride of the halo tile is chosen so that the 4 pixel rows of a transpose read fall into different 32-byte bank
// groups: S * LDX * 2 bytes is an odd multiple of 32 modulo 256 (LDX = CI + 16 for stride 1, CI + 8 for stride 2).
// ---------------------------------------------------------------------------------------------------------------------
template <int NA, int NB, int S, int HB = 12>                      // HB: 16-byte staging loads in flight per thread
__global__ __launch_bounds__(256) void conv3_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ part,
                                                           int B, int H, int W, int Ho, int Wo, int Ntot, int Ctot, int RH, int nregions, int WSo) {
    static_assert(NA % 2 == 0, "two wave rows over the output channels");
    constexpr int WVB = NB % 2 == 0 ? 2 : 1, WVS = 2 / WVB;               // waves: 2 (n) x WVB (c) x WVS (pixel steps)
    constexpr int N = 16 * NA, CI = 16 * NB, LDX = CI + (S == 1 ? 16 : 8), LDY = N + 16, TA = NA / 2, TB = NB / WVB;
    typedef __attribute__((address_space(3))) s4 lds_s4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int wa = wave & 1, wb = WVB == 2 ? wave >> 1 : 0, ws = WVB == 2 ? 0 : wave >> 1;
    const int g = blockIdx.y;                                     // kernel row: taps 3g .. 3g + 2
    const int nsl = Ntot / N;
    const int n0 = ((int)blockIdx.z % nsl) * N, c0 = ((int)blockIdx.z / nsl) * CI;
    const int WH = S * WSo + 2, csegs = Wo / WSo;              // WSo: output columns of a region (column segments of wide maps, as conv3s1_kernel)
    bf16_t* halo = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* sdy = halo + (((S * (RH - 1) + 3) * WH * LDX + 7) & ~7);
    f4 acc[3][TA][TB];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int b = 0; b < TB; ++b) acc[j][a][b] = zero4();
    const int rblocks = (Ho + RH - 1) / RH;
    for (int reg = blockIdx.x; reg < nregions; reg += gridDim.x) {
        const int seg = reg % csegs, rg = reg / csegs;
        const int b = rg / rblocks, y0 = (rg - b * rblocks) * RH;
        const int x0o = seg * WSo, x0 = S * x0o;
        const int rows = min(RH, Ho - y0);
        const int P = rows * WSo, P32 = (P + 31) & ~31, steps = P32 >> 5;
        // ---- stage the input halo and the dy rows (bf16); all loads of a batch before the first LDS store -----------------------
        // (issuing a region's loads before the previous region's MFMAs -- 28 + 12 staging registers per lane carried across the
        // contraction -- was slower on every shape: 175 -> 189 us stage 2, 169 -> 265 us stage 4)
        const int hslots = (S * (rows - 1) + 3) * WH * (CI / 4);
        const float* xb = x + (long)b * H * W * Ctot + c0;
        for (int e0 = tid; e0 < hslots; e0 += 256 * HB) {
            f4 hv[HB]; int ho[HB];
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                const int e = e0 + 256 * j;
                const int hp = e / (CI / 4), c4 = (e - hp * (CI / 4)) * 4;
                const int hy = hp / WH, hx = hp - hy * WH;
                const int iy = S * y0 - 1 + hy, ix = x0 + hx - 1;
                ho[j] = e < hslots ? hp * LDX + c4 : -1;
                hv[j] = zero4();
                if (e < hslots && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) hv[j] = ld4(xb + ((long)iy * W + ix) * Ctot + c4);
            }
#pragma unroll
            for (int j = 0; j < HB; ++j)
                if (ho[j] >= 0) *reinterpret_cast<s4*>(halo + ho[j]) = pack_bf16(hv[j]);
        }
        const int dslots = P32 * (N / 4);
        const float* dyb = dy + ((long)(b * Ho + y0) * Wo + x0o) * Ntot + n0;
        for (int e0 = tid; e0 < dslots; e0 += 256 * HB) {
            f4 hv[HB];
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                const int e = e0 + 256 * j;
                const int p = e / (N / 4), c4 = (e - p * (N / 4)) * 4;
                hv[j] = (e < dslots && p < P) ? ld4(dyb + (csegs == 1 ? (long)p : (long)(p / WSo) * Wo + p % WSo) * Ntot + c4) : zero4();     // rows P .. P32 - 1: zeros (contribute nothing)
            }
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                const int e = e0 + 256 * j;
                if (e < dslots) { const int p = e / (N / 4), c4 = (e - p * (N / 4)) * 4; *reinterpret_cast<s4*>(sdy + p * LDY + c4) = pack_bf16(hv[j]); }
            }
        }
        __syncthreads();
        // ---- contraction over the region's pixels, 32 per MFMA ----------------------------------------------------------------------
        for (int s = ws; s < steps; s += WVS) {
            // the two pixel rows this lane addresses in a transpose read: p = 32 s + 8 q + (i >> 2) (+ 4)
            int hx0[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int p = 32 * s + 8 * q + 4 * h + (i >> 2);
                if (p >= P) p = P - 1;                                               // dy is zero there: any valid pixel will do
                const int py = p / WSo, px = p - py * WSo;
                hx0[h] = ((S * py + g) * WH + S * px) * LDX + 4 * (i & 3);           // tap (g, 0); taps (g, 1), (g, 2): + LDX, + 2 LDX
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
    float* pw = part + (long)(blockIdx.x * WVS + ws) * 9 * Ntot * Ctot;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int bb = 0; bb < TB; ++bb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + 16 * (TA * wa + a) + 4 * q + r, c = c0 + 16 * (TB * wb + bb) + i;
                    pw[((long)(3 * g + j) * Ntot + n) * Ctot + c] = acc[j][a][bb][r];
                }
}

// Same contraction with all NINE taps in one 8-wave workgroup (conv3_wgrad_kernel is bound by its staging loads: the three kernel-row
// workgroups of a region each fetch the same fp32 halo and dy rows).  Waves 0-3 own taps 0-4, waves 4-7 taps 5-8, each wave the same
// 3 x 3 tile block as above per tap (45 / 36 accumulator tiles); a region's halo and dy rows are fetched and converted ONCE.
// One problem of a weight-gradient launch (blockIdx.z picks it: the tower convs of equal depth over the head levels share a launch, round 6).
struct C3WProb { const float* dy; const float* x; float* part; int B, H, W, Ho, Wo, RH, nregions, WSo, workers; };
struct C3WGroup { C3WProb p[8]; };
template <int NA, int NB, int S, int HB = 8>
__global__ __launch_bounds__(512) void conv3_wgrad9_kernel(C3WGroup probs, int Ntot, int Ctot) {
    const C3WProb& pr = probs.p[blockIdx.z];
    if ((int)blockIdx.x >= pr.workers) return;
    const float* __restrict__ dy = pr.dy; const float* __restrict__ x = pr.x; float* __restrict__ part = pr.part;
    const int B = pr.B, H = pr.H, W = pr.W, Ho = pr.Ho, Wo = pr.Wo, RH = pr.RH, nregions = pr.nregions, WSo = pr.WSo, nworkers = pr.workers;
    (void)B;
    static_assert(NA % 2 == 0, "two wave rows over the output channels");
    constexpr int WVB = NB % 2 == 0 ? 2 : 1, WVS = 2 / WVB;               // waves of a tap group: 2 (n) x WVB (c) x WVS (pixel steps)
    constexpr int N = 16 * NA, CI = 16 * NB, LDX = CI + (S == 1 ? 16 : 8), LDY = N + 16, TA = NA / 2, TB = NB / WVB, NTH = 512, NJ = 5;
    typedef __attribute__((address_space(3))) s4 lds_s4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2), w4 = wave & 3;
    const int wa = w4 & 1, wb = WVB == 2 ? w4 >> 1 : 0, ws = WVB == 2 ? 0 : w4 >> 1;
    const int nsl = Ntot / N;
    const int n0 = ((int)blockIdx.y % nsl) * N, c0 = ((int)blockIdx.y / nsl) * CI;
    const int WH = S * WSo + 2, csegs = Wo / WSo;              // WSo: output columns of a region (column segments of wide maps, as conv3s1_kernel)
    bf16_t* halo = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* sdy = halo + (((S * (RH - 1) + 3) * WH * LDX + 7) & ~7);
    int toff[NJ];                                                 // halo offset of this wave's taps (the group of taps 5-8 repeats tap 8: not stored)
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const int t = min(5 * grp + j, 8); toff[j] = ((t / 3) * WH + (t % 3)) * LDX; }
    f4 acc[NJ][TA][TB];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int b = 0; b < TB; ++b) acc[j][a][b] = zero4();
    const int rblocks = (Ho + RH - 1) / RH;
    for (int reg = blockIdx.x; reg < nregions; reg += nworkers) {
        const int seg = reg % csegs, rg = reg / csegs;
        const int b = rg / rblocks, y0 = (rg - b * rblocks) * RH;
        const int x0o = seg * WSo, x0 = S * x0o;
        const int rows = min(RH, Ho - y0);
        const int P = rows * WSo, P32 = (P + 31) & ~31, steps = P32 >> 5;
        const int hslots = (S * (rows - 1) + 3) * WH * (CI / 4);
        const float* xb = x + (long)b * H * W * Ctot + c0;
        for (int e0 = tid; e0 < hslots; e0 += NTH * HB) {
            f4 hv[HB]; int ho[HB];
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                const int e = e0 + NTH * j;
                const int hp = e / (CI / 4), c4 = (e - hp * (CI / 4)) * 4;
                const int hy = hp / WH, hx = hp - hy * WH;
                const int iy = S * y0 - 1 + hy, ix = x0 + hx - 1;
                ho[j] = e < hslots ? hp * LDX + c4 : -1;
                hv[j] = zero4();
                if (e < hslots && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) hv[j] = ld4(xb + ((long)iy * W + ix) * Ctot + c4);
            }
#pragma unroll
            for (int j = 0; j < HB; ++j)
                if (ho[j] >= 0) *reinterpret_cast<s4*>(halo + ho[j]) = pack_bf16(hv[j]);
        }
        const int dslots = P32 * (N / 4);
        const float* dyb = dy + ((long)(b * Ho + y0) * Wo + x0o) * Ntot + n0;
        for (int e0 = tid; e0 < dslots; e0 += NTH * HB) {
            f4 hv[HB];
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                const int e = e0 + NTH * j;
                const int p = e / (N / 4), c4 = (e - p * (N / 4)) * 4;
                hv[j] = (e < dslots && p < P) ? ld4(dyb + (csegs == 1 ? (long)p : (long)(p / WSo) * Wo + p % WSo) * Ntot + c4) : zero4();     // rows P .. P32 - 1: zeros (contribute nothing)
            }
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                const int e = e0 + NTH * j;
                if (e < dslots) { const int p = e / (N / 4), c4 = (e - p * (N / 4)) * 4; *reinterpret_cast<s4*>(sdy + p * LDY + c4) = pack_bf16(hv[j]); }
            }
        }
        __syncthreads();
        for (int s = ws; s < steps; s += WVS) {
            int hx0[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int p = 32 * s + 8 * q + 4 * h + (i >> 2);
                if (p >= P) p = P - 1;                                               // dy is zero there: any valid pixel will do
                const int py = p / WSo, px = p - py * WSo;
                hx0[h] = (S * py * WH + S * px) * LDX + 4 * (i & 3);                 // tap (0, 0); the others: + toff
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
            for (int j = 0; j < NJ; ++j) {
#pragma unroll
                for (int bb = 0; bb < TB; ++bb) {
                    const int co = toff[j] + 16 * (TB * wb + bb);
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
    float* pw = part + (long)(blockIdx.x * WVS + ws) * 9 * Ntot * Ctot;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int t = 5 * grp + j;
        if (t > 8) continue;
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int bb = 0; bb < TB; ++bb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + 16 * (TA * wa + a) + 4 * q + r, c = c0 + 16 * (TB * wb + bb) + i;
                    pw[((long)t * Ntot + n) * Ctot + c] = acc[j][a][bb][r];
                }
    }
}

// dW[n][c][tap] += sum over workers of part[worker][tap][n][c]   (blockIdx.y: problem of a grouped launch)
struct C3RProb { const float* part; float* dW; int workers; };
struct C3RGroup { C3RProb p[8]; };
__global__ __launch_bounds__(256) void conv3_wgrad_reduce_kernel(C3RGroup grp, int N, int CI) {
    const C3RProb& pr = grp.p[blockIdx.y];
    const float* __restrict__ part = pr.part; float* __restrict__ dW = pr.dW; const int workers = pr.workers;
    const int total = 9 * N * CI;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    float s = 0.f;
#pragma unroll 8
    for (int w = 0; w < workers; ++w) s += part[(long)w * total + e];             // 8 independent loads in flight per thread
    const int c = e % CI, r = e / CI; const int n = r % N, tap = r / N;
    dW[((long)n * CI + c) * 9 + tap] += s;
}

static inline size_t conv3_smem(int RH, int W, int Cin, int nto, int S = 1) {       // W: input width
    const int LDB = Cin + (Cin % 32 == 0 ? 16 : 8), LDP = S == 1 ? LDB : Cin + (Cin % 32 == 0 ? 8 : 4);
    const size_t halo = (((size_t)(S * (RH - 1) + 3) * (W + 2) * LDP + 7) & ~(size_t)7) * 2;
    const size_t sb = (size_t)2 * nto * 16 * LDB * 2;
    const size_t so = (size_t)4 * 16 * (nto * 16 + 4) * 4;
    return max(halo + sb, so);
}
// output rows per workgroup: as many as give <= 160 pixels (10 row tiles) and fit the 160 KB of LDS; 0 = does not fit at all
static inline int conv3_rows_per_block(int H, int W, int Cin, int nto, int S = 1) {       // H, W: input size; rows of the OUTPUT
    const int Ho = H / S, Wo = W / S;
    int rh = max(1, min(Ho, 160 / Wo));
    while (rh > 0 && conv3_smem(rh, W, Cin, nto, S) > 160 * 1024) --rh;
    if (S > 1 && rh > 0) rh = cdiv(Ho, cdiv(Ho, rh));             // equal row blocks
    return rh;
}

// (Cin / 16, column tiles per workgroup, stride) combinations conv3s1_kernel is instantiated for: the 48 / 96 / 192 widths of RVT-T / -S and
// the 64 / 128 / 256 widths of RVT-B
static inline int conv3_nto(int Cin, int Cout, int S) {
    const bool rvt_s = Cin == 48 || Cin == 96 || Cin == 192;
    if (rvt_s) return S == 1 ? (Cout == 48 ? 3 : (Cout == 96 || Cout == 192) ? 6 : 0) : (Cout % 96 == 0 ? 6 : 0);
    if (Cin == 64) return S == 1 ? (Cout == 64 ? 4 : 0) : (Cout % 128 == 0 ? 8 : 0);
    if (Cin == 128) return Cout % 128 == 0 && (S == 2 || Cout == 128) ? 8 : 0;
    if (Cin == 256) return Cout % 64 == 0 && (S == 2 || Cout == 256) ? 4 : 0;
    return 0;
}
// column segments (1, 2 or 4) and output rows per workgroup; false: the halo of even a quarter row does not fit
static inline bool conv3_geometry(int H, int W, int Cin, int nto, int S, int& csegs, int& rh) {
    const int Wo = W / S;
    for (csegs = 1; csegs <= 4; csegs *= 2) {
        if (Wo % csegs || Wo / csegs < 4) return false;
        if (Wo / csegs > 160) continue;
        rh = conv3_rows_per_block(H, W / csegs, Cin, nto, S);
        if (rh > 0) return true;
    }
    return false;
}

// forward of the stride-2 convs on the same kernel (S = 2)
bool conv3s2_fwd_supported(int B, int H, int W, int Cin, int Cout) {
    if (leod_precision() != 1) return false;
    if ((H & 1) || (W & 1)) return false;
    // (192 input channels: the weight tile takes half the LDS, 4 output rows per workgroup -- stage 4 of RVT-S, 13440 output pixels,
    // 141 us vs 125 on the LDS GEMM; but the PAFPN bottom-up conv on 32 frames (2560 pixels) would fall to the register-direct GEMM: 114 us)
    // (same for 128 / 256 input channels of RVT-B at 1 Mpx: 22 frames x 48 x 80 outputs 364 vs 292 us, 22 x 24 x 40: 601 vs 303)
    if ((Cin == 192 || Cin == 128 || Cin == 256) && (long)B * (H / 2) * (W / 2) >= 8192) return false;
    const int nto = conv3_nto(Cin, Cout, 2);
    int csegs, rh;
    return nto && conv3_geometry(H, W, Cin, nto, 2, csegs, rh);
}

bool conv3s1_supported(int H, int W, int Cin, int Cout) {
    if (leod_precision() != 1) return false;
    const int nto = conv3_nto(Cin, Cout, 1);
    int csegs, rh;
    return nto && conv3_geometry(H, W, Cin, nto, 1, csegs, rh);
}

size_t conv3s1_pack_bytes(int Cin, int Cout) { return (size_t)9 * Cin * Cout * sizeof(bf16_t); }

// x [B,H,W,Cin] -> y [B,H,W,Cout].  transposed = 0: y = conv3x3(x, w[Cout][Cin][3][3]); 1: the dgrad of a conv whose weight is
// w[Cin][Cout][3][3] (x = dy).  wpack: scratch of conv3s1_pack_bytes.  n problems of one (Cin, Cout, stride) geometry in one launch.
int conv3s1_group_launch(int n, const float* const* x, const float* const* w, float* const* y, double* const* colstats, const int* stat_rep,
                         const int* accumulate, const int* B, const int* H, const int* W, int Cin, int Cout, int transposed,
                         void* const* wpack, const int* packed, hipStream_t stream, int stride, const BnEval& bne) {
    if (n < 1 || n > 8) return LEOD_ERR_ARG;
    const long total = (long)9 * Cin * Cout;
    const int of = (!transposed && leod_opfmt() == 2) ? 2 : 1;      // forward launches of precision mode 16f: fp16 halo and fp16 packed weights
    const int nto = conv3_nto(Cin, Cout, stride);
    if (!nto) return LEOD_ERR_UNSUPPORTED;
    C3Group g{};
    int tw = 2, maxblocks = 0;
    size_t smem = 0;
    for (int k = 0; k < n; ++k) {
        if (bne.w && (accumulate[k] || colstats[k])) return LEOD_ERR_ARG;
        int csegs = 1, RH = 0;
        if (!conv3_geometry(H[k], W[k], Cin, nto, stride, csegs, RH)) return LEOD_ERR_UNSUPPORTED;
        const int Ho = H[k] / stride, Wo = W[k] / stride, WSo = Wo / csegs;
        bf16_t* wp = reinterpret_cast<bf16_t*>(wpack[k]);
        // the packed layout is [tap][out rows][k]; for the dgrad the stored weight is [N = Cin of this call][C = Cout of this call]
        if (!packed[k])
            hipLaunchKernelGGL(conv3_pack_kernel, dim3((int)min((long)1024, (total + 255) / 256)), dim3(256), 0, stream, w[k], wp,
                               transposed ? Cin : Cout, transposed ? Cout : Cin, transposed ? 1 : 0, of == 2 ? 1 : 0);
        // (Two workgroups per CU for the large launches of the inference passes -- fewer output rows per workgroup, <= 80 KB of LDS each --
        // measured equal: 26.55 vs 26.56 ms per pseudo-label chunk, profiles/r04_a_graph_ab.txt; one workgroup per CU and the smaller halo overlap stay.)
        const int nb = B[k] * cdiv(Ho, RH) * csegs;
        g.p[k] = C3Prob{x[k], wp, y[k], colstats[k], stat_rep[k], accumulate[k], B[k], H[k], W[k], RH, WSo, nb};
        maxblocks = max(maxblocks, nb);
        smem = max(smem, conv3_smem(RH, W[k] / csegs, Cin, nto, stride));
        if (cdiv(cdiv(min(RH, Ho) * WSo, 16), 4) > 2) tw = 3;       // row tiles per wave: the largest any problem needs
    }
    const dim3 grid(maxblocks, Cout / (16 * nto), n);
#define C3_CASE(KCV, NTOV, SV) C3_CASE2(KCV, NTOV, 2, SV, 1) C3_CASE2(KCV, NTOV, 3, SV, 1) C3_CASE2(KCV, NTOV, 2, SV, 2) C3_CASE2(KCV, NTOV, 3, SV, 2)
#define C3_CASE2(KCV, NTOV, TWV, SV, OFV)                                                                                            \
    if (Cin == 16 * KCV && nto == NTOV && tw == TWV && stride == SV && of == OFV) {                                                   \
        static bool attr_set = false;                                                                                                \
        if (!attr_set) {                                                                                                             \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3s1_kernel<KCV, NTOV, TWV, SV, OFV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_set = true;                                                                                                         \
        }                                                                                                                            \
        hipLaunchKernelGGL((conv3s1_kernel<KCV, NTOV, TWV, SV, OFV>), grid, dim3(256), smem, stream, g, Cout, bne);                   \
        return leod_launch_status();                                                                                                 \
    }
    C3_CASE(3, 3, 1) C3_CASE(3, 6, 1) C3_CASE(6, 3, 1) C3_CASE(6, 6, 1) C3_CASE(12, 3, 1) C3_CASE(12, 6, 1)
    C3_CASE(3, 6, 2) C3_CASE(6, 6, 2) C3_CASE(12, 6, 2)
    C3_CASE(4, 4, 1) C3_CASE(8, 8, 1) C3_CASE(16, 4, 1) C3_CASE(4, 8, 2) C3_CASE(8, 8, 2) C3_CASE(16, 4, 2)       // RVT-B
#undef C3_CASE
#undef C3_CASE2
    return LEOD_ERR_UNSUPPORTED;
}

int conv3s1_launch(const float* x, const float* w, float* y, double* colstats, int stat_rep, int accumulate, int B, int H, int W,
                   int Cin, int Cout, int transposed, void* wpack, hipStream_t stream, int stride, int packed,
                   const float* bn_w, const float* bn_b, const float* bn_rm, const float* bn_rv, float bn_eps) {
    const BnEval bne{bn_w, bn_b, bn_rm, bn_rv, bn_eps};
    if (bn_w && (!bn_b || !bn_rm || !bn_rv)) return LEOD_ERR_ARG;
    return conv3s1_group_launch(1, &x, &w, &y, &colstats, &stat_rep, &accumulate, &B, &H, &W, Cin, Cout, transposed, &wpack, &packed, stream,
                                stride, bne);
}

// n independent 3x3 / stride-1 / pad-1 problems of one channel geometry in ONE launch (see C3Group): forward with BatchNorm statistics
// (transposed = 0) or input gradients (transposed = 1: x = dy, y = dx, Cin = the conv's output channels).  LEOD_ERR_UNSUPPORTED: run them singly.
int conv3s1_group(int n, const float* const* x, const float* const* w, float* const* y, double* const* colstats, const int* stat_rep,
                  const int* accumulate, const int* B, const int* H, const int* W, int Cin, int Cout, int transposed, void* const* wpack,
                  const int* packed, hipStream_t stream) {
    const BnEval none{nullptr, nullptr, nullptr, nullptr, 0.f};
    return conv3s1_group_launch(n, x, w, y, colstats, stat_rep, accumulate, B, H, W, Cin, Cout, transposed, wpack, packed, stream, 1, none);
}
bool conv3s1_group_supported(int n, const int* H, const int* W, int Cin, int Cout) {
    if (n < 1 || n > 8) return false;
    for (int k = 0; k < n; ++k)
        if (!conv3s1_supported(H[k], W[k], Cin, Cout)) return false;
    return true;
}

// shapes of the direct weight-gradient kernel: stride 1 or 2 (even H, W), output channels in slices of 96, input channels 48 or in
// slices of 96
// (output channels, input channels) per slice, in units of 16: 96 x 96 / 96 x 48 for the widths of RVT-T / -S, 128 x 64 / 64 x 64 for RVT-B
static inline bool conv3_wgrad_slice(int Cin, int Cout, int stride, int& na, int& nb) {
    if (Cout % 96 == 0 && (Cin % 96 == 0 || (Cin == 48 && stride == 2))) { na = 6; nb = Cin == 48 ? 3 : 6; return true; }
    if (Cout % 128 == 0 && Cin % 64 == 0) { na = 8; nb = 4; return true; }
    if (Cout == 64 && Cin == 64 && stride == 1) { na = 4; nb = 4; return true; }
    return false;
}
struct Conv3WgradPlan { int RH, workers, wvs, nslices, ci, na, csegs; bool nine; size_t smem; };
static inline Conv3WgradPlan conv3_wgrad_plan(int B, int H, int W, int Cin, int Cout, int S, bool force_nine = false);
bool conv3_wgrad_supported(int H, int W, int Cin, int Cout, int stride) {
    if (leod_precision() != 1) return false;
    if (stride != 1 && stride != 2) return false;
    if (stride == 2 && ((H & 1) || (W & 1))) return false;
    int na, nb;
    if (W / stride < 4 || !conv3_wgrad_slice(Cin, Cout, stride, na, nb)) return false;
    return conv3_wgrad_plan(1, H, W, Cin, Cout, stride).RH > 0;
}

// the 8-wave / nine-tap kernel takes everything but the smallest problems (a few regions: the three kernel-row workgroups of
// conv3_wgrad_kernel fill more CUs; 96 -> 96 on the 8 x 10 level: 17 vs 21 us)
static inline bool conv3_wgrad_nine(int nregions, int nslices) {
    static const int on = 1;
    return on && nregions * nslices > 32;
}
static inline Conv3WgradPlan conv3_wgrad_plan(int B, int H, int W, int Cin, int Cout, int S, bool force_nine) {
    Conv3WgradPlan pl{};
    const int Ho = H / S, Wo = W / S;
    int na = 6, nb = 6;
    if (!conv3_wgrad_slice(Cin, Cout, S, na, nb)) return pl;
    pl.na = na; pl.ci = 16 * nb;
    pl.wvs = nb % 2 ? 2 : 1;
    pl.nslices = (Cout / (16 * na)) * (Cin / pl.ci);
    const int LDX = pl.ci + (S == 1 ? 16 : 8), LDY = 16 * na + 16;
    static const int ldskb = 160;
    // column segments: 1 unless the halo of a single output row of full width does not fit (320 input columns, stage 2 of RVT-B at 1 Mpx)
    int RH = 0, WSo = Wo;
    for (pl.csegs = 1; pl.csegs <= 4; pl.csegs *= 2) {
        if (Wo % pl.csegs || Wo / pl.csegs < 4) { pl.csegs = 0; break; }
        WSo = Wo / pl.csegs;
        if (WSo > 160) continue;
        RH = max(1, min(Ho, 160 / WSo));
        while (RH > 0) {
            const size_t halo = (((size_t)(S * (RH - 1) + 3) * (S * WSo + 2) * LDX + 7) & ~(size_t)7) * 2;
            const size_t sdy = (size_t)((RH * WSo + 31) & ~31) * LDY * 2;
            pl.smem = halo + sdy;
            if (pl.smem <= (size_t)ldskb * 1024) break;
            --RH;
        }
        if (RH > 0) break;
    }
    if (RH <= 0 || pl.csegs == 0 || pl.csegs > 4) { pl.RH = 0; pl.workers = 0; return pl; }
    RH = cdiv(Ho, cdiv(Ho, RH));                                  // equal row blocks
    pl.smem = (((size_t)(S * (RH - 1) + 3) * (S * WSo + 2) * LDX + 7) & ~(size_t)7) * 2 + (size_t)((RH * WSo + 31) & ~31) * LDY * 2;
    pl.RH = RH;
    const int nregions = B * cdiv(Ho, RH) * pl.csegs;
    // region workers per (kernel row, slice): each walks over nregions / workers regions.  96 -> 96 / stride 1 (level-0 head conv)
    // measured: 64 -> 42 us, 85 -> 47, 128 -> 51; the sliced / strided shapes fill the chip once (3 * slices * workers ~ 256) with a
    // multiple of 8 workers, so that the three kernel rows of a region (dispatch slots workers apart) share an XCD's L2
    static const int cap = 64;
    static const int fill = 256;
    int workers = cap;
    if (!(S == 1 && Cin == 96 && Cout == 96)) {
        workers = max(1, fill / (3 * pl.nslices));
        if (workers >= 8) workers &= ~7;
    }
    pl.nine = force_nine || conv3_wgrad_nine(nregions, pl.nslices) || pl.na != 6;      // (the three-workgroup kernel exists for the 96-channel slices only)
    if (pl.nine) {
        // 9-tap workgroups, one per CU (LDS): slices * workers of them, and each writes a whole 9 x 96 x CI slice of partial sums that
        // the reduce kernel reads back -- 256 workgroups when each gets >= 2 regions (the backbone convs on 168 frames: stage 2
        // 174 -> 103 us, stage 3 141 -> 69, stage 4 169 -> 74), 128 for the PAFPN / head convs on the 32 labelled frames (38 vs 43 us)
        static const int fill9 = 0;
        const int target = fill9 ? fill9 : (nregions * pl.nslices >= 512 ? 256 : 128);
        workers = max(1, target / pl.nslices);
    }
    pl.workers = min(nregions, workers);
    return pl;
}

// floats of scratch conv3_wgrad_launch needs (the workers' partial sums)
size_t conv3_wgrad_workspace_floats(int B, int H, int W, int Cin, int Cout, int stride) {
    const Conv3WgradPlan pl = conv3_wgrad_plan(B, H, W, Cin, Cout, stride);
    return (size_t)pl.workers * pl.wvs * 9 * Cin * Cout;
}

// n problems of one (Cin, Cout, stride) geometry on the nine-tap kernel in ONE launch + one reduce launch.  force: every problem takes the nine-tap
// plan (a grouped call); otherwise the problems' own plans must all be nine-tap (the single call).
size_t conv3_wgrad_group_workspace_floats(int B, int H, int W, int Cin, int Cout, int stride) {
    const Conv3WgradPlan pl = conv3_wgrad_plan(B, H, W, Cin, Cout, stride, true);
    return (size_t)pl.workers * pl.wvs * 9 * Cin * Cout;
}
int conv3_wgrad_group_launch(int n, const float* const* dy, const float* const* x, float* const* dW, float* const* ws, const int* B, const int* H,
                             const int* W, int Cin, int Cout, int stride, hipStream_t stream, bool force) {
    if (n < 1 || n > 8) return LEOD_ERR_ARG;
    C3WGroup g{};
    C3RGroup rg{};
    int maxw = 0, na = 0, ci = 0, nslices = 0, wvs = 1;
    size_t smem = 0;
    for (int k = 0; k < n; ++k) {
        const Conv3WgradPlan pl = conv3_wgrad_plan(B[k], H[k], W[k], Cin, Cout, stride, force);
        if (pl.workers <= 0 || !pl.nine || !ws[k]) return LEOD_ERR_UNSUPPORTED;
        const int Ho = H[k] / stride, Wo = W[k] / stride;
        g.p[k] = C3WProb{dy[k], x[k], ws[k], B[k], H[k], W[k], Ho, Wo, pl.RH, B[k] * cdiv(Ho, pl.RH) * pl.csegs, Wo / pl.csegs, pl.workers};
        rg.p[k] = C3RProb{ws[k], dW[k], pl.workers * pl.wvs};
        maxw = max(maxw, pl.workers); smem = max(smem, pl.smem);
        na = pl.na; ci = pl.ci; nslices = pl.nslices; wvs = pl.wvs;
    }
    (void)wvs;
    const dim3 grid9(maxw, nslices, n);
#define C3W9_CASE(NAV, NBV, SV)                                                                                                          \
    if (na == NAV && ci == 16 * NBV && stride == SV) {                                                                                   \
        static bool attr_set = false;                                                                                                    \
        if (!attr_set) {                                                                                                                 \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_wgrad9_kernel<NAV, NBV, SV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_set = true;                                                                                                             \
        }                                                                                                                                \
        hipLaunchKernelGGL((conv3_wgrad9_kernel<NAV, NBV, SV>), grid9, dim3(512), smem, stream, g, Cout, Cin);                            \
    } else
    C3W9_CASE(6, 6, 1) C3W9_CASE(6, 6, 2) C3W9_CASE(6, 3, 2) C3W9_CASE(8, 4, 1) C3W9_CASE(8, 4, 2) C3W9_CASE(4, 4, 1) return LEOD_ERR_UNSUPPORTED;
#undef C3W9_CASE
    hipLaunchKernelGGL(conv3_wgrad_reduce_kernel, dim3(cdiv(9 * Cin * Cout, 256), n), dim3(256), 0, stream, rg, Cout, Cin);
    return leod_launch_status();
}

// dW[Cout][Cin][3][3] += wgrad of y = conv3x3(x, stride) for dy [B,Ho,Wo,Cout], x [B,H,W,Cin]; ws: conv3_wgrad_workspace_floats floats
int conv3_wgrad_launch(const float* dy, const float* x, float* dW, float* ws, int B, int H, int W, int Cin, int Cout, int stride, hipStream_t stream) {
    const Conv3WgradPlan pl = conv3_wgrad_plan(B, H, W, Cin, Cout, stride);
    if (pl.workers <= 0 || !ws) return LEOD_ERR_UNSUPPORTED;
    const int Ho = H / stride, Wo = W / stride;
    const int nregions = B * cdiv(Ho, pl.RH) * pl.csegs, WSo = Wo / pl.csegs;
    if (pl.nine) return conv3_wgrad_group_launch(1, &dy, &x, &dW, &ws, &B, &H, &W, Cin, Cout, stride, stream, false);
    const dim3 grid(pl.workers, 3, pl.nslices);
#define C3W_CASE(NBV, SV, HBV)                                                                                                           \
    if (pl.ci == 16 * NBV && stride == SV) {                                                                                             \
        static bool attr_set = false;                                                                                                    \
        if (!attr_set) {                                                                                                                 \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_wgrad_kernel<6, NBV, SV, HBV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_set = true;                                                                                                             \
        }                                                                                                                                \
        hipLaunchKernelGGL((conv3_wgrad_kernel<6, NBV, SV, HBV>), grid, dim3(256), pl.smem, stream, dy, x, ws, B, H, W, Ho, Wo, Cout, Cin, pl.RH, nregions, WSo); \
    } else
    C3W_CASE(6, 1, 12) C3W_CASE(6, 2, 12) C3W_CASE(3, 2, 12) return LEOD_ERR_UNSUPPORTED;      // (18 loads in flight: no faster)
#undef C3W_CASE
    { C3RGroup rg{}; rg.p[0] = C3RProb{ws, dW, pl.workers * pl.wvs}; hipLaunchKernelGGL(conv3_wgrad_reduce_kernel, dim3(cdiv(9 * Cin * Cout, 256), 1), dim3(256), 0, stream, rg, Cout, Cin); }
    return leod_launch_status();
}

// ---- stride-2 dgrad ----------------------------------------------------------------------------------------------------------------
static inline size_t conv3s2_dgrad_smem(int RH, int Wo, int N, int bn = 48) {
    const int LDP = N + 16;
    const size_t halo = (((size_t)(RH + 1) * (Wo + 1) * LDP + 7) & ~(size_t)7) * 2;
    const size_t sb = (size_t)2 * bn * LDP * 2;
    const size_t so = (size_t)4 * 16 * (bn + 4) * 4;
    return max(halo + sb, so);
}
static inline int conv3s2_dgrad_rows(int Ho, int Wo, int N, int bn = 48) {
    int rh = max(1, min(Ho, 160 / Wo));
    while (rh > 0 && conv3s2_dgrad_smem(rh, Wo, N, bn) > 160 * 1024) --rh;
    if (rh > 0) rh = cdiv(Ho, cdiv(Ho, rh));
    return rh;
}
// x [B,H,W,Cin] <- dy [B,H/2,W/2,N]
bool conv3s2_dgrad_supported(int H, int W, int Cin, int N) {
    if (leod_precision() != 1) return false;
    if ((H & 1) || (W & 1) || W / 2 > 160 || W / 2 < 4) return false;
    const bool rvt_s = Cin % 48 == 0 && (N == 96 || N == 192 || N == 384);
    const bool rvt_b = Cin % 64 == 0 && (N == 128 || N == 256);      // (512 dy channels: the weight tile of 64 input channels does not fit next to a halo row)
    if (!rvt_s && !rvt_b) return false;
    return conv3s2_dgrad_rows(H / 2, W / 2, N, rvt_s ? 48 : 64) > 0;
}
// w [N][Cin][3][3]; wpack: conv3s1_pack_bytes(Cin, N) bytes of scratch
int conv3s2_dgrad_launch(const float* dy, const float* w, float* dx, int accumulate, int B, int H, int W, int Cin, int N, void* wpack, hipStream_t stream, int packed) {
    bf16_t* wp = reinterpret_cast<bf16_t*>(wpack);
    const long total = (long)9 * Cin * N;
    if (!packed) hipLaunchKernelGGL(conv3_pack_kernel, dim3((int)min((long)1024, (total + 255) / 256)), dim3(256), 0, stream, w, wp, N, Cin, 1, 0);
    const int Ho = H / 2, Wo = W / 2;
    const int bn = (N == 128 || N == 256) ? 64 : 48;
    const int RH = conv3s2_dgrad_rows(Ho, Wo, N, bn);
    if (RH <= 0) return LEOD_ERR_UNSUPPORTED;
    const dim3 grid(B * cdiv(Ho, RH), Cin / bn);
    const size_t smem = conv3s2_dgrad_smem(RH, Wo, N, bn);
    const int tw = cdiv(cdiv(RH * Wo, 16), 4);
#define C3D_CASE(KCV, TWV) C3D_CASE3(KCV, TWV, 3)
#define C3D_CASE3(KCV, TWV, NTOV)                                                                                                        \
    if (N == 16 * KCV && tw == TWV && bn == 16 * NTOV) {                                                                                                    \
        static bool attr_set = false;                                                                                                    \
        if (!attr_set) {                                                                                                                 \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3s2_dgrad_kernel<KCV, TWV, NTOV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            attr_set = true;                                                                                                             \
        }                                                                                                                                \
        hipLaunchKernelGGL((conv3s2_dgrad_kernel<KCV, TWV, NTOV>), grid, dim3(256), smem, stream, dy, wp, dx, accumulate, B, Ho, Wo, Cin, RH); \
        return leod_launch_status();                                                                                                     \
    }
    C3D_CASE(6, 1) C3D_CASE(6, 2) C3D_CASE(6, 3) C3D_CASE(12, 1) C3D_CASE(12, 2) C3D_CASE(12, 3) C3D_CASE(24, 1) C3D_CASE(24, 2) C3D_CASE(24, 3)
    C3D_CASE3(8, 1, 4) C3D_CASE3(8, 2, 4) C3D_CASE3(8, 3, 4) C3D_CASE3(16, 1, 4) C3D_CASE3(16, 2, 4) C3D_CASE3(16, 3, 4)
#undef C3D_CASE3
#undef C3D_CASE
    return LEOD_ERR_UNSUPPORTED;
}
