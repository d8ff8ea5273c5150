// Stem weight gradient in precision mode bf16 (gfx950): dW[n][c,ky,kx] += sum_pixels dY[pixel][n] * x[c][4 oy + ky - 3][4 ox + kx - 3]
// for the 7x7 / stride 4 / pad 3 stem convolution over raw uint8 voxels (reference: models/layers/maxvit/maxvit.py, the `stem`
// downsample of stage 1: nn.Conv2d(20, 48, 7, 4, 3) under autocast, its weight gradient from autograd).
//
// stem_u8_wgrad_kernel (k_conv.hip) assembles every im2col fragment from single LDS bytes (16 ds_read_u8 + conversions per 12
// MFMAs) and re-reads every input patch four times (one workgroup per quarter of the k range): 994 us, LDS-instruction bound.
// Here
//   * the contraction index is laid out k' = (c*7 + ky)*8 + kx with a dead eighth tap, and the input patch of a 4 x 16 pixel
//     tile sits in LDS as bf16 (uint8 counts are exact in bf16), each row starting at column 4*ox0 - 3: the four taps kx = 4g..4g+3
//     of pixel x are then the 8 aligned bytes at element 4x + 4g -- one ds_read_b64_tr_b16 per lane delivers the MFMA B operand of a
//     16-wide k' tile (two ky rows x 8 taps) for 16 pixels with no conversion and no gather;
//   * dY goes to LDS as bf16 in 16 x 16 blocks and is read with the same transpose loads (A operand = dY^T);
//   * v_mfma_f32_16x16x32_bf16 contracts two output rows (32 pixels) per instruction;
//   * ONE workgroup (8 waves, 9 k' tiles x 3 channel tiles of accumulators per wave) covers the whole 48 x 1120 gradient, so a
//     patch is fetched once; the next tile's patch and dY rows are prefetched into registers under the MFMAs;
//   * persistent grid (one workgroup per CU), one fp32 atomic per dW element and workgroup at the end.
#include "common.hpp"
#include "stem.hpp"
#include <stdlib.h>

namespace {

constexpr int PR = 19;                 // patch rows of a 4-row output tile: 4*3 + 7
constexpr int RD = 17;                 // source dwords per patch row: columns 4*ox0 - 3 .. 4*ox0 + 64
constexpr int RS = 4 * RD;             // bf16 elements per patch row
constexpr int BST = 16 * 16 + 16;      // elements per 16 x 16 dY block (+ pad)
constexpr int KT = 9;                  // k' tiles per wave: 8 waves x 9 x 16 = 1152 >= 20 * 7 * 8
constexpr int RX = 13;                 // patch dwords per thread: 20 * 19 * 17 <= 13 * 512
typedef __attribute__((address_space(3))) s4 lds_s4;
typedef unsigned u4s_ __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(512, 1) void stem_wgrad_bf16_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ x,
                                                                 float* __restrict__ dW, int B, int Cin, int H, int W, int Ho,
                                                                 int Wo, int N, int tiles_x, int tiles_y, int ldn) {
    // blockIdx.y: slice of N output channels out of the ldn of a dY row (64 channels of RVT-B = two slices of 32: four channel tiles of
    // accumulators per wave spill, 41 VGPRs at 2 waves per SIMD)
    dy += blockIdx.y * N;
    dW += (long)blockIdx.y * N * (Cin * 49);
    constexpr int RY = (64 * NT * 4 + 511) / 512;             // dY float4s per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned short* patch = reinterpret_cast<unsigned short*>(smem_raw);
    const int patch_dw = Cin * PR * RD;                        // source dwords = 4-element groups of the patch
    unsigned short* sdy = patch + (((size_t)patch_dw * 4 + 7) & ~(size_t)7);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int ntiles = B * tiles_x * tiles_y;
    const int rows = Cin * 7, K = Cin * 49, W4 = W >> 2, N4 = N >> 2;
    // B operand: element offset of this lane's 4 taps for pixel 4q + (i >> 2) of output row 0, per k' tile
    int boff[KT];
#pragma unroll
    for (int b = 0; b < KT; ++b) {
        int row = 2 * (wave * KT + b) + ((i & 3) >> 1);
        if (row >= rows) row = 0;                              // dead tile: any valid address, its accumulators are dropped
        const int c = row / 7, ky = row - 7 * c;
        boff[b] = (c * PR + ky) * RS + 4 * (4 * q + (i >> 2)) + 4 * (i & 1);
    }
    const int offA = (4 * q + (i >> 2)) * 16 + 4 * (i & 3);
    f4 acc[NT][KT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < KT; ++b) acc[a][b] = zero4();
    // staging registers of the NEXT tile.  One UNALIGNED dword load per 4-element group (columns 4m-3 .. 4m straddle two aligned
    // dwords; gfx950 global loads take any byte address), unconditional on a clamped address so that all 13 are in flight together;
    // the frame edges are patched at stash time from a 2-bit code per group: 0 as loaded, 1 left edge (loaded at column 0: << 24),
    // 2 right edge (loaded at column W-4: >> 8), 3 outside (zero).
    typedef uint32_t u32u __attribute__((aligned(1)));
    uint32_t rx[RX], rmask = 0, ymask = 0; f4 ry[RY];
    auto fetch = [&](int tile) {
        int t = tile;
        const int tx = t % tiles_x; t /= tiles_x;
        const int ty = t % tiles_y; const int bb = t / tiles_y;
        const int iy0 = 16 * ty - 3, m0 = 16 * tx;              // group d of a row covers columns 4 (m0 + d) - 3 .. 4 (m0 + d)
        const uint8_t* xb = x + (long)bb * Cin * H * W;
        rmask = 0; ymask = 0;
#pragma unroll
        for (int p = 0; p < RX; ++p) {
            const int e = min(tid + 512 * p, patch_dw - 1);
            const int cr = e / RD, d = e - cr * RD, c = cr / PR, r = cr - c * PR;
            const int iy = iy0 + r, col = 4 * (m0 + d) - 3;
            const bool rok = (unsigned)iy < (unsigned)H;
            const int cc = min(max(col, 0), W - 4);
            rx[p] = *reinterpret_cast<const u32u*>(xb + (c * H + min(max(iy, 0), H - 1)) * W + cc);
            const uint32_t code = (!rok || col > W - 3 || col < -3) ? 3u : (col < 0 ? 1u : (col > W - 4 ? 2u : 0u));
            rmask |= code << (2 * p);
        }
#pragma unroll
        for (int p = 0; p < RY; ++p) {
            const int e = min(tid + 512 * p, 64 * N4 - 1), px = e / N4, c4 = e - px * N4;   // pixel 0..63 = 16 * row + col
            const int oy = 4 * ty + (px >> 4), ox = 16 * tx + (px & 15);
            ymask |= (uint32_t)(oy < Ho && ox < Wo) << p;
            ry[p] = ld4(dy + (((long)bb * Ho + min(oy, Ho - 1)) * Wo + min(ox, Wo - 1)) * ldn + 4 * c4);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int p = 0; p < RX; ++p) {
            const int e = tid + 512 * p;
            if (e < patch_dw) {
                const uint32_t code = (rmask >> (2 * p)) & 3u, w = rx[p];
                const uint32_t v = code == 0u ? w : (code == 1u ? w << 24 : (code == 2u ? w >> 8 : 0u));
                const f4 f = {(float)(v & 255u), (float)((v >> 8) & 255u), (float)((v >> 16) & 255u), (float)(v >> 24)};
                *reinterpret_cast<s4*>(patch + 4 * e) = pack_bf16(f);
            }
        }
#pragma unroll
        for (int p = 0; p < RY; ++p) {
            const int e = tid + 512 * p, px = e / N4, c4 = e - px * N4;
            if (e < 64 * N4)
                *reinterpret_cast<s4*>(sdy + ((px >> 4) * NT + (c4 >> 2)) * BST + (px & 15) * 16 + 4 * (c4 & 3)) =
                    pack_bf16((ymask >> p) & 1u ? ry[p] : zero4());
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) { fetch(tile); stash(); }
    __syncthreads();
    for (; tile < ntiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        if (next < ntiles) fetch(next);                        // flies under this tile's MFMAs
#pragma unroll
        for (int half = 0; half < 2; ++half) {                 // output rows 2 half, 2 half + 1 of the tile: 32 pixels per MFMA
            s8v av[NT];
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(sdy + ((2 * half) * NT + a) * BST + offA));
                const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(sdy + ((2 * half + 1) * NT + a) * BST + offA));
                av[a] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            const unsigned short* pr = patch + (2 * half) * 4 * RS;
#pragma unroll
            for (int b = 0; b < KT; ++b) {
                const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(pr + boff[b]));
                const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(pr + boff[b] + 4 * RS));
                const s8v bv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int a = 0; a < NT; ++a) acc[a][b] = mfma32_bf16(av[a], bv, acc[a][b]);
            }
        }
        __syncthreads();
        if (next < ntiles) stash();
        __syncthreads();
    }
    // ---- accumulator (row 4q + r = channel n, column i = k' within the tile) -> dW[n][c][ky][kx] ------------------------------------
#pragma unroll
    for (int b = 0; b < KT; ++b) {
        const int row = 2 * (wave * KT + b) + (i >> 3), kx = i & 7;
        if (row >= rows || kx == 7) continue;
        const int c = row / 7, ky = row - 7 * c;
        const int kidx = c * 49 + ky * 7 + kx;
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 16 * a + 4 * q + r;
                if (n < N) atomicAdd(dW + (long)n * K + kidx, acc[a][b][r]);
            }
    }
}

template <int NT>
int launch(const float* dy, const uint8_t* x, float* dW, int B, int Cin, int H, int W, int Ho, int Wo, int N, hipStream_t s, int slices = 1) {
    const int tiles_x = cdiv(Wo, 16), tiles_y = cdiv(Ho, 4);
    const int ntiles = B * tiles_x * tiles_y;
    const size_t patch = (((size_t)Cin * PR * RS + 7) & ~(size_t)7) * 2;
    const size_t lds = patch + (size_t)4 * NT * BST * 2;
    static const int workers = 256;
    const int gx = ntiles < workers ? ntiles : workers;
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_wgrad_bf16_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr = true;
    }
    static const bool reg = (leod_register_input_kernel(reinterpret_cast<const void*>(&stem_wgrad_bf16_kernel<NT>), 1, 13), true);
    (void)reg;
    hipLaunchKernelGGL((stem_wgrad_bf16_kernel<NT>), dim3(gx, slices), dim3(512), lds, s, dy, x, dW, B, Cin, H, W, Ho, Wo, N, tiles_x, tiles_y, N * slices);
    return leod_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------
// Stem FORWARD on the same bf16 patch (round 3; stem_u8_fwd_kernel of k_conv.hip assembles every im2col fragment from single LDS
// bytes through a k -> byte-offset table: 434 us for 81 GFLOP and 410 MB, LDS-instruction bound):
//   y[pixel][n] = sum_k' W'[n][k'] * P[k'][pixel],  k' = (c*7 + ky)*8 + kx  (dead eighth tap: zero weight)
//   * A operand = W' (bf16, all N rows resident in LDS for the life of the persistent workgroup: rows of KS*32 + 16 elements, a
//     stride == 8 mod 16 dwords, so a fragment is ONE conflict-free ds_read_b128 of 8 consecutive k');
//   * B operand = the patch: the 8 taps of (c, ky) for pixel x are the 16 bytes at element 4x of patch row (c, 4*row + ky) -- two
//     ds_read_b64, no conversion, no gather (uint8 counts are exact in bf16);
//   * accumulator rows = channels: a lane holds 4 consecutive channels of one pixel -> 16-byte stores straight from the registers;
//   * waves = 4 output rows x 2 channel groups (n-tiles [0, NT0) and [NT0, NT): a SIMD hosts one wave of each, so the MFMA load
//     per SIMD is even although 3 tiles do not split in two);
//   * PD tiles of raw uint8 patch dwords are in flight in registers (13 dwords per thread and tile): one workgroup per CU cannot hide
//     an HBM round trip behind ~1 us of MFMAs with a single tile in flight.
// ---------------------------------------------------------------------------------------------------------------------
// OF: operand format, 1 = bf16, 2 = fp16 (precision mode 16f; the uint8 counts are exact in both, the weights round to 11 bits)
template <int NT, int CIN, int PD, int OF = 1>
__global__ __launch_bounds__(512, 1) void stem_fwd_bf16_kernel(const uint8_t* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                               int B, int H, int W, int Ho, int Wo, int N, int tiles_x, int tiles_y, int dbg) {
    constexpr int NT0 = (NT + 1) / 2;                          // n-tiles of channel group 0
    constexpr int rows = CIN * 7, KS = (rows + 3) >> 2, LDW = KS * 32 + 16, K = CIN * 49;
    constexpr int patch_dw = CIN * PR * RD, RXN = (patch_dw + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned short* sw = reinterpret_cast<unsigned short*>(smem_raw);            // [16 NT][LDW]
    unsigned short* patch = sw + 16 * NT * LDW;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, q = lane >> 4;
    const int pr = wave & 3, ns = wave >> 2;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (round-robin dispatch); it takes frames == b % 8 (mod 8) and, with the other
    // gs / 8 workgroups of its XCD, walks their tiles row-major -- x- / y-adjacent tiles (64 of a tile's 68 input columns share their
    // 128-byte lines with a neighbour, 3 of 19 rows) are fetched through ONE L2 at about the same time instead of through two
    const int tpf = tiles_x * tiles_y, gs = gridDim.x;
    const int nx = gs >= 8 ? 8 : 1, xcd = blockIdx.x % nx, slots = gs / nx;       // (gs is a multiple of 8 or < 8, see the launcher)
    const int umax = ((B + nx - 1) / nx) * tpf;                                    // work items per XCD (frames past B are skipped)
    auto decode = [&](int u, int& bb, int& ty, int& tx) -> bool {
        const int f = u / tpf, r = u - f * tpf;
        bb = xcd + nx * f; ty = r / tiles_x; tx = r - ty * tiles_x;
        return u < umax && bb < B;
    };
    typedef uint32_t u32u __attribute__((aligned(1)));
    // staging slot p of this thread: source group e = tid + 512 p = (channel c, patch row r, dword dd); its offset inside an interior
    // tile is a per-thread constant, the frame edges take the clamped path with a 2-bit patch code per group (see the weight gradient)
    int eoff[RXN];
#pragma unroll
    for (int p = 0; p < RXN; ++p) {
        const int e = min(tid + 512 * p, patch_dw - 1);
        const int cr = e / RD, dd = e - cr * RD, c = cr / PR, r = cr - c * PR;
        eoff[p] = (c * H + r) * W + 4 * dd - 3;
    }
    uint32_t rx[PD][RXN], rmask[PD];
    auto fetch = [&](int d, int u) {
        int bb, ty, tx;
        if (!decode(u, bb, ty, tx)) return;
        const int iy0 = 16 * ty - 3, m0 = 16 * tx;
        const uint8_t* xb = x + (long)bb * CIN * H * W;
        if (iy0 >= 0 && iy0 + PR <= H && m0 > 0 && 4 * (m0 + RD - 1) <= W - 4) {     // interior tile (workgroup-uniform)
            const uint8_t* tb = xb + iy0 * W + 4 * m0;
#pragma unroll
            for (int p = 0; p < RXN; ++p) rx[d][p] = *reinterpret_cast<const u32u*>(tb + eoff[p]);
            rmask[d] = 0;
            return;
        }
        uint32_t m = 0;
#pragma unroll
        for (int p = 0; p < RXN; ++p) {
            const int e = min(tid + 512 * p, patch_dw - 1);
            const int cr = e / RD, dd = e - cr * RD, c = cr / PR, r = cr - c * PR;
            const int iy = iy0 + r, col = 4 * (m0 + dd) - 3;
            const bool rok = (unsigned)iy < (unsigned)H;
            const int cc = min(max(col, 0), W - 4);
            rx[d][p] = *reinterpret_cast<const u32u*>(xb + (c * H + min(max(iy, 0), H - 1)) * W + cc);
            const uint32_t code = (!rok || col > W - 3 || col < -3) ? 3u : (col < 0 ? 1u : (col > W - 4 ? 2u : 0u));
            m |= code << (2 * p);
        }
        rmask[d] = m;
    };
    auto stash = [&](int d) {
        if (dbg & 2) return;
        const bool plain = __builtin_amdgcn_ballot_w64(rmask[d] != 0u) == 0ull;      // wave-uniform: no edge group in this wave
#pragma unroll
        for (int p = 0; p < RXN; ++p) {
            const int e = tid + 512 * p;
            if (p < RXN - 1 || e < patch_dw) {
                uint32_t v = rx[d][p];
                if (!plain) {
                    const uint32_t code = (rmask[d] >> (2 * p)) & 3u;
                    v = code == 0u ? v : (code == 1u ? v << 24 : (code == 2u ? v >> 8 : 0u));
                }
                const f4 f = {(float)(v & 255u), (float)((v >> 8) & 255u), (float)((v >> 16) & 255u), (float)(v >> 24)};
                *reinterpret_cast<s4*>(patch + 4 * e) = pack16_raw<OF>(f);
            }
        }
    };
    int tile = blockIdx.x / nx;                               // work index u of this workgroup: slot + slots * j
#pragma unroll
    for (int d = 0; d < PD; ++d) fetch(d, tile + d * slots);
    // ---- weights -> LDS (once per workgroup; under the first patches' loads): sw[n][(c*7 + ky)*8 + kx], dead tap and padding zero ----
    for (int e = tid; e < 16 * NT * (LDW / 8); e += 512) {
        const int n = e / (LDW / 8), row = e - n * (LDW / 8);
        f4 lo = zero4(), hi = zero4();
        if (n < N && row < rows) {
            const float* wp = w + (long)n * K + row * 7;
            lo = f4{wp[0], wp[1], wp[2], wp[3]};
            hi = f4{wp[4], wp[5], wp[6], 0.f};
        }
        *reinterpret_cast<s4*>(sw + n * LDW + 8 * row) = pack16_raw<OF>(lo);
        *reinterpret_cast<s4*>(sw + n * LDW + 8 * row + 4) = pack16_raw<OF>(hi);
    }
    if (tile < umax) stash(0);
    fetch(0, tile + PD * slots);
    __syncthreads();
    const unsigned short* pa0 = sw + ((ns ? 16 * NT0 : 0) + i) * LDW + 8 * q;
    const unsigned short* pb0 = patch + (4 * pr) * RS + 4 * i;
    int roff[KS];                                              // patch row (c, ky) of k-step s for lane group q: row = 4 s + q
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) {
        const int row = 4 * s2 + q;
        roff[s2] = row < rows ? ((row / 7) * PR + row % 7) * RS : 0;
    }
    // one tile: MFMAs from the LDS patch, stores, then the next tile's registers (set dn) go to LDS and take the loads of tile + (PD+1) gs
    auto do_tile = [&](int dn) -> bool {
        f4 acc[NT0];
#pragma unroll
        for (int t = 0; t < NT0; ++t) acc[t] = zero4();
        if (!(dbg & 1)) {
            if (ns == 0) {
#pragma unroll
                for (int s2 = 0; s2 < KS; ++s2) {
                    const u2_ b0 = *reinterpret_cast<const u2_*>(pb0 + roff[s2]);
                    const u2_ b1 = *reinterpret_cast<const u2_*>(pb0 + roff[s2] + 4);
                    const u4s_ bq = {b0.x, b0.y, b1.x, b1.y};
                    const s8v bv = __builtin_bit_cast(s8v, bq);
#pragma unroll
                    for (int t = 0; t < NT0; ++t) acc[t] = mfma32_16<OF>(*reinterpret_cast<const s8v*>(pa0 + 16 * t * LDW + 32 * s2), bv, acc[t]);
                    if (s2 % 6 == 5) __builtin_amdgcn_sched_barrier(0);       // bound how far the scheduler hoists fragment reads (registers)
                }
            } else {
#pragma unroll
                for (int s2 = 0; s2 < KS; ++s2) {
                    const u2_ b0 = *reinterpret_cast<const u2_*>(pb0 + roff[s2]);
                    const u2_ b1 = *reinterpret_cast<const u2_*>(pb0 + roff[s2] + 4);
                    const u4s_ bq = {b0.x, b0.y, b1.x, b1.y};
                    const s8v bv = __builtin_bit_cast(s8v, bq);
#pragma unroll
                    for (int t = 0; t < NT - NT0; ++t) acc[t] = mfma32_16<OF>(*reinterpret_cast<const s8v*>(pa0 + 16 * t * LDW + 32 * s2), bv, acc[t]);
                    if (s2 % 6 == 5) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        int bb, ty, tx;
        const bool live = decode(tile, bb, ty, tx);
        const int oy = 4 * ty + pr, ox = 16 * tx + i;
        if (live && oy < Ho && ox < Wo) {
            float* yp = y + (((long)bb * Ho + oy) * Wo + ox) * N + (ns ? 16 * NT0 : 0) + 4 * q;
#pragma unroll
            for (int t2 = 0; t2 < NT0; ++t2)
                if ((ns == 0 || t2 < NT - NT0) && (ns ? 16 * NT0 : 0) + 16 * t2 + 4 * q < N) *reinterpret_cast<f4*>(yp + 16 * t2) = acc[t2];
        }
        tile += slots;
        if (tile >= umax) return false;
        __syncthreads();                                       // every wave is done with the patch
        stash(dn);
        if (!(dbg & 4)) fetch(dn, tile + PD * slots);
        __syncthreads();
        return true;
    };
    if (tile >= umax) return;
    while (true) {
#pragma unroll
        for (int d = 0; d < PD; ++d)
            if (!do_tile((d + 1) % PD)) return;
    }
}

template <int NT, int CIN>
int launch_fwd(const uint8_t* x, const float* w, float* y, int B, int H, int W, int Ho, int Wo, int N, hipStream_t s) {
    constexpr int PD = 3;
    const int tiles_x = cdiv(Wo, 16), tiles_y = cdiv(Ho, 4);
    const int ntiles = B * tiles_x * tiles_y;
    constexpr int KS = (CIN * 7 + 3) >> 2, LDW = KS * 32 + 16;
    const size_t lds = (size_t)16 * NT * LDW * 2 + (size_t)CIN * PR * RS * 2;
    static const int workers = 256;
    static const int dbg = 0;
    int gx = ntiles < workers ? ntiles : workers;
    if (gx >= 8) gx &= ~7;                                     // whole XCD rounds (see the tile order of the kernel)
    LEOD_BY_OPFMT16({
        static bool attr = false;
        if (!attr) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_fwd_bf16_kernel<NT, CIN, PD, OF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            leod_register_input_kernel(reinterpret_cast<const void*>(&stem_fwd_bf16_kernel<NT, CIN, PD, OF>), 0, 12);
            attr = true;
        }
        hipLaunchKernelGGL((stem_fwd_bf16_kernel<NT, CIN, PD, OF>), dim3(gx), dim3(512), lds, s, x, w, y, B, H, W, Ho, Wo, N, tiles_x, tiles_y, dbg);
    });
    return leod_launch_status();
}

}  // namespace

bool stem_wgrad_bf16_supported(const void* x, int Cin, int H, int W, int N, int stride, int pad) {
    static const int on = 1;
    return on && stride == 4 && pad == 3 && N >= 16 && N <= 64 && !(N & 15) && !(W & 3) && Cin * PR * RD <= RX * 512 &&
           Cin * 7 <= 2 * 8 * KT && ((uintptr_t)x & 3) == 0 && ((long)Cin * H * W) % 4 == 0;
}

int stem_wgrad_bf16_launch(const float* dy, const void* x, float* dW, int B, int Cin, int H, int W, int Ho, int Wo, int N,
                           hipStream_t s) {
    switch (N / 16) {
        case 1: return launch<1>(dy, (const uint8_t*)x, dW, B, Cin, H, W, Ho, Wo, N, s);
        case 2: return launch<2>(dy, (const uint8_t*)x, dW, B, Cin, H, W, Ho, Wo, N, s);
        case 3: return launch<3>(dy, (const uint8_t*)x, dW, B, Cin, H, W, Ho, Wo, N, s);
        default: return launch<2>(dy, (const uint8_t*)x, dW, B, Cin, H, W, Ho, Wo, 32, s, 2);     // 64 channels: two slices of 32
    }
}

// forward: the 20 event-representation channels of every RVT configuration; weights of all N channels (16 NT x 35 x 32 + pad bf16) and
// the patch fill the 160 KB of LDS (N = 64, RVT-B, does not fit: it stays on stem_u8_fwd_kernel)
bool stem_fwd_bf16_supported(const void* x, int Cin, int H, int W, int N, int stride, int pad) {
    static const int on = 1;
    return on && stride == 4 && pad == 3 && Cin == 20 && (N == 32 || N == 48) && !(W & 3) && ((uintptr_t)x & 3) == 0 &&
           ((long)Cin * H * W) % 4 == 0;
}

int stem_fwd_bf16_launch(const void* x, const float* w, float* y, int B, int Cin, int H, int W, int Ho, int Wo, int N, hipStream_t s) {
    (void)Cin;
    if (N == 32) return launch_fwd<2, 20>((const uint8_t*)x, w, y, B, H, W, Ho, Wo, N, s);
    return launch_fwd<3, 20>((const uint8_t*)x, w, y, B, H, W, Ho, Wo, N, s);
}
