// Wide-tile bf16 GEMM of precision mode bf16 (round 3):  Out[m][n] = sum_k A(m,k) * B(n,k)  for the mid-size row contractions of
// RVT stages 2-4 (13 k - 215 k rows, K and N in 96 .. 1536), forward and dgrad -- the shapes gemm_lds_kernel<.., BF = true> served.
//
// What bound the 64 x 64 workgroups of gemm_lds_kernel on these shapes (profiles/r02_i_kbench_per_op_bf16.txt: 100-120 TFLOP/s,
// 0.9-2.3 TB/s): every workgroup pulls its A tile AND its weight tile through the CU's vector cache (64 B/clk per CU) once per 48-k
// chunk -- 16 FLOP per byte of L1 traffic, and in LayerNorm mode the loader fetched the LayerNorm weight / bias again for every
// staging slot (two thirds of the A-side requests) -- with ONE 12-MFMA chunk between two barriers, on the 16-k bf16 MFMA that
// issues at half rate.  Here:
//   * 128 x (64 * NTW) output tile per 8-wave workgroup (NTW = 3 / 4: 192 / 256 columns), waves in a 2 x 4 grid, each owning
//     64 rows x 16*NTW columns = 4 x NTW accumulator tiles: 38-44 FLOP per L1 byte; A is fetched N / 192 times instead of N / 64;
//   * 64-k chunks = two v_mfma_f32_16x16x32_bf16 steps (24-32 MFMAs per wave) per barrier pair; fragments are single ds_read_b128
//     (rows of KCH + 16 bf16: stride == 8 mod 16 dwords, the conflict-free stride of the 16-lane service groups);
//   * transposed weights (dgrad: W[k][n], n contiguous) stay in natural [16 k][16 n] blocks and are read with two
//     ds_read_b64_tr_b16 per fragment (k = 4q..4q+3 and 16+4q..+3); the A tile is stored with the SAME k permutation inside each
//     32-k group, so its fragments stay single 16-byte reads;
//   * every staging slot of a thread has the same k offset (512 threads = 32 rows x 16 k-slots): LayerNorm weight / bias and the
//     per-k scale are loaded ONCE per chunk and thread;
//   * ONE persistent workgroup per CU walks a contiguous range of tiles: double-buffered LDS tiles with ONE barrier per chunk --
//     MFMAs of chunk c, then the registers (loads of chunk c + 1, issued an iteration earlier) go to the other buffer, then the loads
//     of chunk c + 2 are issued -- and the chunk stream runs across tile boundaries, so the epilogue of a tile (row-layout epilogues
//     of gemm16.hpp, EpStore / EpLsRes::run_rows, through wave-private 16 x 64 tiles) overlaps the loads of the next tile.
#pragma once

// WN = 4, KCH = 64: the 8-wave workgroup described above (one per CU).  WN = 2, KCH = 32 (NTW = 3 only): a 4-wave workgroup with the same
// 128 x 192 tile -- waves 2 x 2, each 64 rows x 96 columns = 4 x 6 accumulator tiles, epilogue in two 48-column halves -- on 76 KB of LDS,
// so two workgroups can share a CU (kept as a template option; see launch_gemm_wide for what it measured).
// OF: operand format, 1 = bf16, 2 = fp16 (forward launches of precision mode 16f)
template <int NTW, class AL, class BL, class EP, int OF = 1, int WN = 4, int KCH = 64>
__global__ __launch_bounds__(128 * WN, 2) void gemm_wide_bf16_kernel(AL al, BL bl, EP ep, int M, int K, int nblocks_n, int dbg) {
    constexpr int K4 = KCH / 4, BM = 128, BN = 64 * NTW, LD = KCH + 16;
    constexpr int NTHR = 128 * WN, NWAVE = 2 * WN, RS = NTHR / K4;   // threads, waves, rows per staging pass
    constexpr int TW = 4 * NTW / WN, NH = TW / NTW;                  // accumulator column tiles per wave; epilogue halves of NTW tiles
    static_assert(BM % RS == 0 && TW * WN == 4 * NTW && NH * NTW == TW, "wave grid");
    constexpr int NTB = BN / 16;                     // 16-column blocks of the B tile
    constexpr int BST = 16 * 16 + 16;                // bf16 elements per [16 k][16 n] block of transposed weights (+ 32 bytes)
    constexpr int RA = BM / RS;                      // = 4 staging slots per thread (rows r0 + RS p)
    constexpr int RB = BL::kTrans ? KCH * (BN / 4) / NTHR : BN * K4 / NTHR;
    constexpr int ASZ = BM * LD;                     // bf16 elements
    constexpr int BSZ = BL::kTrans ? (KCH / 16) * NTB * BST : BN * LD;
    constexpr int LDO = 64;
    static_assert(std::is_base_of<BLRows, BL>::value || std::is_base_of<BLTrans, BL>::value, "weights as W[n][k] or W[k][n]");
    __shared__ __attribute__((aligned(16))) unsigned short sOp[2][ASZ + BSZ];       // double-buffered operand tiles
    __shared__ __attribute__((aligned(16))) float sOut[NWAVE][16 * LDO];                // wave-private transposition tiles of the epilogue
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
    // ---- persistent workgroup: a contiguous range of (row block, n-block) tiles, n-blocks of a row block consecutive (the A rows
    // are re-read from this CU's L1 / this XCD's L2) ----------------------------------------------------------------------------------
    const int nch = (K + KCH - 1) / KCH;
    const long ntiles = (long)((M + BM - 1) / BM) * nblocks_n;
    const int tile_begin = (int)(ntiles * blockIdx.x / gridDim.x), tile_end = (int)(ntiles * (blockIdx.x + 1) / gridDim.x);
    if (tile_begin >= tile_end) return;
    // ---- staging slots: thread = (row r0 = tid / 16, k offset ak = 4 * (tid % 16)); slot p adds 32 rows -------------------------------
    const int ak = (tid % K4) * 4, r0 = tid / K4;
    // A tile position of k offset ak inside its 32-k group: natural for row-major weights, (k = 4g + j -> 8 (g % 4) + 4 (g / 4) + j)
    // for transposed weights (the order the two transpose reads of a B fragment deliver)
    const int g8 = (ak >> 2) & 7;
    const int apos = BL::kTrans ? (ak & 32) + 8 * (g8 & 3) + 4 * (g8 >> 2) : ak;
    const int a_off = r0 * LD + apos;
    int bn[RB], bk[RB], b_off[RB];
#pragma unroll
    for (int p = 0; p < RB; ++p) {
        const int e = tid + NTHR * p;
        if constexpr (!BL::kTrans) { bn[p] = e / K4; bk[p] = (e - bn[p] * K4) * 4; b_off[p] = bn[p] * LD + bk[p]; }
        else {
            const int kl = e / (BN / 4), n4 = (e - kl * (BN / 4)) * 4;
            bn[p] = n4; bk[p] = kl;
            b_off[p] = ((kl >> 4) * NTB + (n4 >> 4)) * BST + (kl & 15) * 16 + (n4 & 15);
        }
    }
    constexpr bool ATP = a_two_phase<AL>::value;
    static_assert(ATP, "gemm_wide_bf16_kernel stages A through the two-phase row loaders (ALRowsM)");
    // fetch side: the chunk whose loads are (about to be) in flight
    int tile_f = tile_begin, ch_f = 0, brow_f = 0, ncol_f = 0;
    typename AL::St ast[RA];
    auto set_fetch_tile = [&]() {
        const int rbk = tile_f / nblocks_n;
        brow_f = rbk * BM; ncol_f = (tile_f - rbk * nblocks_n) * BN;
#pragma unroll
        for (int p = 0; p < RA; ++p) ast[p] = al.init((dbg & 32) ? r0 + RS * p : brow_f + r0 + RS * p, M, 0, false);
    };
    f4 ra[RA], rb[RB];
    typename a_two_phase<AL>::Raw raw_a[RA];
    auto fetch = [&]() {                                       // issue the loads of chunk (tile_f, ch_f): unconditional, clamped
        const int k0 = ch_f * KCH;
        const int ka = min(k0 + ak, K - 4);
#pragma unroll
        for (int p = 0; p < RA; ++p) al.raw_x(ast[p], ka, raw_a[p]);
        al.raw_k(ka, raw_a[0]);                                // LayerNorm weight / bias, per-k scale: once per chunk and thread
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            const long off = !BL::kTrans ? (long)min(ncol_f + bn[p], bl.N - 1) * bl.ld + min(k0 + bk[p], K - 4)
                                         : (long)min(k0 + bk[p], K - 1) * bl.ld + min(ncol_f + bn[p], bl.N - 4);
            if constexpr (bl_is16<BL>::value) rb[p] = shadow_bits(*reinterpret_cast<const u2_*>(bl.w16 + off));    // bf16 shadow: 8-byte loads, no conversion
            else rb[p] = ld4(bl.w + off);
        }
    };
    auto stash = [&](int buf) {                                // raw registers -> operand values -> bf16 tiles of buffer `buf`
        if (dbg & 8) return;
        const int k0 = ch_f * KCH;
        unsigned short* sA = sOp[buf];
        unsigned short* sB = sA + ASZ;
        const bool kok = k0 + ak < K;
#pragma unroll
        for (int p = 0; p < RA; ++p) {
            const f4 v = al.fin_k(ast[p], raw_a[p], raw_a[0]);
            *reinterpret_cast<s4*>(sA + a_off + RS * p * LD) = pack16<OF>((ast[p].ok && kok) ? v : zero4());
        }
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            const bool ok = (ncol_f + bn[p] < bl.N) && (k0 + bk[p] < K);
            if constexpr (bl_is16<BL>::value) *reinterpret_cast<s4*>(sB + b_off[p]) = ok ? shadow_s4(rb[p]) : s4{0, 0, 0, 0};
            else *reinterpret_cast<s4*>(sB + b_off[p]) = pack16_raw<OF>(ok ? rb[p] : zero4());
        }
    };
    auto advance_fetch = [&]() -> bool {                       // next chunk of this workgroup's range; false: none left
        if (++ch_f == nch) {
            ch_f = 0;
            if (++tile_f >= tile_end) return false;
            set_fetch_tile();
        }
        return true;
    };
    f4 acc[4][TW];
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int t = 0; t < TW; ++t) acc[w][t] = zero4();
    set_fetch_tile();
    fetch();
    stash(0);
    bool pending = advance_fetch();
    if (pending) fetch();
    __syncthreads();
    const int pa_off = (64 * wm + i) * LD + 8 * q;
    const int pb_off = ASZ + (BL::kTrans ? (TW * wn) * BST + (4 * q + (i >> 2)) * 16 + 4 * (i & 3) : (16 * TW * wn + i) * LD + 8 * q);
    typedef __attribute__((address_space(3))) s4 lds_s4;
    float* so = sOut[wave];
    BL blw = bl;
    blw.NT = NTW;                                             // column mapping of the epilogue: virtual n-block = 4 nblk + wn
    int cur = 0, ch_c = 0, tile_c = tile_begin;
    while (true) {
        // ---- MFMAs of the current chunk (buffer cur) ------------------------------------------------------------------------------
        const unsigned short* pa = sOp[cur] + pa_off;
        const unsigned short* pb = sOp[cur] + pb_off;
#pragma unroll
        for (int c = 0; c < KCH / 32; ++c) {
            if (dbg & 2) break;
            s8v av[4], bv[TW];
#pragma unroll
            for (int w = 0; w < 4; ++w) av[w] = *reinterpret_cast<const s8v*>(pa + 16 * w * LD + 32 * c);
#pragma unroll
            for (int t = 0; t < TW; ++t) {
                if constexpr (BL::kTrans) {
                    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(pb + ((2 * c) * NTB + t) * BST));
                    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(pb + ((2 * c + 1) * NTB + t) * BST));
                    bv[t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                } else {
                    bv[t] = *reinterpret_cast<const s8v*>(pb + 16 * t * LD + 32 * c);
                }
            }
#pragma unroll
            for (int t = 0; t < TW; ++t)
#pragma unroll
                for (int w = 0; w < 4; ++w) acc[w][t] = mfma32_16<OF>(av[w], bv[t], acc[w][t]);
        }
        // ---- last chunk of a tile: epilogue, one 16 x (16 NTW) accumulator fragment at a time through the wave-private tile; the
        // loads of the next tile's first chunk are already in flight ---------------------------------------------------------------
        const bool last = ++ch_c == nch;
        if (last) {
            const int rbk = tile_c / nblocks_n;
            const int brow_c = rbk * BM, nblk_c = tile_c - rbk * nblocks_n;
            // one body per tile (see gemm_lds_kernel): mode 0 = generic row epilogue, modes >= 1 = the epilogue's branch-free bodies
            auto ep_tile = [&](auto modec) {
                constexpr int MODE = decltype(modec)::value;
                // fragments f = (w, h): 16 rows x (16 NTW) columns; virtual n-block (of 16 NTW columns) = 4 nblk + NH wn + h
                const int vb0 = nblk_c * 4 + NH * wn;
                RowPre pre;
                if constexpr (MODE > 0) pre = ep.template prefetch_full<MODE, NTW, BL>(blw, brow_c + 64 * wm, vb0, lane);
                else if constexpr (EP::kPrefetchRows) pre = ep.template prefetch_rows<NTW, BL>(blw, brow_c + 64 * wm, vb0, lane, M);
                // (one call per fragment with a compile-time index: the eight copies of the generic body exceed the unroller's size limit,
                // and a loop that stays rolled indexes acc dynamically -> the accumulators would live in scratch memory)
                auto frag = [&](auto fc) {
                    constexpr int f = decltype(fc)::value;
                    constexpr int w = f / NH, h = f % NH;
                    const int row0 = brow_c + 64 * wm + 16 * w;
                    const RowPre cur = pre;
                    if (f + 1 < 4 * NH) {
                        const int rown = brow_c + 64 * wm + 16 * ((f + 1) / NH), vbn = vb0 + (f + 1) % NH;
                        if constexpr (MODE > 0) pre = ep.template prefetch_full<MODE, NTW, BL>(blw, rown, vbn, lane);
                        else if constexpr (EP::kPrefetchRows) pre = ep.template prefetch_rows<NTW, BL>(blw, rown, vbn, lane, M);
                    }
                    // LDS operations of one wave execute in order: only the COMPILER has to keep write -> read -> write order
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int t = 0; t < NTW; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) so[(4 * q + r) * LDO + 16 * t + i] = acc[w][h * NTW + t][r];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    if (!(dbg & 1)) {
                        if constexpr (MODE > 0) ep.template run_rows_full<MODE, NTW, BL>(so, LDO, blw, row0, vb0 + h, lane, cur);
                        else if (row0 < M) {
                            if constexpr (EP::kPrefetchRows) ep.template run_rows<NTW, BL>(so, LDO, blw, row0, vb0 + h, lane, M, cur);
                            else ep.template run_rows<NTW, BL>(so, LDO, blw, row0, vb0 + h, lane, M);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < NTW; ++t) acc[w][h * NTW + t] = zero4();
                };
                frag(std::integral_constant<int, 0>{}); frag(std::integral_constant<int, 1>{});
                frag(std::integral_constant<int, 2>{}); frag(std::integral_constant<int, 3>{});
                if constexpr (NH == 2) {
                    frag(std::integral_constant<int, 4>{}); frag(std::integral_constant<int, 5>{});
                    frag(std::integral_constant<int, 6>{}); frag(std::integral_constant<int, 7>{});
                }
            };
            const int fm = brow_c + BM <= M ? ep.fast_mode() : 0;          // workgroup-uniform
            if (!(dbg & 16)) dispatch_fast_mode<EP::kFastModes>(fm, ep_tile);
            ch_c = 0;
            ++tile_c;
        }
        // ---- the chunk in the registers goes to the other buffer (nobody reads it: its last readers passed the barrier below one
        // iteration ago), then the registers take the loads of the chunk after it ---------------------------------------------------
        if (!pending) break;                                   // nothing staged: the chunk just computed was the last one
        stash(cur ^ 1);
        pending = advance_fetch();
        if (pending && !(dbg & 4)) fetch();
        __syncthreads();
        cur ^= 1;
    }
}

// Column tiles per wave for N output columns: 3 (192-column workgroups) or 4 (256); 0 = the shape stays on gemm_lds_kernel
static inline int gemm_wide_ntw(int M, int N, int K) {
    if (leod_precision() != 1 || M < 4096 || N < 144 || K < 32 || (K & 3) || (N & 3)) return 0;
    const long p3 = (long)cdiv(N, 192) * 192, p4 = (long)cdiv(N, 256) * 256;
    const int ntw = p3 <= p4 ? 3 : 4;
    const long pad = ntw == 3 ? p3 : p4;
    if (pad * 3 > (long)N * 4) return 0;                       // more than a third of the MFMAs on padding columns
    return ntw;
}

template <int NTW, class AL, class BL, class EP>
static inline int launch_gemm_wide(const AL& al, const BL& bl, const EP& ep, int M, int K, int N, hipStream_t s) {
#ifdef LEOD_SHADOW_KERNELS
    if constexpr (!std::is_void<typename bl_shadow_type<BL>::type>::value) {
        if (const unsigned short* sh = bl_shadow_ptr(bl)) return launch_gemm_wide<NTW>(al, bl_as16(bl, sh), ep, M, K, N, s);
    }
#endif
    const int nbn = cdiv(N, 64 * NTW);
    const long ntiles = (long)cdiv(M, 128) * nbn;
    constexpr int dbg = 0;          // ablation bits (skip stores / MFMAs / loads): compile-time, for experiments
    // (The WN = 2 / KCH = 32 variant -- two 4-wave workgroups per CU -- is not instantiated: tools/kbench_gemm.py, stages 3-4, measured it
    // equal on the 53 k-row launches and 1.2-1.5x slower on the 13 k-row ones, whose 105-210 tiles then run on four waves per CU each:
    // 40 -> 57 us dgrad of fc1, 59 -> 75 us fc2; LN -> fc1 -> GELU alone gained, 66 -> 60 us.  The step did not move.)
    // one persistent 8-wave workgroup per CU (132-152 KB of LDS each)
    LEOD_BY_OPFMT16_IF(bl_fwd<BL>::value, hipLaunchKernelGGL((gemm_wide_bf16_kernel<NTW, AL, BL, EP, OF>), dim3((unsigned)(ntiles < 256 ? ntiles : 256)), dim3(512), 0, s, al, bl, ep, M, K, nbn, dbg));
    return leod_launch_status();
}
