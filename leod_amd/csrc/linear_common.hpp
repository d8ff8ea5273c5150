// Shared by the token-row contraction translation units (k_linear.hip: fp32-tensor forward; k_linear16.hip: 16-bit-tensor forward;
// k_linear_bwd.hip: input gradients; k_linear_wgrad.hip: weight gradients): tile pickers and the wave-autonomous row-streaming kernels.
// The entry points were one file until round 5; it took 7+ minutes to compile on its own, the four files build in parallel.
#pragma once
#include <type_traits>

#define LEOD_SHADOW_KERNELS 1        // bf16-shadow weight loaders for the Linear layers (gemm16.hpp: BLRows16 / BLTrans16)
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


static inline EpStore ep_store(float* out, long ld, int N) {
    EpStore e{};
    e.out = out; e.ld = ld; e.N = N; e.act = ACT_NONE;
    return e;
}

// ---------------------------------------------------------------------------------------------------------------------
// Wave-autonomous row-streaming GEMM for the short contractions of stages 1 and 2 (K = 48: N = 144 / 192, K = 96: N = 288 / 384
// in column slabs: LN -> qkv, LN -> fc1 + GELU).  A 64 x 48 workgroup of the LDS-staged GEMM lives ~11 us for 36 MFMAs per wave (operand loads, barrier, MFMAs, barrier,
// transposition, barrier, stores: two serialized memory latencies) and 4 of them per CU keep only ~2.5 TB/s in flight.  Here the
// whole weight matrix stays in LDS for the life of a persistent workgroup, every WAVE streams its own 16-row tiles with the A
// fragments loaded straight into the MFMA operand layout two tiles ahead, and the only LDS traffic besides the B fragments is a
// wave-private 16 x 64 transposition tile for 16-byte row stores -- no workgroup barrier after the prologue.
// ---------------------------------------------------------------------------------------------------------------------
// K = 16 * KC.  A workgroup owns the NTT column tiles of slab blockIdx.y (its weights: NTT*16 x K floats in LDS); the slabs of
// one row range get workgroup ids that differ by a multiple of 8, i.e. run on the same XCD and re-read the rows from its L2.
// DG = 0: forward (W [Ntot][K]; optional LayerNorm on load, optional exact-GELU second output).
// DG = 1 / 2: dgrad of a Linear, dx = (dy * kscale) @ W with W [K][Ntot] (read transposed into LDS once per workgroup); ln_w
// carries kscale (may be NULL), DG = 2 multiplies by gelu'(aux) with aux = out2 [M][Ntot] (the pre-activation), loaded at the
// START of the tile so that its wait falls behind the tile's MFMAs.
// BF (precision mode bf16): the resident weights are rounded to bf16 ONCE in the prologue (rows of K + 8 bf16: a 4 * odd dword
// stride keeps the ds_read_b64 fragment reads conflict-free), the streamed A fragments are packed in registers, and one
// v_mfma_f32_16x16x16_bf16 replaces the four fp32 MFMAs of a 16-k chunk.
// UF (precision mode bf16, stages 1-2 of the MLP): the pre-activation u is kept ONCE, as fp16 -- ACT stores only fp16(u) through out2
// (no fp32 u, no gelu(u) copy: 2 instead of 8 bytes per hidden element; its consumers apply GELU / GELU' on load), DG = 2 reads it back.
// OB: the output (the gradient du of the dgrad through GELU) is stored as bf16 -- both of its consumers (dgrad of fc1, fc1 weight
// gradient) feed it to bf16 MFMAs, so nothing changes numerically and 4 -> 2 bytes move on the write and on both reads.
// BF / OB are operand / output formats: 0 = fp32, 1 = bf16, 2 = fp16 (forward launches of precision mode 16f: weights and rows packed to fp16,
// the qkv rows stored as fp16)
template <int KC, int NTT, bool LN, bool ACT, int DG = 0, int BF = 0, bool UF = false, int OB = 0>
__global__ __launch_bounds__(256, (KC == 3 && NTT <= 9) ? 3 : 2) void rowstream48_kernel(const float* __restrict__ x, long ldx, float* __restrict__ stats_out,
                                                             const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps,
                                                             const float* __restrict__ W, const float* __restrict__ bias,
                                                             float* __restrict__ out, float* __restrict__ out2, int M, int Ntot) {
    constexpr int K = 16 * KC, LD = K + 8, N = NTT * 16, LDO = 68, NG = (NTT + 3) / 4;
    const int n0 = blockIdx.y * N;                              // first column of this slab
    W += DG ? (long)n0 : (long)n0 * K;
    if (bias) bias += n0;
    if (OB) out = reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(out) + n0);
    else if (!(ACT && UF)) out += n0;
    unsigned short* u16 = reinterpret_cast<unsigned short*>(out2) + n0;      // UF: the fp16 pre-activation [M][Ntot]
    if ((ACT || DG == 2) && !UF) out2 += n0;
    __shared__ __attribute__((aligned(16))) float sW[BF ? (N * LD) / 2 : N * LD];
    __shared__ __attribute__((aligned(16))) float sO[4][16 * LDO];
    unsigned short* sWh = reinterpret_cast<unsigned short*>(sW);        // BF: [N][LD] bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    if (DG == 0) {
        for (int e = tid; e < N * (K / 4); e += 256) {
            const int n = e / (K / 4), k4 = (e - n * (K / 4)) * 4;
            if constexpr (BF) *reinterpret_cast<s4*>(&sWh[n * LD + k4]) = pack16_raw<BF>(ld4(W + (long)n * K + k4));
            else *reinterpret_cast<f4*>(&sW[n * LD + k4]) = ld4(W + (long)n * K + k4);
        }
    } else {
        for (int e = tid; e < K * (N / 4); e += 256) {
            const int k = e / (N / 4), n4 = (e - k * (N / 4)) * 4;
            const f4 w = ld4(W + (long)k * Ntot + n4);
            if constexpr (BF) {
                const s4 wh = pack16_raw<BF>(w);
#pragma unroll
                for (int j = 0; j < 4; ++j) sWh[(n4 + j) * LD + k] = (unsigned short)wh[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) sW[(n4 + j) * LD + k] = w[j];
            }
        }
    }
    // everything a tile needs besides its own rows is loaded ONCE: a global load inside the tile loop makes the compiler wait
    // for vmcnt(0) at its first use, i.e. for every prefetched fragment and every store still in flight
    f4 lw[KC], lb[KC], bv[NG];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        lw[c] = (LN || (DG && ln_w)) ? ld4(ln_w + 16 * c + 4 * q) : f4{1.f, 1.f, 1.f, 1.f};
        lb[c] = LN ? ld4(ln_b + 16 * c + 4 * q) : zero4();
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int n = 64 * g + 4 * i;
        bv[g] = (bias && n < N) ? ld4(bias + n) : zero4();
    }
    __syncthreads();
    const int ntiles = (M + 15) / 16;
    const int stride = gridDim.x * 4;
    float* so = sO[wave];
    struct Frag { f4 a[KC]; };
    // branch-free: out-of-range tiles / rows read the last row again (never stored), so the loads carry no select and the
    // compiler has no reason to wait for them before their first real use two tiles later
    auto load = [&](Frag& f, int tile) {
        const long row = min((long)tile * 16 + i, (long)M - 1);
        const float* p = x + row * ldx + 4 * q;
#pragma unroll
        for (int c = 0; c < KC; ++c) f.a[c] = ld4(p + 16 * c);
    };
    // FULL tiles store unconditionally: a store behind a branch is invisible to the compiler's vmcnt bookkeeping, which then
    // waits for (nearly) everything in flight before the next tile's first MFMA
    auto compute = [&](const Frag& f, int tile, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const long row0 = (long)tile * 16;
        // LayerNorm statistics of row i from the fragments themselves (the row's 48 values sit in the 4 lanes i, i+16, i+32,
        // i+48): two-pass mean / variance like the reference's LayerNorm, (mean, rstd) kept for the backward pass
        float mean = 0.f, rstd = 1.f;
        if (LN) {
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < KC; ++c) sum += (f.a[c][0] + f.a[c][1]) + (f.a[c][2] + f.a[c][3]);
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            mean = sum * (1.0f / K);
            float var = 0.f;
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const f4 d = f.a[c] - mean;
                var += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
            }
            var += __shfl_xor(var, 16, 64);
            var += __shfl_xor(var, 32, 64);
            rstd = rsqrtf(var * (1.0f / K) + eps);
            if (blockIdx.y == 0 && (FULL || row0 + i < M)) {  // the 4 lanes of a row write the same pair (no q == 0 branch)
                float2 st; st.x = mean; st.y = rstd;
                *reinterpret_cast<float2*>(stats_out + 2 * (row0 + i)) = st;
            }
        }
        // dgrad through GELU: this tile's slice of the pre-activation, in the lane layout of the row stores below
        f4 ug[DG == 2 ? NTT : 1];
        if (DG == 2) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (NTT - 4 * g >= 4) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const long row = FULL ? row0 + q + 4 * p : min(row0 + q + 4 * p, (long)M - 1);
                        if constexpr (UF) ug[4 * g + p] = unpack_h16(*reinterpret_cast<const s4*>(u16 + row * Ntot + 64 * g + 4 * i));
                        else ug[4 * g + p] = ld4(out2 + row * Ntot + 64 * g + 4 * i);
                    }
                } else {
                    const long row = FULL ? row0 + (lane >> 2) : min(row0 + (lane >> 2), (long)M - 1);
                    if constexpr (UF) ug[4 * g] = unpack_h16(*reinterpret_cast<const s4*>(u16 + row * Ntot + 64 * g + 4 * (lane & 3)));
                    else ug[4 * g] = ld4(out2 + row * Ntot + 64 * g + 4 * (lane & 3));
                }
            }
        }
        f4 acc[NTT];
#pragma unroll
        for (int t = 0; t < NTT; ++t) acc[t] = zero4();
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            f4 av = f.a[c];
            if (LN) av = (av - mean) * rstd * lw[c] + lb[c];
            if (DG) av = av * lw[c];
            if constexpr (BF) {
                const s4 pa = pack16<BF>(av);
#pragma unroll
                for (int t = 0; t < NTT; ++t)
                    acc[t] = mfma16_16<BF>(pa, *reinterpret_cast<const s4*>(&sWh[(16 * t + i) * LD + 16 * c + 4 * q]), acc[t]);
            } else {
#pragma unroll
            for (int t = 0; t < NTT; ++t) {
                const f4 b = *reinterpret_cast<const f4*>(&sW[(16 * t + i) * LD + 16 * c + 4 * q]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t] = mfma16(av[j], b[j], acc[t]);
            }
            }
        }
        auto emit = [&](long row, int n, f4 v, const f4& u) {
            if (!FULL && row >= M) return;
            if (DG == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= gelu_erf_grad(u[j]);
            }
            if constexpr (ACT && UF) { *reinterpret_cast<s4*>(u16 + row * Ntot + n) = pack_h16(v); return; }
            if constexpr (OB) { *reinterpret_cast<s4*>(reinterpret_cast<unsigned short*>(out) + row * Ntot + n) = pack16<OB>(v); return; }
            *reinterpret_cast<f4*>(out + row * Ntot + n) = v;
            if (ACT) {
                f4 ge;
#pragma unroll
                for (int j = 0; j < 4; ++j) ge[j] = gelu_erf(v[j]);
                *reinterpret_cast<f4*>(out2 + row * Ntot + n) = ge;
            }
        };
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int nt_g = NTT - 4 * g >= 4 ? 4 : NTT - 4 * g;              // column tiles of this group (compile time after unroll)
            // wave-private tile: LDS operations of one wave execute in order, so only the COMPILER must keep write -> read ->
            // write order (no fence: it would drain vmcnt, i.e. the prefetched fragments and the stores in flight)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 4 * g; t < NTT && t < 4 * g + 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) so[(4 * q + r) * LDO + 16 * (t - 4 * g) + i] = acc[t][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (nt_g == 4) {                                                 // 64 columns: lane = (4 rows q + 4p) x 16-byte column i
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int lr = q + 4 * p;
                    emit(row0 + lr, 64 * g + 4 * i, *reinterpret_cast<const f4*>(&so[lr * LDO + 4 * i]) + bv[g], ug[DG == 2 ? 4 * g + p : 0]);
                }
            } else {                                                         // 16 columns: every lane stores one (row, 16-byte column)
                const int lr = lane >> 2, c4 = lane & 3;
                const f4 b4 = {__shfl(bv[g][0], c4, 64), __shfl(bv[g][1], c4, 64), __shfl(bv[g][2], c4, 64), __shfl(bv[g][3], c4, 64)};
                emit(row0 + lr, 64 * g + 4 * c4, *reinterpret_cast<const f4*>(&so[lr * LDO + 4 * c4]) + b4, ug[DG == 2 ? 4 * g : 0]);
            }
        }
    };
    static_assert(NTT % 4 == 0 || NTT % 4 == 1, "remainder groups of 2 or 3 column tiles are not laid out");
    const int nfull = M / 16;
    int tile = blockIdx.x * 4 + wave;
    Frag f0, f1, f2;
    load(f0, tile);
    load(f1, tile + stride);
    const std::true_type full{};
    while (true) {
        load(f2, tile + 2 * stride);
        if (tile >= nfull) break;
        compute(f0, tile, full); tile += stride;
        load(f0, tile + 2 * stride);
        if (tile >= nfull) { f0 = f1; break; }
        compute(f1, tile, full); tile += stride;
        load(f1, tile + 2 * stride);
        if (tile >= nfull) { f0 = f2; break; }
        compute(f2, tile, full); tile += stride;
    }
    if (tile == nfull && (M & 15)) compute(f0, tile, std::false_type{});       // the ragged last tile, on whichever wave owns it
}

// ---------------------------------------------------------------------------------------------------------------------
// The same streaming scheme for the wide-input, 48-column-output contractions of stage 1 (K = 144 / 192 -> 48):
//   MODE 0: out = res + gamma * (A W^T + bias)      W [48][K]   (fc2 + LayerScale + residual, maxvit.py:268-269)
//   MODE 1: out = A W                                W [K][48]   (dgrad of fc1 / qkv)
//   MODE 2: MODE 1 followed by the LayerNorm backward of the producer in the same epilogue (maxvit.py:267-269: x -> norm ->
//           Linear): dn = A W never leaves the registers; in accumulator layout a lane holds columns 16t + i of rows 4q + r, so
//           the two row means are one 16-lane reduction each, dx = rstd (dn w - mean(dn w) - xhat mean(dn w xhat)) + dres goes
//           through the transposition tile, and dgamma / dbeta accumulate per lane over all tiles (one atomic per column and
//           wave at the end).  xin / stats / dres of the tile are loaded before its MFMAs.
// One wave = one 16-row tile; 48 output columns = 12 float4 per row = 3 per lane (idx = 64 p + lane -> row idx / 12,
// column idx % 12); the residual slice of the tile is loaded before its MFMAs.
// ---------------------------------------------------------------------------------------------------------------------
// AF = 1 (MODE 0, precision mode bf16): x is the fp16 pre-activation of the MLP hidden, A = gelu(x) evaluated on the fragments.
// AF = 2 (MODE 1 / 2): x is a bf16 gradient (du / dqkv): its fragments ARE the MFMA operands.
// NTN = output column tiles: 3 (48 columns, stage 1: 4 waves, two workgroups per CU) or 6 (96 columns, stage 2, bf16 mode with
// 16-bit A rows only: 8 waves, the 75 KB weight tile allows one workgroup per CU).
template <int KC, int MODE, int BF = 0, int AF = 0, int NTN = 3>
__global__ __launch_bounds__(NTN == 3 ? 256 : 512, NTN == 3 ? 2 : 1) void rowstream_narrow_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                  const float* __restrict__ bias, const float* __restrict__ gamma,
                                                                  const float* __restrict__ res, float* __restrict__ out, int M,
                                                                  const float* __restrict__ xin = nullptr,
                                                                  const float* __restrict__ stats = nullptr,
                                                                  float* __restrict__ dgamma = nullptr, float* __restrict__ dbeta = nullptr) {
    constexpr int K = 16 * KC, LD = K + 8, N = 16 * NTN, LDO = N + 4, NWV = NTN == 3 ? 4 : 8, NTHR = 64 * NWV, F4R = N / 4;
    __shared__ __attribute__((aligned(16))) float sW[BF ? (N * LD) / 2 : N * LD];
    __shared__ __attribute__((aligned(16))) float sO[NWV][16 * LDO];
    unsigned short* sWh = reinterpret_cast<unsigned short*>(sW);        // BF: [N][LD] bf16 (see rowstream48_kernel)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    if (MODE == 0) {
        for (int e = tid; e < N * (K / 4); e += NTHR) {
            const int n = e / (K / 4), k4 = (e - n * (K / 4)) * 4;
            if constexpr (BF) *reinterpret_cast<s4*>(&sWh[n * LD + k4]) = pack16_raw<BF>(ld4(W + (long)n * K + k4));
            else *reinterpret_cast<f4*>(&sW[n * LD + k4]) = ld4(W + (long)n * K + k4);
        }
    } else {
        for (int e = tid; e < K * (N / 4); e += NTHR) {
            const int k = e / (N / 4), n4 = (e - k * (N / 4)) * 4;
            const f4 w = ld4(W + (long)k * N + n4);
            if constexpr (BF) {
                const s4 wh = pack16_raw<BF>(w);
#pragma unroll
                for (int j = 0; j < 4; ++j) sWh[(n4 + j) * LD + k] = (unsigned short)wh[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) sW[(n4 + j) * LD + k] = w[j];
            }
        }
    }
    int lr[NTN], c4[NTN];
    f4 b4[NTN], g4[NTN];
#pragma unroll
    for (int p = 0; p < NTN; ++p) {
        const int idx = 64 * p + lane;
        lr[p] = idx / F4R; c4[p] = idx - lr[p] * F4R;
        b4[p] = (MODE == 0 && bias) ? ld4(bias + 4 * c4[p]) : zero4();
        g4[p] = (MODE == 0 && gamma) ? ld4(gamma + 4 * c4[p]) : f4{1.f, 1.f, 1.f, 1.f};
    }
    float lnw[NTN], agam[NTN], abet[NTN];
#pragma unroll
    for (int t = 0; t < NTN; ++t) { lnw[t] = 1.f; agam[t] = 0.f; abet[t] = 0.f; }
    if (MODE == 2) {
#pragma unroll
        for (int t = 0; t < NTN; ++t) lnw[t] = gamma[16 * t + i];                 // LayerNorm weight of column 16t + i
    }
    __syncthreads();
    const int stride = gridDim.x * NWV;
    float* so = sO[wave];
    typedef typename std::conditional<AF != 0, s4, f4>::type AFrag;
    struct Frag { AFrag a[KC]; };
    auto load = [&](Frag& f, int tile) {
        const long row = min((long)tile * 16 + i, (long)M - 1);
        if constexpr (AF) {
            const unsigned short* p = reinterpret_cast<const unsigned short*>(x) + row * K + 4 * q;
#pragma unroll
            for (int c = 0; c < KC; ++c) f.a[c] = *reinterpret_cast<const s4*>(p + 16 * c);
        } else {
            const float* p = x + row * K + 4 * q;
#pragma unroll
            for (int c = 0; c < KC; ++c) f.a[c] = ld4(p + 16 * c);
        }
    };
    auto compute = [&](const Frag& f, int tile, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const long row0 = (long)tile * 16;
        f4 r4[NTN];
        if (MODE == 0 || (MODE == 2 && res)) {
#pragma unroll
            for (int p = 0; p < NTN; ++p) {
                const long row = FULL ? row0 + lr[p] : min(row0 + lr[p], (long)M - 1);
                r4[p] = ld4(res + row * N + 4 * c4[p]);
            }
        }
        float xi[NTN][4], mean[4], rstd[4];
        if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long row = FULL ? row0 + 4 * q + r : min(row0 + 4 * q + r, (long)M - 1);
                const float2 st = *reinterpret_cast<const float2*>(stats + 2 * row);
                mean[r] = st.x; rstd[r] = st.y;
#pragma unroll
                for (int t = 0; t < NTN; ++t) xi[t][r] = xin[row * N + 16 * t + i];
            }
        }
        f4 acc[NTN];
#pragma unroll
        for (int t = 0; t < NTN; ++t) acc[t] = zero4();
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            if constexpr (BF) {
                s4 pa;
                if constexpr (AF == 2) pa = f.a[c];
                else if constexpr (AF == 1) {
                    f4 u = unpack_h16(f.a[c]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) u[j] = gelu_erf(u[j]);
                    pa = pack16<BF>(u);
                } else pa = pack16<BF>(f.a[c]);
#pragma unroll
                for (int t = 0; t < NTN; ++t)
                    acc[t] = mfma16_16<BF>(pa, *reinterpret_cast<const s4*>(&sWh[(16 * t + i) * LD + 16 * c + 4 * q]), acc[t]);
            } else if constexpr (AF == 0) {
#pragma unroll
            for (int t = 0; t < NTN; ++t) {
                const f4 b = *reinterpret_cast<const f4*>(&sW[(16 * t + i) * LD + 16 * c + 4 * q]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t] = mfma16(f.a[c][j], b[j], acc[t]);
            }
            }
        }
        if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool live = FULL || row0 + 4 * q + r < M;
                float s1 = 0.f, s2 = 0.f, xh[NTN], gw[NTN];
#pragma unroll
                for (int t = 0; t < NTN; ++t) {
                    const float dn = live ? acc[t][r] : 0.f;
                    xh[t] = (xi[t][r] - mean[r]) * rstd[r];
                    gw[t] = dn * lnw[t];
                    agam[t] += dn * xh[t]; abet[t] += dn;
                    s1 += gw[t]; s2 += gw[t] * xh[t];
                }
                s1 = row16_sum(s1) * (1.0f / N);
                s2 = row16_sum(s2) * (1.0f / N);
#pragma unroll
                for (int t = 0; t < NTN; ++t) acc[t][r] = (gw[t] - s1 - xh[t] * s2) * rstd[r];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // wave-private tile: compiler ordering only (see above)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < NTN; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) so[(4 * q + r) * LDO + 16 * t + i] = acc[t][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < NTN; ++p) {
            f4 v = *reinterpret_cast<const f4*>(&so[lr[p] * LDO + 4 * c4[p]]);
            if (MODE == 0) v = r4[p] + g4[p] * (v + b4[p]);
            if (MODE == 2 && res) v = v + r4[p];
            if (FULL || row0 + lr[p] < M) *reinterpret_cast<f4*>(out + (row0 + lr[p]) * N + 4 * c4[p]) = v;
        }
    };
    const int nfull = M / 16;
    int tile = blockIdx.x * NWV + wave;
    const std::true_type full{};
    if constexpr (NTN == 3) {                                 // fragments two tiles ahead (three register sets)
        Frag f0, f1, f2;
        load(f0, tile);
        load(f1, tile + stride);
        while (true) {
            load(f2, tile + 2 * stride);
            if (tile >= nfull) break;
            compute(f0, tile, full); tile += stride;
            load(f0, tile + 2 * stride);
            if (tile >= nfull) { f0 = f1; break; }
            compute(f1, tile, full); tile += stride;
            load(f1, tile + 2 * stride);
            if (tile >= nfull) { f0 = f2; break; }
            compute(f2, tile, full); tile += stride;
        }
        if (tile == nfull && (M & 15)) compute(f0, tile, std::false_type{});
    } else {                                                  // 96 columns: one tile ahead (two sets: a third would spill), 8 waves per CU
        Frag f0, f1;
        load(f0, tile);
        while (true) {
            load(f1, tile + stride);
            if (tile >= nfull) break;
            compute(f0, tile, full); tile += stride;
            load(f0, tile + stride);
            if (tile >= nfull) { f0 = f1; break; }
            compute(f1, tile, full); tile += stride;
        }
        if (tile == nfull && (M & 15)) compute(f0, tile, std::false_type{});
    }
    if (MODE == 2) {                                         // column sums of this wave: over the 4 row groups, then one atomic
#pragma unroll
        for (int t = 0; t < NTN; ++t) {
            float a = agam[t], b = abet[t];
            a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
            if (q == 0) { atomicAdd(dgamma + 16 * t + i, a); atomicAdd(dbeta + 16 * t + i, b); }
        }
    }
}
static inline bool use_rowstream_narrow(int M, int Kc, int Nout) {
    constexpr int on = 2;
    return on >= 2 && M >= 16384 && Nout == 48 && (Kc == 144 || Kc == 192);
}
// stage 2 of RVT-S in precision mode bf16: 16-bit A rows (fp16 hidden, bf16 du / dqkv), 288 / 384 -> 96 columns
static inline bool use_rowstream_narrow96(int M, int Kc, int Nout) {
    static const int on = 1;
    return on && leod_precision() == 1 && M >= 16384 && Nout == 96 && (Kc == 288 || Kc == 384);
}
template <int MODE>
static int launch_rowstream_narrow(const float* x, const float* W, const float* bias, const float* gamma, const float* res,
                                   float* out, int M, int Kc, hipStream_t s) {
    const int grid = min(cdiv(cdiv(M, 16), 4), 256 * 2);
    LEOD_BY_OPFMT_IF(MODE == 0, {
        if (Kc == 192) hipLaunchKernelGGL((rowstream_narrow_kernel<12, MODE, OF>), dim3(grid), dim3(256), 0, s, x, W, bias, gamma, res, out, M);
        else hipLaunchKernelGGL((rowstream_narrow_kernel<9, MODE, OF>), dim3(grid), dim3(256), 0, s, x, W, bias, gamma, res, out, M);
    });
    return leod_launch_status();
}

// (K, N) -> column tiles per slab; 0 = shape not covered.  K = 48: the whole N (9 / 12 tiles); K = 96: slabs of 9 (N = 288) or
// 8 (N = 384) tiles, so that weights + wave tiles of two workgroups fit the 160 KB of a CU
static inline int rowstream_slab(int M, int N, int K) {
    constexpr int on = 2;     // 0 off, 1 stage 1 only, 2 stages 1 + 2
    if (!on || M < 16384) return 0;
    if (K == 48 && (N == 144 || N == 192)) return N / 16;                    // RVT-S stage 1
    if (K == 96 && on >= 2) return N == 288 ? 9 : (N == 384 ? 8 : 0);       // RVT-S stage 2
    if (K == 64 && on >= 2) return N == 192 ? 12 : (N == 256 ? 8 : 0);      // RVT-B stage 1 (qkv whole, fc1 in two slabs)
    return 0;
}
template <int KC, int NTT, bool ACT>
static int launch_rowstream48(const float* x, long ldx, float* stats, const float* ln_w, const float* ln_b, float eps, const float* W,
                              const float* bias, float* out, float* out2, int M, int N, hipStream_t s) {
    const int per_cu = (KC == 3 && NTT <= 9) ? 3 : 2;        // resident workgroups per CU (registers / LDS): one wave of them
    const int slabs = N / (16 * NTT);
    const int gx = min(cdiv(cdiv(M, 16), 4), max(8, (256 * per_cu / slabs) & ~7));   // multiple of 8: slabs of a row range share an XCD
    const dim3 grid(gx, slabs);
    LEOD_BY_OPFMT({
        if (stats) hipLaunchKernelGGL((rowstream48_kernel<KC, NTT, true, ACT, 0, OF>), grid, dim3(256), 0, s, x, ldx, stats, ln_w, ln_b, eps, W, bias, out, out2, M, N);
        else hipLaunchKernelGGL((rowstream48_kernel<KC, NTT, false, ACT, 0, OF>), grid, dim3(256), 0, s, x, ldx, stats, ln_w, ln_b, eps, W, bias, out, out2, M, N);
    });
    return leod_launch_status();
}

template <int KC, int NTT>
static int launch_rowstream_dgrad(const float* dy, long lddy, const float* kscale, const float* W, const float* aux_u, float* dx,
                                  int M, int Nout, hipStream_t s) {
    const int slabs = Nout / (16 * NTT);
    const int gx = min(cdiv(cdiv(M, 16), 4), max(8, (256 * 2 / slabs) & ~7));
    const dim3 grid(gx, slabs);
    float* aux = const_cast<float*>(aux_u);
    if (leod_precision() == 1) {
        if (aux_u) hipLaunchKernelGGL((rowstream48_kernel<KC, NTT, false, false, 2, 1>), grid, dim3(256), 0, s, dy, lddy, nullptr, kscale, nullptr, 0.f, W, nullptr, dx, aux, M, Nout);
        else hipLaunchKernelGGL((rowstream48_kernel<KC, NTT, false, false, 1, 1>), grid, dim3(256), 0, s, dy, lddy, nullptr, kscale, nullptr, 0.f, W, nullptr, dx, aux, M, Nout);
        return leod_launch_status();
    }
    if (aux_u) hipLaunchKernelGGL((rowstream48_kernel<KC, NTT, false, false, 2>), grid, dim3(256), 0, s, dy, lddy, nullptr, kscale, nullptr, 0.f, W, nullptr, dx, aux, M, Nout);
    else hipLaunchKernelGGL((rowstream48_kernel<KC, NTT, false, false, 1>), grid, dim3(256), 0, s, dy, lddy, nullptr, kscale, nullptr, 0.f, W, nullptr, dx, aux, M, Nout);
    return leod_launch_status();
}
