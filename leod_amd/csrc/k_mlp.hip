// Fused MLP of a MaxViT block, precision mode bf16, stage-1 width (maxvit.py:110-118, 268-269):
//
//     z = y + gamma2 * ( gelu( LN(y) W1^T + b1 ) W2^T + b2 )
//
// as ONE wave-autonomous row-streaming kernel: a wave owns a 16-row tile of y and never meets another wave after the prologue.  The
// unfused pair (norm2 -> fc1 -> fp16 hidden, then fc2 + LayerScale + residual) moved y, the hidden twice and z through HBM: 1.16 GB per
// stage-1 block of the RVT-S step; here the hidden lives in registers -- 0.33 GB in inference, 0.66 GB when the backward pass wants the
// fp16 pre-activation and the LayerNorm statistics back (SU).
//
// MFMA chaining without a transposition: fc1 is evaluated TRANSPOSED -- u^T tile = W1 tile (A operand, rows = hidden units) x n^T (B
// operand: lane (row i, quad q) holds n[i][4q..4q+3], which is exactly how a row fragment of y is loaded) -- so that its accumulator
// layout, lane (i, q) holds u[row i][16t + 4q + r], IS the A-operand layout of fc2's contraction over the hidden units (k = 16t + 4q + r
// of row i).  Two hidden tiles t, t + 1 feed one v_mfma_f32_16x16x32_bf16 (the 8 k-values of a lane are a permutation of the chunk; W2's
// fragments are read with the same permutation).  fc2's accumulators are in the usual C layout and leave through the wave-private
// transposition tile as 16-byte row stores, LayerScale and residual applied on the way.
//
// What bounds it (profiles/r04_d_kbench_mlp.txt): the GELU.  Without it the inference form runs at 99 us for M = 860 160 (3.3 TB/s);
// with it 168 us -- v_rcp_f32 + v_exp_f32 per hidden element are quarter-rate instructions (48 elements per lane and tile), and the
// unfused kernels paid the same.  Packed fp32 arithmetic around them (gelu_parts2) changed nothing; the 16-byte stores of the permuted
// hidden units (mlp_unit_of_row) were worth 30 us in the training form.
#include "common.hpp"
#include <type_traits>

// GELU / GELU' (the Abramowitz-Stegun form of common.hpp) on TWO values at a time: the kernels below evaluate them on 48 hidden elements
// per lane and 16-row tile, which made the scalar form their largest VALU cost; f2_ arithmetic maps to v_pk_mul_f32 / v_pk_fma_f32.
__device__ __forceinline__ void gelu_parts2(f2_ x, f2_& cdf, f2_& e) {
    const f2_ z = __builtin_elementwise_abs(x) * 0.70710678118654752440f;
    const f2_ d = z * 0.3275911f + 1.0f;
    const f2_ t = {fast_rcp(d.x), fast_rcp(d.y)};
    const f2_ poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const f2_ a = (z * z) * -1.4426950408889634f;
    e = f2_{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
    const f2_ ht = (0.5f * poly) * e;
    cdf = f2_{x.x >= 0.f ? 1.0f - ht.x : ht.x, x.y >= 0.f ? 1.0f - ht.y : ht.y};
}
__device__ __forceinline__ f4 gelu4(f4 u) {
    f2_ c0, e0, c1, e1;
    const f2_ lo = {u.x, u.y}, hi = {u.z, u.w};
    gelu_parts2(lo, c0, e0);
    gelu_parts2(hi, c1, e1);
    const f2_ a = lo * c0, b = hi * c1;
    return f4{a.x, a.y, b.x, b.y};
}
__device__ __forceinline__ f4 gelu_grad4(f4 u) {              // Phi(u) + u * phi(u)
    f2_ c0, e0, c1, e1;
    const f2_ lo = {u.x, u.y}, hi = {u.z, u.w};
    gelu_parts2(lo, c0, e0);
    gelu_parts2(hi, c1, e1);
    const f2_ a = (lo * 0.39894228040143267794f) * e0 + c0, b = (hi * 0.39894228040143267794f) * e1 + c1;
    return f4{a.x, a.y, b.x, b.y};
}

// Hidden units are assigned to the MFMA rows of a tile PAIR (t = 2p, 2p + 1) such that the 8 values a lane ends up with -- rows 4q + r of
// both tiles -- are 8 CONSECUTIVE hidden units 32p + 8q .. + 7: one 16-byte store of the fp16 pre-activation / bf16 gradient per pair and
// one 16-byte fragment read of the next contraction's weights.  LDS row R = 16t + i of the A-operand matrices (and of the bias) holds unit
__host__ __device__ constexpr int mlp_unit_of_row(int R) { return 32 * (R >> 5) + 8 * ((R & 15) >> 2) + 4 * ((R >> 4) & 1) + (R & 3); }

// KC: input / output width K = 16 KC; NHT: hidden width H = 16 NHT (even); SU: also store u16 = fp16(u) [M][H] and stats [M][2]
// OF: operand format of the three contractions, 1 = bf16, 2 = fp16 (precision mode 16f)
template <int KC, int NHT, bool SU, int OF = 1>
__global__ __launch_bounds__(256, 2) void mlp_fwd_fused_kernel(const float* __restrict__ y, const float* __restrict__ ln_w,
                                                               const float* __restrict__ ln_b, float eps, const float* __restrict__ W1,
                                                               const float* __restrict__ b1, const float* __restrict__ W2,
                                                               const float* __restrict__ b2, const float* __restrict__ g2,
                                                               float* __restrict__ out, unsigned short* __restrict__ u16,
                                                               float* __restrict__ stats, int M) {
    constexpr int K = 16 * KC, H = 16 * NHT, LD1 = K + 8, LD2 = H + 16, LDO = K + 4, F4R = K / 4, NP = KC;     // LD2: 8 (mod 16) dwords for b128 reads
    static_assert(NHT % 2 == 0, "hidden tiles are consumed in pairs");
    __shared__ __attribute__((aligned(16))) unsigned short sW1[H * LD1];      // [hidden j][k] bf16, row stride 4 * odd dwords
    __shared__ __attribute__((aligned(16))) unsigned short sW2[K * LD2];      // [out n][hidden j] bf16
    __shared__ __attribute__((aligned(16))) float sB1[H];
    __shared__ __attribute__((aligned(16))) float sO[4][16 * LDO];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    for (int e = tid; e < H * (K / 4); e += 256) {
        const int R = e / (K / 4), k4 = (e - R * (K / 4)) * 4;
        *reinterpret_cast<s4*>(&sW1[R * LD1 + k4]) = pack16_raw<OF>(ld4(W1 + (long)mlp_unit_of_row(R) * K + k4));
    }
    for (int e = tid; e < K * (H / 4); e += 256) {
        const int n = e / (H / 4), j4 = (e - n * (H / 4)) * 4;
        *reinterpret_cast<s4*>(&sW2[n * LD2 + j4]) = pack16_raw<OF>(ld4(W2 + (long)n * H + j4));
    }
    for (int e = tid; e < H; e += 256) sB1[e] = b1 ? b1[mlp_unit_of_row(e)] : 0.f;
    f4 lw[KC], lb[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) { lw[c] = ld4(ln_w + 16 * c + 4 * q); lb[c] = ld4(ln_b + 16 * c + 4 * q); }
    int lr[NP], c4[NP];
    f4 b4[NP], g4[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int idx = 64 * p + lane;
        lr[p] = idx / F4R; c4[p] = idx - lr[p] * F4R;
        b4[p] = b2 ? ld4(b2 + 4 * c4[p]) : zero4();
        g4[p] = g2 ? ld4(g2 + 4 * c4[p]) : f4{1.f, 1.f, 1.f, 1.f};
    }
    __syncthreads();
    const int stride = gridDim.x * 4;
    float* so = sO[wave];
    struct Frag { f4 a[KC]; };
    auto load = [&](Frag& f, int tile) {                       // branch-free: out-of-range rows read the last row again (never stored)
        const long row = min((long)tile * 16 + i, (long)M - 1);
        const float* p = y + row * K + 4 * q;
#pragma unroll
        for (int c = 0; c < KC; ++c) f.a[c] = ld4(p + 16 * c);
    };
    auto compute = [&](const Frag& f, int tile, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const long row0 = (long)tile * 16;
        f4 r4[NP];                                             // residual slice in the layout of the row stores
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const long row = FULL ? row0 + lr[p] : min(row0 + lr[p], (long)M - 1);
            r4[p] = ld4(y + row * K + 4 * c4[p]);
        }
        // LayerNorm statistics of row i from the fragments (its K values sit in the 4 lanes i, i + 16, i + 32, i + 48): two-pass
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < KC; ++c) sum += (f.a[c][0] + f.a[c][1]) + (f.a[c][2] + f.a[c][3]);
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / K);
        float var = 0.f;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const f4 d = f.a[c] - mean;
            var += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
        var += __shfl_xor(var, 16, 64);
        var += __shfl_xor(var, 32, 64);
        const float rstd = rsqrtf(var * (1.0f / K) + eps);
        const bool row_ok = FULL || row0 + i < M;
        if (SU && row_ok) {
            float2 st; st.x = mean; st.y = rstd;
            *reinterpret_cast<float2*>(stats + 2 * (row0 + i)) = st;
        }
        s4 nb[KC];
#pragma unroll
        for (int c = 0; c < KC; ++c) nb[c] = pack16<OF>((f.a[c] - mean) * rstd * lw[c] + lb[c]);
        // fc1 transposed, GELU, fc2: hidden tiles two at a time
        f4 acc[KC];
#pragma unroll
        for (int c = 0; c < KC; ++c) acc[c] = zero4();
        unsigned short* urow = u16 + (row0 + i) * H + 8 * q;
#pragma unroll
        for (int t = 0; t < NHT; t += 2) {
            s4 ha[2], hu[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f4 u = *reinterpret_cast<const f4*>(&sB1[16 * (t + h) + 4 * q]);
#pragma unroll
                for (int c = 0; c < KC; ++c)
                    u = mfma16_16<OF>(*reinterpret_cast<const s4*>(&sW1[(16 * (t + h) + i) * LD1 + 16 * c + 4 * q]), nb[c], u);
                if (SU) hu[h] = pack_h16(u);
                ha[h] = pack16<OF>(gelu4(u));
            }
            // the lane's 8 values of this pair = hidden units 16t + 8q .. + 7 of row i
            if (SU && row_ok) *reinterpret_cast<s8v*>(urow + 16 * t) = __builtin_shufflevector(hu[0], hu[1], 0, 1, 2, 3, 4, 5, 6, 7);
            const s8v a8 = __builtin_shufflevector(ha[0], ha[1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int c = 0; c < KC; ++c)
                acc[c] = mfma32_16<OF>(a8, *reinterpret_cast<const s8v*>(&sW2[(16 * c + i) * LD2 + 16 * t + 8 * q]), acc[c]);
        }
        // acc[c][r] = t[row 4q + r][col 16c + i] -> rows through the wave-private tile (LDS operations of one wave execute in order: only
        // the compiler must keep write -> read -> write order, no fence that would drain the prefetched fragments)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) so[(4 * q + r) * LDO + 16 * c + i] = acc[c][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            f4 v = *reinterpret_cast<const f4*>(&so[lr[p] * LDO + 4 * c4[p]]);
            v = r4[p] + g4[p] * (v + b4[p]);
            if (FULL || row0 + lr[p] < M) *reinterpret_cast<f4*>(out + (row0 + lr[p]) * K + 4 * c4[p]) = v;
        }
    };
    const int nfull = M / 16;
    int tile = blockIdx.x * 4 + wave;
    const std::true_type full{};
    Frag f0, f1, f2;                                          // fragments two tiles ahead
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
}

// z[M,K] = y + g2 * (gelu(LN(y) W1^T + b1) W2^T + b2); u16 / stats (both or neither): the fp16 pre-activation [M,H] and the LayerNorm
// (mean, rstd) [M,2] the backward pass reads.  Precision mode bf16, K = 48, H = 192, M >= 16384 (RVT-S / -T stage 1): anything else
// returns LEOD_ERR_UNSUPPORTED and the caller runs leod_ln_linear_gelu16_fwd + leod_linear_lsres_gelu16_fwd.
LEOD_API int leod_mlp_fwd_fused(const float* y, const float* ln_w, const float* ln_b, float eps, const float* W1, const float* b1,
                                const float* W2, const float* b2, const float* g2, float* out, void* u16, float* stats, int M, int H,
                                int K, hipStream_t stream) {
    if (!y || !ln_w || !ln_b || !W1 || !W2 || !out || ((u16 == nullptr) != (stats == nullptr))) return LEOD_ERR_ARG;
    if (leod_precision() != 1 || !((K == 48 && H == 192) || (K == 64 && H == 256)) || M < 16384) return LEOD_ERR_UNSUPPORTED;
    const int grid = min(cdiv(cdiv(M, 16), 4), 256 * 2);       // two resident workgroups per CU (1 / 4 per CU measured slower: 222 / 184 vs 181 us)
    LeodFwdScope fwd_scope;
    if (K == 64) {                                             // RVT-B stage 1
        LEOD_BY_OPFMT16({
            if (u16) hipLaunchKernelGGL((mlp_fwd_fused_kernel<4, 16, true, OF>), dim3(grid), dim3(256), 0, stream, y, ln_w, ln_b, eps, W1, b1, W2, b2, g2, out,
                                        reinterpret_cast<unsigned short*>(u16), stats, M);
            else hipLaunchKernelGGL((mlp_fwd_fused_kernel<4, 16, false, OF>), dim3(grid), dim3(256), 0, stream, y, ln_w, ln_b, eps, W1, b1, W2, b2, g2, out,
                                    nullptr, nullptr, M);
        });
        return leod_launch_status();
    }
    LEOD_BY_OPFMT16({
        if (u16) hipLaunchKernelGGL((mlp_fwd_fused_kernel<3, 12, true, OF>), dim3(grid), dim3(256), 0, stream, y, ln_w, ln_b, eps, W1, b1, W2, b2, g2, out,
                                    reinterpret_cast<unsigned short*>(u16), stats, M);
        else hipLaunchKernelGGL((mlp_fwd_fused_kernel<3, 12, false, OF>), dim3(grid), dim3(256), 0, stream, y, ln_w, ln_b, eps, W1, b1, W2, b2, g2, out,
                                nullptr, nullptr, M);
    });
    return leod_launch_status();
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// Backward of the same block along the activation path (the dgrad chain of the MLP), one launch:
//     u   = LN(y) W1^T + b1                 (recomputed from y: 36 MFMAs per 16 rows instead of reading 2 bytes x H per row)
//     du  = ((dz * g2) W2) * gelu'(u)       -> stored as bf16 rows for the fc1 weight gradient (SDU)
//     dy  = dz + LayerNorm-backward(du W1)  ; dgamma / dbeta of norm2 accumulate
// replacing leod_linear_dgrad_gelu16 + leod_linear_dgrad_lnbwd (the first wrote du, the second read it back together with the fp16
// hidden).  Same chaining as the forward kernel: u and dh = (dz g2) W2 are evaluated TRANSPOSED (weights as A operands, the row
// fragments of y / dz as B operands), so du arrives in the A-operand layout of the contraction over the hidden units that follows
// (du W1, against W1^T rows in LDS); that product is in C layout, where the LayerNorm backward of rowstream_narrow_kernel<.., MODE 2>
// applies unchanged (row means by 16-lane reductions, dx through the wave-private transposition tile).
template <int KC, int NHT, bool SDU>
__global__ __launch_bounds__(256, (KC <= 3 ? 2 : 1)) void mlp_bwd_dgrad_fused_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                                                                     const float* __restrict__ stats, const float* __restrict__ ln_w,
                                                                     const float* __restrict__ ln_b, const float* __restrict__ W1,
                                                                     const float* __restrict__ b1, const float* __restrict__ W2,
                                                                     const float* __restrict__ g2, float* __restrict__ dy,
                                                                     unsigned short* __restrict__ du16, float* __restrict__ dgamma,
                                                                     float* __restrict__ dbeta, int M) {
    constexpr int K = 16 * KC, H = 16 * NHT, LD1 = K + 8, LD2 = H + 16, LDO = K + 4, F4R = K / 4, NP = KC;
    static_assert(NHT % 2 == 0, "hidden tiles are consumed in pairs");
    extern __shared__ __attribute__((aligned(16))) unsigned char mlp_smem[];
    unsigned short* sW1 = reinterpret_cast<unsigned short*>(mlp_smem);        // [j][k]   (A operand of u^T)
    unsigned short* sW2T = sW1 + H * LD1;                                      // [j][n] = W2[n][j]   (A operand of dh^T)
    unsigned short* sW1T = sW2T + H * LD1;                                     // [k][j] = W1[j][k]   (B operand of du W1)
    float* sB1 = reinterpret_cast<float*>(sW1T + K * LD2);
    float* sOall = sB1 + H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    for (int e = tid; e < H * (K / 4); e += 256) {               // row R of sW1 / sW2T / sB1 holds hidden unit mlp_unit_of_row(R); sW1T is in natural order
        const int R = e / (K / 4), k4 = (e - R * (K / 4)) * 4, j = mlp_unit_of_row(R);
        const s4 w = pack_bf16(ld4(W1 + (long)j * K + k4));
        *reinterpret_cast<s4*>(&sW1[R * LD1 + k4]) = w;
#pragma unroll
        for (int a = 0; a < 4; ++a) sW1T[(k4 + a) * LD2 + j] = (unsigned short)w[a];
    }
    for (int e = tid; e < H * (K / 4); e += 256) {
        const int R = e / (K / 4), n4 = (e - R * (K / 4)) * 4, j = mlp_unit_of_row(R);
        f4 w;
#pragma unroll
        for (int a = 0; a < 4; ++a) w[a] = W2[(long)(n4 + a) * H + j];
        *reinterpret_cast<s4*>(&sW2T[R * LD1 + n4]) = pack_bf16(w);
    }
    for (int e = tid; e < H; e += 256) sB1[e] = b1 ? b1[mlp_unit_of_row(e)] : 0.f;
    f4 lw[KC], lb[KC], gq[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        lw[c] = ld4(ln_w + 16 * c + 4 * q); lb[c] = ld4(ln_b + 16 * c + 4 * q);
        gq[c] = g2 ? ld4(g2 + 16 * c + 4 * q) : f4{1.f, 1.f, 1.f, 1.f};
    }
    int lr[NP], c4[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int idx = 64 * p + lane;
        lr[p] = idx / F4R; c4[p] = idx - lr[p] * F4R;
    }
    float lnw[KC], agam[KC], abet[KC];
#pragma unroll
    for (int t = 0; t < KC; ++t) { lnw[t] = ln_w[16 * t + i]; agam[t] = 0.f; abet[t] = 0.f; }
    __syncthreads();
    const int stride = gridDim.x * 4;
    float* so = sOall + wave * 16 * LDO;
    struct Frag { f4 a[KC], g[KC]; };
    auto load = [&](Frag& f, int tile) {
        const long row = min((long)tile * 16 + i, (long)M - 1);
        const float* p = y + row * K + 4 * q;
        const float* pg = dz + row * K + 4 * q;
#pragma unroll
        for (int c = 0; c < KC; ++c) { f.a[c] = ld4(p + 16 * c); f.g[c] = ld4(pg + 16 * c); }
    };
    auto compute = [&](const Frag& f, int tile, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const long row0 = (long)tile * 16;
        // what the LayerNorm backward needs in accumulator / row-store layout, loaded before the MFMAs
        f4 r4[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const long row = FULL ? row0 + lr[p] : min(row0 + lr[p], (long)M - 1);
            r4[p] = ld4(dz + row * K + 4 * c4[p]);
        }
        float xi[KC][4], mean4[4], rstd4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long row = FULL ? row0 + 4 * q + r : min(row0 + 4 * q + r, (long)M - 1);
            const float2 st = *reinterpret_cast<const float2*>(stats + 2 * row);
            mean4[r] = st.x; rstd4[r] = st.y;
#pragma unroll
            for (int t = 0; t < KC; ++t) xi[t][r] = y[row * K + 16 * t + i];
        }
        // LayerNorm of row i for the recomputed fc1 (statistics of the forward pass)
        const float2 sti = *reinterpret_cast<const float2*>(stats + 2 * min(row0 + i, (long)M - 1));
        s4 nb[KC], gb[KC];
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            nb[c] = pack_bf16((f.a[c] - sti.x) * sti.y * lw[c] + lb[c]);
            gb[c] = pack_bf16(f.g[c] * gq[c]);
        }
        const bool row_ok = FULL || row0 + i < M;
        f4 acc[KC];
#pragma unroll
        for (int c = 0; c < KC; ++c) acc[c] = zero4();
        unsigned short* durow = du16 + (row0 + i) * H + 8 * q;
#pragma unroll
        for (int t = 0; t < NHT; t += 2) {
            s4 da[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f4 u = *reinterpret_cast<const f4*>(&sB1[16 * (t + h) + 4 * q]);
                f4 dh = zero4();
#pragma unroll
                for (int c = 0; c < KC; ++c) {
                    u = mfma16_bf16(*reinterpret_cast<const s4*>(&sW1[(16 * (t + h) + i) * LD1 + 16 * c + 4 * q]), nb[c], u);
                    dh = mfma16_bf16(*reinterpret_cast<const s4*>(&sW2T[(16 * (t + h) + i) * LD1 + 16 * c + 4 * q]), gb[c], dh);
                }
                da[h] = pack_bf16(dh * gelu_grad4(u));
            }
            const s8v a8 = __builtin_shufflevector(da[0], da[1], 0, 1, 2, 3, 4, 5, 6, 7);      // hidden units 16t + 8q .. + 7 of row i
            if (SDU && row_ok) *reinterpret_cast<s8v*>(durow + 16 * t) = a8;
#pragma unroll
            for (int c = 0; c < KC; ++c)
                acc[c] = mfma32_bf16(a8, *reinterpret_cast<const s8v*>(&sW1T[(16 * c + i) * LD2 + 16 * t + 8 * q]), acc[c]);
        }
        // acc[c][r] = dn[row 4q + r][col 16c + i]: LayerNorm backward in this layout (as rowstream_narrow_kernel, MODE 2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool live = FULL || row0 + 4 * q + r < M;
            float s1 = 0.f, s2 = 0.f, xh[KC], gw[KC];
#pragma unroll
            for (int t = 0; t < KC; ++t) {
                const float dn = live ? acc[t][r] : 0.f;
                xh[t] = (xi[t][r] - mean4[r]) * rstd4[r];
                gw[t] = dn * lnw[t];
                agam[t] += dn * xh[t]; abet[t] += dn;
                s1 += gw[t]; s2 += gw[t] * xh[t];
            }
            s1 = row16_sum(s1) * (1.0f / K);
            s2 = row16_sum(s2) * (1.0f / K);
#pragma unroll
            for (int t = 0; t < KC; ++t) acc[t][r] = (gw[t] - s1 - xh[t] * s2) * rstd4[r];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < KC; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) so[(4 * q + r) * LDO + 16 * c + i] = acc[c][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const f4 v = *reinterpret_cast<const f4*>(&so[lr[p] * LDO + 4 * c4[p]]) + r4[p];
            if (FULL || row0 + lr[p] < M) *reinterpret_cast<f4*>(dy + (row0 + lr[p]) * K + 4 * c4[p]) = v;
        }
    };
    const int nfull = M / 16;
    int tile = blockIdx.x * 4 + wave;
    const std::true_type full{};
    Frag f0, f1;                                              // one tile ahead (two fragment sets of y and dz)
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
#pragma unroll
    for (int t = 0; t < KC; ++t) {                            // column sums of this wave: over the 4 row groups, then one atomic
        float a = agam[t], b = abet[t];
        a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
        b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
        if (q == 0) { atomicAdd(dgamma + 16 * t + i, a); atomicAdd(dbeta + 16 * t + i, b); }
    }
}

template <int KC, int NHT>
static int launch_mlp_bwd_dgrad_fused(const float* dz, const float* y, const float* stats, const float* ln_w, const float* ln_b,
                                      const float* W1, const float* b1, const float* W2, const float* g2, float* dy, void* du16,
                                      float* dgamma, float* dbeta, int M, hipStream_t stream) {
    constexpr int Kc = 16 * KC, Hc = 16 * NHT;
    constexpr int LDS = (2 * Hc * (Kc + 8) + Kc * (Hc + 16)) * 2 + Hc * 4 + 4 * 16 * (Kc + 4) * 4;
    const int grid = min(cdiv(cdiv(M, 16), 4), 256 * 2);
    static bool attr[2] = {false, false};
    if (du16) {
        auto kern = mlp_bwd_dgrad_fused_kernel<KC, NHT, true>;
        if (!attr[1]) { hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr[1] = true; }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, stream, dz, y, stats, ln_w, ln_b, W1, b1, W2, g2, dy,
                           reinterpret_cast<unsigned short*>(du16), dgamma, dbeta, M);
    } else {
        auto kern = mlp_bwd_dgrad_fused_kernel<KC, NHT, false>;
        if (!attr[0]) { hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr[0] = true; }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, stream, dz, y, stats, ln_w, ln_b, W1, b1, W2, g2, dy, nullptr, dgamma, dbeta, M);
    }
    return leod_launch_status();
}

// dy[M,K] = dz + LN-backward(((dz * g2) W2 * gelu'(LN(y) W1^T + b1)) W1), du16 (optional) [M,H] bf16 = the gradient of the hidden
// pre-activation, dgamma / dbeta [K] += the LayerNorm weight / bias gradients.  stats [M,2]: the (mean, rstd) the forward pass saved.
// Same coverage as leod_mlp_fwd_fused; LEOD_ERR_UNSUPPORTED otherwise (callers run leod_linear_dgrad_gelu16 + leod_linear_dgrad_lnbwd).
LEOD_API int leod_mlp_bwd_dgrad_fused(const float* dz, const float* y, const float* stats, const float* ln_w, const float* ln_b,
                                      const float* W1, const float* b1, const float* W2, const float* g2, float* dy, void* du16,
                                      float* dgamma, float* dbeta, int M, int H, int K, hipStream_t stream) {
    if (!dz || !y || !stats || !ln_w || !ln_b || !W1 || !W2 || !dy || !dgamma || !dbeta) return LEOD_ERR_ARG;
    if (leod_precision() != 1 || !((K == 48 && H == 192) || (K == 64 && H == 256)) || M < 16384) return LEOD_ERR_UNSUPPORTED;
    if (K == 64) return launch_mlp_bwd_dgrad_fused<4, 16>(dz, y, stats, ln_w, ln_b, W1, b1, W2, g2, dy, du16, dgamma, dbeta, M, stream);   // RVT-B stage 1
    return launch_mlp_bwd_dgrad_fused<3, 12>(dz, y, stats, ln_w, ln_b, W1, b1, W2, g2, dy, du16, dgamma, dbeta, M, stream);
}

