// ConvLSTM recurrence of a whole sequence in ONE launch per direction (DWSConvLSTM2d with dws_conv = False,
// models/layers/rnn.py:37-70, unrolled over the L timesteps of modules/detection.py:188-226).
//
// With a 1 x 1 cell convolution the recurrence of a pixel depends on no other pixel, so a workgroup can carry its 16 rows
// (pixels) through all T timesteps: the per-timestep schedule paid 3 dependent launches per timestep and stage (cell forward; gate
// backward + dgrad), each too small to fill 256 CUs -- 5 ms of a 28 ms RVT-S training step.  Here
//   * one workgroup = one 16-row tile, one WAVE per 16 channels (C / 16 waves): the wave owns the four gate columns (f, i, o, g)
//     of its channels, i.e. all of a channel's gate arithmetic is lane-local in the MFMA accumulator layout;
//   * the wave's slice of the weights stays in REGISTERS as MFMA B fragments for the whole sequence (forward: W[4 gates x 16
//     channels][K]; backward: W_h[4C][16 channels]) -- nothing but activations is read inside the time loop;
//   * the A operand of timestep t ([x_t | h_{t-1}] forward, dgates_t backward) is exchanged between the waves through a
//     double-buffered LDS tile: ONE workgroup barrier per timestep;
//   * c_t lives in registers (fp32), h_t / c_t / gates_t are written for the backward pass as the per-timestep kernels did.
// Forward variants: FX = true contracts [x_t | h_{t-1}] (K = 2C); FX = false takes the time-batched x projection gx =
// x W_x^T + b (one large GEMM over all T * M rows, computed by the caller) as accumulator input and contracts h_{t-1} only.
// BF = true: bf16 operands (A tile and B fragments rounded to bf16, v_mfma_f32_16x16x16_bf16), fp32 accumulation and state --
// the same rounding points as the per-timestep bf16 kernels.  BF = false: v_mfma_f32_16x16x4_f32, with FX the same fmaf chain
// as the per-timestep fp32 kernel.
#include <stdlib.h>
#include <type_traits>

#include "common.hpp"

template <bool FAST> __device__ __forceinline__ float tanh_(float x) {
    if constexpr (FAST) return 2.0f * sigmoidf_(2.0f * x) - 1.0f;         // v_exp / v_rcp: |error| ~1e-7, bf16 mode only
    else return tanhf(x);
}

typedef unsigned u4_ __attribute__((ext_vector_type(4)));
// G16 (precision mode bf16): the post-activation gates -- read back only by the backward kernel of the SAME lane mapping -- are kept
// as fp16 in an opaque lane-linear layout, 16 halfs (gate-major: f, i, o, g x the lane's four rows) = two 16-byte accesses per lane
// and channel group, fully coalesced (the [M][4][C] fp32 layout cost sixteen scattered 4-byte stores per lane and timestep); the gate
// gradients go out as the bf16 [M][4C] rows the LDS exchange tile already holds (their consumers, dx = dgates W_x and the weight
// gradient, feed them to bf16 MFMAs as they are), copied out with 16-byte stores.  Stage 1 of RVT-S: 1.16 -> 0.83 GB forward,
// 1.65 -> 0.99 GB backward, and half the bytes in the two GEMMs behind them.
__device__ __forceinline__ void gates16_store(u4_* dst, const float (&gv)[4][4]) {
    u2_ h[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) h[g] = __builtin_bit_cast(u2_, pack_h16(f4{gv[g][0], gv[g][1], gv[g][2], gv[g][3]}));
    dst[0] = u4_{h[0].x, h[0].y, h[1].x, h[1].y};
    dst[1] = u4_{h[2].x, h[2].y, h[3].x, h[3].y};
}
__device__ __forceinline__ void gates16_load(const u4_* src, float (&gv)[4][4]) {
    const u4_ a = src[0], b = src[1];
    const u2_ h[4] = {u2_{a.x, a.y}, u2_{a.z, a.w}, u2_{b.x, b.y}, u2_{b.z, b.w}};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f4 v = unpack_h16(__builtin_bit_cast(s4, h[g]));
#pragma unroll
        for (int r = 0; r < 4; ++r) gv[g][r] = v[r];
    }
}
// dgates tile [16][KA] bf16 of the LDS exchange buffer (row stride LD) -> dg16 rows row0 .. of timestep t, 16-byte stores
template <int KA, int LD, int NTHR>
__device__ __forceinline__ void dgates16_copy(unsigned short* __restrict__ dg16, const unsigned short* tile, long t, long row0, int M, int tid) {
    constexpr int U = KA / 8;
    for (int e = tid; e < 16 * U; e += NTHR) {
        const int row = e / U, u = e - row * U;
        if (row0 + row < M)
            *reinterpret_cast<u4_*>(dg16 + ((t * M + row0 + row) * KA + 8 * u)) = *reinterpret_cast<const u4_*>(tile + row * LD + 8 * u);
    }
}

// BF (int): operand format 0 = fp32, 1 = bf16, 2 = fp16 (forward kernels of precision mode 16f)
template <int BF> struct AElem { typedef unsigned short T; };
template <> struct AElem<0> { typedef float T; };
__device__ __forceinline__ unsigned short to_bf16(float v) {
    const f2_ p = {v, 0.f};
    return (unsigned short)(__builtin_bit_cast(unsigned, __builtin_convertvector(p, bf2_)) & 0xffffu);
}
__device__ __forceinline__ unsigned short to_h16(float v) {   // |h| < 1 and LayerNorm-ed / bounded x: no saturation needed
    return __builtin_bit_cast(unsigned short, (_Float16)v);
}
template <int OF> __device__ __forceinline__ unsigned short to_op16(float v) { if constexpr (OF == 2) return to_h16(v); else return to_bf16(v); }
template <int BF> __device__ __forceinline__ typename AElem<BF>::T a_elem(float v) {
    if constexpr (BF == 2) return to_h16(v); else if constexpr (BF == 1) return to_bf16(v); else return v;
}

// One timestep's MFMAs of a wave: acc[g] += A[16 x 16*KC] . B_g   (A fragments from the LDS tile, B fragments in registers)
template <int KC, int NG, int BF, class BT>
__device__ __forceinline__ void tile_mfma(f4 (&acc)[NG], const typename AElem<BF>::T* __restrict__ arow, const BT (&b)[NG][KC]) {
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        // long contractions: keep the scheduler from hoisting all KC fragment reads (2 registers each) above the first MFMA
        if (KC > 12 && kc % 8 == 0) __builtin_amdgcn_sched_barrier(0);
        if constexpr (BF) {
            const s4 a = *reinterpret_cast<const s4*>(arow + 16 * kc);
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g] = mfma16_16<BF>(a, b[g][kc], acc[g]);
        } else {
            const f4 a = *reinterpret_cast<const f4*>(arow + 16 * kc);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = mfma16(a[j], b[g][kc][j], acc[g]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward
//   xin   FX: x_seq [T][M][C]            !FX: gx [T][M][4C] = x_t W_x^T + b (pre-activation x part, bias included)
//   hbuf, cbuf [T+1][M][C]: slot 0 = incoming state (zero_state != 0: treated as zeros and not read), slots 1..T written
//   W [4C][2C] (gate-major rows f, i, o, g; columns [x | h]), bias [4C]; gates_out [T][M][4][C] post-activation or NULL
// ---------------------------------------------------------------------------------------------------------------------
template <int C, bool FX, int BF, bool G16 = false>
__global__ __launch_bounds__(C * 4) void lstm_seq_fwd_kernel(const float* __restrict__ xin, float* __restrict__ hbuf, float* __restrict__ cbuf,
                                                              const float* __restrict__ W, const float* __restrict__ bias,
                                                              float* __restrict__ gates_out, int M, int T, int zero_state) {
    constexpr int NW = C / 16;                       // waves per workgroup
    constexpr int KA = FX ? 2 * C : C;               // contraction length = columns of the A tile
    constexpr int KC = KA / 16;
    constexpr int LD = KA + 8;                       // bf16: rows of 4 * odd dwords; fp32: stride == 8 (mod 16) dwords
    constexpr int HOFF = FX ? C : 0;                 // column of h inside the A tile
    constexpr bool PF = C < 192;                     // prefetch the next timestep's projection (register room permitting)
    typedef typename AElem<BF>::T AT;
    typedef typename std::conditional<BF != 0, s4, f4>::type BT;
    __shared__ __attribute__((aligned(16))) AT sA[2][16 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const long row0 = (long)blockIdx.x * 16;
    const int ch = 16 * wave + i;                    // this lane's channel (accumulator column)
    const long MC = (long)M * C;
    // ---- resident B fragments: B_g[k][j = i] = W[g*C + ch][koff + k] -----------------------------------------------------------
    BT bw[4][KC];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float* wr = W + (long)(g * C + ch) * (2 * C) + (FX ? 0 : C) + 4 * q;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const f4 w = ld4(wr + 16 * kc);
            if constexpr (BF != 0) bw[g][kc] = pack16_raw<BF>(w); else bw[g][kc] = w;
        }
    }
    float bg[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bg[g] = FX ? bias[g * C + ch] : 0.f;
    // accumulator-layout rows of this lane: row0 + 4q + r; loads are clamped (never stored when out of range).  Element offsets
    // inside one timestep's slab fit 32 bits (M * 4C < 2^31, checked by the launcher): per-timestep base pointers are scalar
    int oc[4]; bool rok[4];                          // row * C + ch  (h, c);  the gate / projection offset is 4 * oc - 3 * ch
#pragma unroll
    for (int r = 0; r < 4; ++r) { rok[r] = row0 + 4 * q + r < M; oc[r] = (int)(rok[r] ? row0 + 4 * q + r : (long)M - 1) * C + ch; }
    // staging slot of this lane for the x part of the A tile: (row = lane >> 2, 4 channels 16*wave + 4*(lane & 3))
    const int srow = lane >> 2, sc4 = 16 * wave + 4 * (lane & 3);
    const int sxo = (int)min(row0 + srow, (long)M - 1) * C + sc4;
    // ---- initial state --------------------------------------------------------------------------------------------------
    float cst[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        cst[r] = zero_state ? 0.f : cbuf[oc[r]];
        const float h0 = zero_state ? 0.f : hbuf[oc[r]];
        sA[0][(4 * q + r) * LD + HOFF + ch] = a_elem<BF>(h0);
    }
    f4 xs = zero4();
    float gx[4][4];
    if (FX) {
        xs = ld4(xin + sxo);
#pragma unroll
        for (int j = 0; j < 4; ++j) sA[0][srow * LD + sc4 + j] = a_elem<BF>(xs[j]);
        if (T > 1) xs = ld4(xin + MC + sxo);
    } else if (PF) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) gx[g][r] = xin[4 * oc[r] - 3 * ch + g * C];
    }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const int buf = t & 1;
        f4 acc[4];
        if (FX) {
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = zero4();
        } else if (!PF) {                             // C >= 192: no register room for a prefetched projection (3 waves per SIMD hide it)
            const float* gp = xin + (long)t * M * 4 * C;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[g][r] = gp[4 * oc[r] - 3 * ch + g * C];
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[g][r] = gx[g][r];
            if (t + 1 < T) {                          // next timestep's x projection flies under this timestep's MFMAs
                const float* gp = xin + (long)(t + 1) * M * 4 * C;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) gx[g][r] = gp[4 * oc[r] - 3 * ch + g * C];
            }
        }
        tile_mfma<KC, 4, BF, BT>(acc, &sA[buf][i * LD + 4 * q], bw);
        // ---- gates of channel ch, rows 4q + r (rnn.py:58-68) --------------------------------------------------------------------
        float* hp = hbuf + (long)(t + 1) * MC;
        float* cp = cbuf + (long)(t + 1) * MC;
        float* gp = gates_out ? gates_out + (long)t * M * 4 * C : nullptr;
        float gv[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float f = sigmoidf_(acc[0][r] + bg[0]), ig = sigmoidf_(acc[1][r] + bg[1]), o = sigmoidf_(acc[2][r] + bg[2]);
            const float g = tanh_<(BF != 0)>(acc[3][r] + bg[3]);
            const float cn = f * cst[r] + ig * g;
            const float hn = o * tanh_<(BF != 0)>(cn);
            cst[r] = cn;
            gv[0][r] = f; gv[1][r] = ig; gv[2][r] = o; gv[3][r] = g;
            sA[buf ^ 1][(4 * q + r) * LD + HOFF + ch] = a_elem<BF>(hn);
            if (rok[r]) {
                hp[oc[r]] = hn;
                if (gates_out || t + 1 == T) cp[oc[r]] = cn;       // the c history is read by the backward pass only (which needs the gates too)
                if (!G16 && gp) {
                    float* gr = gp + (4 * oc[r] - 3 * ch);
                    gr[0] = f; gr[C] = ig; gr[2 * C] = o; gr[3 * C] = g;
                }
            }
        }
        if (G16 && gates_out)
            gates16_store(reinterpret_cast<u4_*>(gates_out) + ((((long)t * gridDim.x + blockIdx.x) * NW + wave) * 64 + lane) * 2, gv);
        if (FX && t + 1 < T) {
#pragma unroll
            for (int j = 0; j < 4; ++j) sA[buf ^ 1][srow * LD + sc4 + j] = a_elem<BF>(xs[j]);
            if (t + 2 < T) xs = ld4(xin + (long)(t + 2) * MC + sxo);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward through time
//   dh_seq [T][M][C] gradient of every h_t from the layers above (or NULL), dc_last [M][C] gradient of c_T (or NULL)
//   gates [T][M][4][C], cbuf [T+1][M][C] as written by the forward pass; zero_state: c_0 = 0 and not read
//   dgates_out [T][M][4C] pre-activation gate gradients (for the weight gradient and dx = dgates W_x, one GEMM each over all T)
//   dh0 / dc0 [M][C] (optional): gradients of the incoming state
// Per timestep: gate backward (lane-local) -> dgates tile in LDS -> barrier -> dh_{t-1} += dgates_t W_h for the wave's 16 channels.
// ---------------------------------------------------------------------------------------------------------------------
template <int C, bool BF, bool G16 = false>
__global__ __launch_bounds__(C * 4) void lstm_seq_bwd_kernel(const float* __restrict__ dh_seq, const float* __restrict__ dc_last,
                                                              const float* __restrict__ gates, const float* __restrict__ cbuf,
                                                              const float* __restrict__ W, float* __restrict__ dgates_out,
                                                              float* __restrict__ dh0, float* __restrict__ dc0, int M, int T, int zero_state) {
    constexpr int NW = C / 16;
    constexpr int KA = 4 * C, KC = KA / 16, LD = KA + 8;
    typedef typename AElem<BF>::T AT;
    typedef typename std::conditional<BF, s4, f4>::type BT;
    __shared__ __attribute__((aligned(16))) AT sA[2][16 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const long row0 = (long)blockIdx.x * 16;
    const int ch = 16 * wave + i;
    const long MC = (long)M * C;
    // resident B fragments: B[k][j = i] = W_h[k][ch] = W[k][C + ch], k = 16 kc + 4 q + jj
    BT bw[1][KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        const float* wr = W + (long)(16 * kc + 4 * q) * (2 * C) + C + ch;
        f4 w; w.x = wr[0]; w.y = wr[2 * C]; w.z = wr[4 * C]; w.w = wr[6 * C];
        if constexpr (BF) bw[0][kc] = pack_bf16(w); else bw[0][kc] = w;
    }
    int oc[4]; bool rok[4];                          // row * C + ch; gate offset = 4 * oc - 3 * ch (see the forward kernel)
#pragma unroll
    for (int r = 0; r < 4; ++r) { rok[r] = row0 + 4 * q + r < M; oc[r] = (int)(rok[r] ? row0 + 4 * q + r : (long)M - 1) * C + ch; }
    float dcn[4], dhr[4];                            // dc flowing to t - 1, dh from timestep t + 1
#pragma unroll
    for (int r = 0; r < 4; ++r) { dcn[r] = dc_last ? dc_last[oc[r]] : 0.f; dhr[r] = 0.f; }
    struct In { float g[4][4], cp[4], ct[4], dh[4]; };
    auto load = [&](In& in, int t) {
        const float* gp = gates + (long)t * M * 4 * C;
        const float* c0 = cbuf + (long)t * MC;
        if constexpr (G16) gates16_load(reinterpret_cast<const u4_*>(gates) + ((((long)t * gridDim.x + blockIdx.x) * NW + wave) * 64 + lane) * 2, in.g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* gr = gp + (4 * oc[r] - 3 * ch);
            if constexpr (!G16) {
#pragma unroll
                for (int g = 0; g < 4; ++g) in.g[g][r] = gr[g * C];
            }
            in.cp[r] = (zero_state && t == 0) ? 0.f : c0[oc[r]];
            in.ct[r] = c0[MC + oc[r]];
            in.dh[r] = dh_seq ? dh_seq[(long)t * MC + oc[r]] : 0.f;
        }
    };
    // C >= 192 (12+ waves per workgroup, 3+ per SIMD): no register room for a second input set -- the co-resident waves hide the loads
    constexpr bool PF = C < 192;
    In cur, nxt;
    load(cur, T - 1);
    for (int t = T - 1; t >= 0; --t) {
        const int buf = t & 1;
        if (PF && t > 0) load(nxt, t - 1);            // flies under this timestep's arithmetic
        float* dgp = dgates_out + (long)t * M * 4 * C;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float f = cur.g[0][r], ig = cur.g[1][r], o = cur.g[2][r], g = cur.g[3][r];
            const float th = tanh_<BF>(cur.ct[r]);
            const float dhv = cur.dh[r] + dhr[r];
            const float dc = dhv * o * (1.0f - th * th) + dcn[r];
            const float d0 = dc * cur.cp[r] * f * (1.0f - f), d1 = dc * g * ig * (1.0f - ig);
            const float d2 = dhv * th * o * (1.0f - o), d3 = dc * ig * (1.0f - g * g);
            dcn[r] = dc * f;
            AT* ar = &sA[buf][(4 * q + r) * LD + ch];
            ar[0] = a_elem<BF>(d0); ar[C] = a_elem<BF>(d1); ar[2 * C] = a_elem<BF>(d2); ar[3 * C] = a_elem<BF>(d3);
            if (!G16 && rok[r]) {
                float* dr = dgp + (4 * oc[r] - 3 * ch);
                dr[0] = d0; dr[C] = d1; dr[2 * C] = d2; dr[3 * C] = d3;
            }
        }
        __syncthreads();
        if constexpr (G16 && BF) dgates16_copy<KA, LD, C * 4>(reinterpret_cast<unsigned short*>(dgates_out), reinterpret_cast<const unsigned short*>(sA[buf]), t, row0, M, tid);
        f4 acc[1] = {zero4()};
        tile_mfma<KC, 1, BF, BT>(acc, &sA[buf][i * LD + 4 * q], bw);
#pragma unroll
        for (int r = 0; r < 4; ++r) dhr[r] = acc[0][r];
        if (PF) cur = nxt;
        else if (t > 0) load(cur, t - 1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (rok[r]) {
            if (dh0) dh0[oc[r]] = dhr[r];
            if (dc0) dc0[oc[r]] = dcn[r];
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// Stage 4 (C = 384; also 256): the weight slice of a wave no longer fits its registers (4 gates x 24 chunks per 16 channels, and a
// workgroup can have at most 16 waves).  Same structure -- one workgroup per 16-row tile for all T timesteps, h exchanged through
// LDS, one barrier per timestep -- but every wave owns 32 channels (two groups of 16) and STREAMS its B fragments from a bf16
// copy of W_h packed in fragment order (512 contiguous bytes per wave load, L2-resident: 1.2 MB for C = 384), four chunks (forward) / eight (backward) ahead in
// a second register set.  Only 40 workgroups exist for RVT-S (640 rows per timestep), so the per-timestep launches it replaces
// (33 + 28 us per timestep for a few hundred kilobytes of state) were pure latency; here a timestep costs the L2 -> CU stream of
// the slice.  bf16 mode, hoisted x projection (xin = gx) only.
// ---------------------------------------------------------------------------------------------------------------------
// Fragments of v_mfma_f32_16x16x32_bf16 (8 bf16 = 16 bytes per lane: one global_load_dwordx4 is a whole B fragment; the 16-k form
// issues at the same cost per instruction, i.e. half the rate, and needed twice the load instructions):
// wpf[((w*2 + grp)*KC + kc)*4 + g][lane] = bf16 W[g*C + 32w + 16grp + i][C + 32kc + 8q .. +7]          (forward, KC = C/32)
// wpb[(w*2 + grp)*4KC + kc][lane]        = bf16 (W[32kc + 8q + j][C + 32w + 16grp + i]), j = 0..7        (backward)
// fwd_h16: the forward copy holds fp16 fragments (precision mode 16f); the backward copy is always bf16
__global__ __launch_bounds__(256) void lstm_pack_kernel(const float* __restrict__ W, s8v* __restrict__ wpf, s8v* __restrict__ wpb, int C, int fwd_h16) {
    const int KC = C / 32, NWV = C / 32;
    const int nf = NWV * 2 * KC * 4 * 64, nb = NWV * 2 * 4 * KC * 64;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < nf + nb; e += gridDim.x * 256) {
        if (e < nf) {
            const int lane = e & 63, g = (e >> 6) & 3, r = e >> 8, kc = r % KC, wg = r / KC;      // wg = w*2 + grp
            const int i = lane & 15, q = lane >> 4;
            const float* wr = W + (long)(g * C + 16 * wg + i) * (2 * C) + C + 32 * kc + 8 * q;
            const s4 lo = fwd_h16 ? pack_h16_raw(ld4(wr)) : pack_bf16(ld4(wr)), hi = fwd_h16 ? pack_h16_raw(ld4(wr + 4)) : pack_bf16(ld4(wr + 4));
            wpf[e] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        } else {
            const int o = e - nf, lane = o & 63, r = o >> 6, kc = r % (4 * KC), wg = r / (4 * KC);
            const int i = lane & 15, q = lane >> 4;
            const float* wr = W + (long)(32 * kc + 8 * q) * (2 * C) + C + 16 * wg + i;
            const f4 w0 = {wr[0], wr[2 * C], wr[4 * C], wr[6 * C]}, w1 = {wr[8 * C], wr[10 * C], wr[12 * C], wr[14 * C]};
            const s4 lo = pack_bf16(w0), hi = pack_bf16(w1);
            wpb[o] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }
}

template <int C, bool G16 = false, int OF = 1>
__global__ __launch_bounds__(C * 2) void lstm_seq_fwd_stream_kernel(const float* __restrict__ gxin, float* __restrict__ hbuf, float* __restrict__ cbuf,
                                                                     const s8v* __restrict__ wpf, float* __restrict__ gates_out, int M, int T,
                                                                     int zero_state) {
    constexpr int KC = C / 32, LD = C + 16, NB = C >= 512 ? 1 : 2, NBT = KC / NB;   // (16 waves: 128 registers per lane, one chunk per batch)          // 32-k chunks, NB per register batch; LD / 2 == 8 (mod 16) dwords: conflict-free 16-byte A reads
    static_assert(KC % NB == 0, "chunk batches");
    __shared__ __attribute__((aligned(16))) unsigned short sA[2][16 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const long row0 = (long)blockIdx.x * 16;
    const long MC = (long)M * C;
    unsigned oc[2][4]; bool rok[4];      // unsigned: zero-extended offsets address as scalar base + 32-bit lane offset (no 64-bit address pair per access)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        rok[r] = row0 + 4 * q + r < M;
        const int row = (int)(rok[r] ? row0 + 4 * q + r : (long)M - 1);
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) oc[grp][r] = (unsigned)(row * C + 32 * wave + 16 * grp + i);
    }
    float cst[2][4];
#pragma unroll
    for (int grp = 0; grp < 2; ++grp)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            cst[grp][r] = zero_state ? 0.f : cbuf[oc[grp][r]];
            sA[0][(4 * q + r) * LD + 32 * wave + 16 * grp + i] = to_op16<OF>(zero_state ? 0.f : hbuf[oc[grp][r]]);
        }
    __syncthreads();
    // wave-uniform base (scalar registers) + lane: the 192 fragment loads of a timestep address as SGPR base + lane offset + immediate;
    // a per-lane 64-bit pointer made the compiler keep one address pair per load (1.5 KB of spills per lane)
    const s8v* wb0 = wpf + (long)(__builtin_amdgcn_readfirstlane(wave) * 2) * KC * 4 * 64;       // [grp][kc][g][64 lanes]
    for (int t = 0; t < T; ++t) {
        const int buf = t & 1;
        // the fragment addresses are the same every timestep: without this the compiler hoists all 192 loads out of the time loop and
        // parks the VALUES in scratch (1.5 KB per lane = a private 1.15 MB copy of the slice per workgroup)
        const s8v* wb = wb0;
        asm volatile("" : "+s"(wb));
        const float* gp = gxin + (long)t * M * 4 * C;
        float* hp = hbuf + (long)(t + 1) * MC;
        float* cp = cbuf + (long)(t + 1) * MC;
        float* go = gates_out ? gates_out + (long)t * M * 4 * C : nullptr;
        const unsigned short* arow = &sA[buf][i * LD + 8 * q];
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {                    // one channel group at a time: 4 accumulator tiles live
            const unsigned ch = 32 * wave + 16 * grp + i;
            // derived offsets (4 oc - 3 ch + g C ...) are recomputed per timestep: hoisted out of the time loop they cost ~150 registers
            unsigned ocg[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { ocg[r] = oc[grp][r]; asm volatile("" : "+v"(ocg[r])); }
            s8v bb[2][NB][4];
            auto loadb = [&](s8v (&b)[NB][4], int bt) {
#pragma unroll
                for (int k = 0; k < NB; ++k)
#pragma unroll
                    for (int g = 0; g < 4; ++g) b[k][g] = wb[((grp * KC + bt * NB + k) * 4 + g) * 64 + lane];
            };
            loadb(bb[0], 0);
            f4 acc[4];                                          // the x projection of the group's gate columns, in accumulator layout
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[g][r] = gp[4u * ocg[r] - 3u * ch + (unsigned)(g * C)];
#pragma unroll
            for (int bt = 0; bt < NBT; ++bt) {
                __builtin_amdgcn_sched_barrier(0);             // keep the scheduler from hoisting every batch's loads to the top
                if (bt + 1 < NBT) loadb(bb[(bt + 1) & 1], bt + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    const s8v a = *reinterpret_cast<const s8v*>(arow + 32 * (bt * NB + k));
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] = mfma32_16<OF>(a, bb[bt & 1][k][g], acc[g]);
                }
            }
            float gv[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float f = sigmoidf_(acc[0][r]), ig = sigmoidf_(acc[1][r]), o = sigmoidf_(acc[2][r]);
                const float g = tanh_<true>(acc[3][r]);
                const float cn = f * cst[grp][r] + ig * g;
                const float hn = o * tanh_<true>(cn);
                cst[grp][r] = cn;
                gv[0][r] = f; gv[1][r] = ig; gv[2][r] = o; gv[3][r] = g;
                sA[buf ^ 1][(4 * q + r) * LD + ch] = to_op16<OF>(hn);
                if (rok[r]) {
                    hp[ocg[r]] = hn;
                    if (go || t + 1 == T) cp[ocg[r]] = cn;           // the c history is read by the backward pass only (which needs the gates too)
                    if (!G16 && go) {
                        float* gr = go + (4u * ocg[r] - 3u * ch);
                        gr[0] = f; gr[C] = ig; gr[2 * C] = o; gr[3 * C] = g;
                    }
                }
            }
            if (G16 && gates_out)
                gates16_store(reinterpret_cast<u4_*>(gates_out) + (((((long)t * gridDim.x + blockIdx.x) * (C / 32) + wave) * 2 + grp) * 64 + lane) * 2, gv);
        }
        __syncthreads();
    }
}

template <int C, bool G16 = false>
__global__ __launch_bounds__(C * 2) void lstm_seq_bwd_stream_kernel(const float* __restrict__ dh_seq, const float* __restrict__ dc_last,
                                                                     const float* __restrict__ gates, const float* __restrict__ cbuf,
                                                                     const s8v* __restrict__ wpb, float* __restrict__ dgates_out,
                                                                     float* __restrict__ dh0, float* __restrict__ dc0, int M, int T) {
    constexpr int KA = 4 * C, KC = KA / 32, LD = KA + 16, NB = C >= 512 ? 2 : 4, NBT = KC / NB;
    static_assert(KC % NB == 0, "chunk batches");
    __shared__ __attribute__((aligned(16))) unsigned short sA[2][16 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, q = lane >> 4;
    const long row0 = (long)blockIdx.x * 16;
    const long MC = (long)M * C;
    unsigned oc[2][4]; bool rok[4];      // unsigned: zero-extended offsets address as scalar base + 32-bit lane offset (no 64-bit address pair per access)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        rok[r] = row0 + 4 * q + r < M;
        const int row = (int)(rok[r] ? row0 + 4 * q + r : (long)M - 1);
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) oc[grp][r] = (unsigned)(row * C + 32 * wave + 16 * grp + i);
    }
    float dcn[2][4], dhr[2][4];
#pragma unroll
    for (int grp = 0; grp < 2; ++grp)
#pragma unroll
        for (int r = 0; r < 4; ++r) { dcn[grp][r] = dc_last ? dc_last[oc[grp][r]] : 0.f; dhr[grp][r] = 0.f; }
    const s8v* wb0 = wpb + (long)(__builtin_amdgcn_readfirstlane(wave) * 2) * KC * 64;             // [grp][kc][64 lanes], wave-uniform base
    for (int t = T - 1; t >= 0; --t) {
        const int buf = t & 1;
        const s8v* wb = wb0;
        asm volatile("" : "+s"(wb));                          // see the forward kernel: keeps the fragment loads inside the time loop
        const float* gp = gates + (long)t * M * 4 * C;
        const float* c0 = cbuf + (long)t * MC;
        float* dgp = dgates_out + (long)t * M * 4 * C;
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
            const unsigned ch = 32 * wave + 16 * grp + i;
            float gg[4][4], cpv[4], ctv[4], dhv[4];
            if constexpr (G16) gates16_load(reinterpret_cast<const u4_*>(gates) + (((((long)t * gridDim.x + blockIdx.x) * (C / 32) + wave) * 2 + grp) * 64 + lane) * 2, gg);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* gr = gp + (4u * oc[grp][r] - 3u * ch);
                if constexpr (!G16) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) gg[g][r] = gr[g * C];
                }
                cpv[r] = c0[oc[grp][r]];
                ctv[r] = c0[MC + oc[grp][r]];
                dhv[r] = dh_seq ? dh_seq[(long)t * MC + oc[grp][r]] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float f = gg[0][r], ig = gg[1][r], o = gg[2][r], g = gg[3][r];
                const float th = tanh_<true>(ctv[r]);
                const float dh = dhv[r] + dhr[grp][r];
                const float dc = dh * o * (1.0f - th * th) + dcn[grp][r];
                const float d0 = dc * cpv[r] * f * (1.0f - f), d1 = dc * g * ig * (1.0f - ig);
                const float d2 = dh * th * o * (1.0f - o), d3 = dc * ig * (1.0f - g * g);
                dcn[grp][r] = dc * f;
                unsigned short* ar = &sA[buf][(4 * q + r) * LD + ch];
                ar[0] = to_bf16(d0); ar[C] = to_bf16(d1); ar[2 * C] = to_bf16(d2); ar[3 * C] = to_bf16(d3);
                if (!G16 && rok[r]) {
                    float* dr = dgp + (4u * oc[grp][r] - 3u * ch);
                    dr[0] = d0; dr[C] = d1; dr[2 * C] = d2; dr[3 * C] = d3;
                }
            }
        }
        __syncthreads();
        if constexpr (G16) dgates16_copy<KA, LD, C * 2>(reinterpret_cast<unsigned short*>(dgates_out), sA[buf], t, row0, M, tid);
        // dh_{t-1}[rows, own 32 channels] = dgates_t [16 x 4C] . W_h[4C x 32]: A fragments shared by the two channel groups
        f4 acc[2] = {zero4(), zero4()};
        s8v bb[2][NB][2];
        auto loadb = [&](s8v (&b)[NB][2], int bt) {
#pragma unroll
            for (int k = 0; k < NB; ++k)
#pragma unroll
                for (int grp = 0; grp < 2; ++grp) b[k][grp] = wb[(grp * KC + bt * NB + k) * 64 + lane];
        };
        const unsigned short* arow = &sA[buf][i * LD + 8 * q];
        loadb(bb[0], 0);
#pragma unroll
        for (int bt = 0; bt < NBT; ++bt) {
            __builtin_amdgcn_sched_barrier(0);
            if (bt + 1 < NBT) loadb(bb[(bt + 1) & 1], bt + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const s8v a = *reinterpret_cast<const s8v*>(arow + 32 * (bt * NB + k));
#pragma unroll
                for (int grp = 0; grp < 2; ++grp) acc[grp] = mfma32_bf16(a, bb[bt & 1][k][grp], acc[grp]);
            }
        }
#pragma unroll
        for (int grp = 0; grp < 2; ++grp)
#pragma unroll
            for (int r = 0; r < 4; ++r) dhr[grp][r] = acc[grp][r];
    }
#pragma unroll
    for (int grp = 0; grp < 2; ++grp)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (rok[r]) {
                if (dh0) dh0[oc[grp][r]] = dhr[grp][r];
                if (dc0) dc0[oc[grp][r]] = dcn[grp][r];
            }
}

// registers of the resident B fragments per lane: 4 gates x K/16 chunks x (2 | 4) dwords
static inline int fwd_bregs(int K, bool bf) { return 4 * (K / 16) * (bf ? 2 : 4); }

/* 1 = fused [x | h] contraction, 2 = hoisted x projection (xin = gx), 3 = hoisted + streamed weights (wpack from leod_convlstm_seq_pack),
 * 0 = sequence kernel not available for this C / precision */
LEOD_API int leod_convlstm_seq_mode(int C) {
    const bool bf = leod_precision() == 1;
    static const int stream_on = 1;
    if (bf && stream_on && (C == 256 || C == 384 || C == 512)) return 3;               // hoisted x projection + weights streamed from a packed bf16 copy
    // C = 192: the register-resident kernels spill (96 weight registers of the 168 a wave gets at 12 waves per workgroup: 79 / 83 spilled
    // VGPRs, tools/kernel_regs.py) -- streamed fragments (295 KB per timestep and workgroup from L2) are the faster of the two
    static const int stream192 = 1;
    if (bf && stream_on && stream192 && C == 192) return 3;
    if (C != 32 && C != 48 && C != 64 && C != 96 && C != 128 && C != 192) return 0;
    if (fwd_bregs(2 * C, bf) <= 96) return 1;                               // beyond ~100 resident registers the kernels spill
    if (fwd_bregs(C, bf) <= (C >= 192 ? 96 : 128)) return 2;
    return 0;
}

#define LSTM_FWD_CASE(CV, FXV)                                                                                                  \
    if (C == CV && fx == FXV) {                                                                                                 \
        if (bf && gates16) { LEOD_BY_OPFMT16(hipLaunchKernelGGL((lstm_seq_fwd_kernel<CV, FXV, OF, true>), grid, dim3(CV * 4), 0, stream, xin, hbuf, cbuf, W, bias, gates_out, M, T, zero_state)); } \
        else if (bf) { LEOD_BY_OPFMT16(hipLaunchKernelGGL((lstm_seq_fwd_kernel<CV, FXV, OF>), grid, dim3(CV * 4), 0, stream, xin, hbuf, cbuf, W, bias, gates_out, M, T, zero_state)); } \
        else hipLaunchKernelGGL((lstm_seq_fwd_kernel<CV, FXV, 0>), grid, dim3(CV * 4), 0, stream, xin, hbuf, cbuf, W, bias, gates_out, M, T, zero_state);  \
        return leod_launch_status();                                                                                            \
    }

// 1: the sequence kernels of this channel count keep the gates as fp16 (opaque layout, T x ceil(M / 16) * 16 x 4C halfs) and write the
// gate gradients as bf16 rows [T][M][4C] when asked to (gates16 of leod_convlstm_seq_fwd / _bwd); 0: fp32 tensors only
LEOD_API int leod_convlstm_seq_gates16_ok(int C) {
    static const int on = 1;
    if (!on || leod_precision() != 1) return 0;
    const int mode = leod_convlstm_seq_mode(C);
    if (mode == 3) return 1;
    return mode != 0 && (4 * C / 16) * 2 <= (C >= 192 ? 96 : 128);         // the backward sequence kernel must exist as well
}

// bytes of the packed bf16 weight copy mode 3 needs (0 otherwise)
LEOD_API long leod_convlstm_seq_pack_bytes(int C) { return leod_convlstm_seq_mode(C) == 3 ? (long)2 * 4 * C * C * 2 : 0; }

// wpack <- the two fragment-ordered bf16 copies of W_h (once per step: forward and backward of the same weights share it)
LEOD_API int leod_convlstm_seq_pack(const float* W, void* wpack, int C, hipStream_t stream) {
    if (!W || !wpack || leod_convlstm_seq_mode(C) != 3) return LEOD_ERR_ARG;
    s8v* wpf = reinterpret_cast<s8v*>(wpack);                    // C * C / 2 sixteen-byte fragments each for the forward and the backward copy
    hipLaunchKernelGGL(lstm_pack_kernel, dim3(cdiv((long)C * C, 256)), dim3(256), 0, stream, W, wpf, wpf + (long)C * C / 2, C,
                       leod_precision_mode() == 2 ? 1 : 0);
    return leod_launch_status();
}

LEOD_API int leod_convlstm_seq_fwd(const float* xin, int x_is_projection, float* hbuf, float* cbuf, const float* W, const float* bias,
                                   float* gates_out, const void* wpack, int M, int C, int T, int zero_state, int gates16, hipStream_t stream) {
    LeodFwdScope fwd_scope;
    if (gates16 && !leod_convlstm_seq_gates16_ok(C)) return LEOD_ERR_ARG;
    if (!xin || !hbuf || !cbuf || !W || !bias || M <= 0 || T <= 0 || (long)M * 4 * C >= (1L << 31)) return LEOD_ERR_ARG;
    const int mode = leod_convlstm_seq_mode(C);
    if (mode == 0 || (mode == 1) != (x_is_projection == 0)) return LEOD_ERR_UNSUPPORTED;
    if (mode == 3) {
        if (!wpack) return LEOD_ERR_ARG;
        const s8v* wpf = reinterpret_cast<const s8v*>(wpack);
        const dim3 g3(cdiv(M, 16));
        LEOD_BY_OPFMT16({
            if (C == 512 && gates16) hipLaunchKernelGGL((lstm_seq_fwd_stream_kernel<512, true, OF>), g3, dim3(1024), 0, stream, xin, hbuf, cbuf, wpf, gates_out, M, T, zero_state);
            else if (C == 512) hipLaunchKernelGGL((lstm_seq_fwd_stream_kernel<512, false, OF>), g3, dim3(1024), 0, stream, xin, hbuf, cbuf, wpf, gates_out, M, T, zero_state);
            else if (C == 384 && gates16) hipLaunchKernelGGL((lstm_seq_fwd_stream_kernel<384, true, OF>), g3, dim3(768), 0, stream, xin, hbuf, cbuf, wpf, gates_out, M, T, zero_state);
            else if (C == 384) hipLaunchKernelGGL((lstm_seq_fwd_stream_kernel<384, false, OF>), g3, dim3(768), 0, stream, xin, hbuf, cbuf, wpf, gates_out, M, T, zero_state);
            else if (C == 192 && gates16) hipLaunchKernelGGL((lstm_seq_fwd_stream_kernel<192, true, OF>), g3, dim3(384), 0, stream, xin, hbuf, cbuf, wpf, gates_out, M, T, zero_state);
            else if (C == 192) hipLaunchKernelGGL((lstm_seq_fwd_stream_kernel<192, false, OF>), g3, dim3(384), 0, stream, xin, hbuf, cbuf, wpf, gates_out, M, T, zero_state);
            else if (gates16) hipLaunchKernelGGL((lstm_seq_fwd_stream_kernel<256, true, OF>), g3, dim3(512), 0, stream, xin, hbuf, cbuf, wpf, gates_out, M, T, zero_state);
            else hipLaunchKernelGGL((lstm_seq_fwd_stream_kernel<256, false, OF>), g3, dim3(512), 0, stream, xin, hbuf, cbuf, wpf, gates_out, M, T, zero_state);
        });
        return leod_launch_status();
    }
    const bool bf = leod_precision() == 1, fx = mode == 1;
    const dim3 grid(cdiv(M, 16));
    LSTM_FWD_CASE(32, true) LSTM_FWD_CASE(48, true) LSTM_FWD_CASE(64, true) LSTM_FWD_CASE(96, true)
    LSTM_FWD_CASE(64, false) LSTM_FWD_CASE(96, false) LSTM_FWD_CASE(128, false) LSTM_FWD_CASE(192, false)
    return LEOD_ERR_UNSUPPORTED;
}

#define LSTM_BWD_CASE(CV)                                                                                                       \
    if (C == CV) {                                                                                                              \
        if (bf && gates16) hipLaunchKernelGGL((lstm_seq_bwd_kernel<CV, true, true>), grid, dim3(CV * 4), 0, stream, dh_seq, dc_last, gates, cbuf, W, dgates_out, dh0, dc0, M, T, zero_state); \
        else if (bf) hipLaunchKernelGGL((lstm_seq_bwd_kernel<CV, true>), grid, dim3(CV * 4), 0, stream, dh_seq, dc_last, gates, cbuf, W, dgates_out, dh0, dc0, M, T, zero_state); \
        else hipLaunchKernelGGL((lstm_seq_bwd_kernel<CV, false>), grid, dim3(CV * 4), 0, stream, dh_seq, dc_last, gates, cbuf, W, dgates_out, dh0, dc0, M, T, zero_state);  \
        return leod_launch_status();                                                                                            \
    }

LEOD_API int leod_convlstm_seq_bwd(const float* dh_seq, const float* dc_last, const float* gates, const float* cbuf, const float* W,
                                   float* dgates_out, float* dh0, float* dc0, const void* wpack, int M, int C, int T, int zero_state,
                                   int gates16, hipStream_t stream) {
    if (gates16 && !leod_convlstm_seq_gates16_ok(C)) return LEOD_ERR_ARG;
    if (!gates || !cbuf || !W || !dgates_out || M <= 0 || T <= 0 || (long)M * 4 * C >= (1L << 31)) return LEOD_ERR_ARG;
    const bool bf = leod_precision() == 1;
    if (leod_convlstm_seq_mode(C) == 3) {
        if (!wpack) return LEOD_ERR_ARG;
        const s8v* wpb = reinterpret_cast<const s8v*>(wpack) + (long)C * C / 2;
        const dim3 g3(cdiv(M, 16));
        if (C == 512 && gates16) hipLaunchKernelGGL((lstm_seq_bwd_stream_kernel<512, true>), g3, dim3(1024), 0, stream, dh_seq, dc_last, gates, cbuf, wpb, dgates_out, dh0, dc0, M, T);
        else if (C == 512) hipLaunchKernelGGL((lstm_seq_bwd_stream_kernel<512>), g3, dim3(1024), 0, stream, dh_seq, dc_last, gates, cbuf, wpb, dgates_out, dh0, dc0, M, T);
        else if (C == 384 && gates16) hipLaunchKernelGGL((lstm_seq_bwd_stream_kernel<384, true>), g3, dim3(768), 0, stream, dh_seq, dc_last, gates, cbuf, wpb, dgates_out, dh0, dc0, M, T);
        else if (C == 384) hipLaunchKernelGGL((lstm_seq_bwd_stream_kernel<384>), g3, dim3(768), 0, stream, dh_seq, dc_last, gates, cbuf, wpb, dgates_out, dh0, dc0, M, T);
        else if (C == 192 && gates16) hipLaunchKernelGGL((lstm_seq_bwd_stream_kernel<192, true>), g3, dim3(384), 0, stream, dh_seq, dc_last, gates, cbuf, wpb, dgates_out, dh0, dc0, M, T);
        else if (C == 192) hipLaunchKernelGGL((lstm_seq_bwd_stream_kernel<192>), g3, dim3(384), 0, stream, dh_seq, dc_last, gates, cbuf, wpb, dgates_out, dh0, dc0, M, T);
        else if (gates16) hipLaunchKernelGGL((lstm_seq_bwd_stream_kernel<256, true>), g3, dim3(512), 0, stream, dh_seq, dc_last, gates, cbuf, wpb, dgates_out, dh0, dc0, M, T);
        else hipLaunchKernelGGL((lstm_seq_bwd_stream_kernel<256>), g3, dim3(512), 0, stream, dh_seq, dc_last, gates, cbuf, wpb, dgates_out, dh0, dc0, M, T);
        return leod_launch_status();
    }
    // B fragments of the backward pass: 4C/16 chunks x (2 | 4) dwords
    if ((4 * C / 16) * (bf ? 2 : 4) > (C >= 192 ? 96 : 128)) return LEOD_ERR_UNSUPPORTED;
    const dim3 grid(cdiv(M, 16));
    LSTM_BWD_CASE(32) LSTM_BWD_CASE(48) LSTM_BWD_CASE(64) LSTM_BWD_CASE(96) LSTM_BWD_CASE(128) LSTM_BWD_CASE(192)
    return LEOD_ERR_UNSUPPORTED;
}
