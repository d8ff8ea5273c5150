// Row-wise normalisation / pointwise kernels of the LEOD path: LayerNorm fwd/bwd, LayerScale bwd,
// ConvLSTM gate backward, BatchNorm(+SiLU) apply fwd/bwd.  fp32, channels-last rows.
// These are pure HBM-streaming kernels: 16-byte accesses, 16 lanes per row (4 rows per wave) so that
// C = 32..512 channel rows keep most lanes busy, per-column partial sums kept in registers and
// flushed with one atomic per column per workgroup.
#include "common.hpp"

#define MAXCJ 8          // C <= 512

// lane (i = l&15, rg = l>>4) of wave w handles row = base + rg and channels 4i + 64j .. +3
template <int CJ>
struct RowIter {
    int lane, wave, i, rg;
    __device__ RowIter() { lane = threadIdx.x & 63; wave = threadIdx.x >> 6; i = lane & 15; rg = lane >> 4; }
};

// ---------------------------------------------------------------------------------------------------
// LayerNorm forward (after the downsample conv): y = (x-mean)*rstd*w + b ; stats[M,2] = (mean, rstd)
// Reference: models/layers/maxvit/maxvit.py:172-178 (timm LayerNorm, eps 1e-5)
// ---------------------------------------------------------------------------------------------------
template <int CJ>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, float* __restrict__ y,
                                                     float* __restrict__ stats, int M, int C, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, rg = lane >> 4;
    for (long base = ((long)blockIdx.x * 4 + wave) * 4; base < M; base += (long)gridDim.x * 16) {
        const long row = base + rg;
        const bool rok = row < M;
        f4 v[CJ];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const int c = 4 * i + 64 * j;
            v[j] = (rok && c < C) ? ld4(x + row * C + c) : zero4();
            sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
        const float mean = row16_sum(sum) / (float)C;
        float var = 0.f;
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const int c = 4 * i + 64 * j;
            if (c < C) { const f4 d = v[j] - mean; var += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w); }
        }
        const float rstd = rsqrtf(row16_sum(var) / (float)C + eps);
        if (!rok) continue;
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const int c = 4 * i + 64 * j;
            if (c < C) *reinterpret_cast<f4*>(y + row * C + c) = (v[j] - mean) * rstd * ld4(w + c) + ld4(b + c);
        }
        if (stats && i == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
    }
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm backward: dx = rstd*(g - mean(g) - xhat*mean(g*xhat)) (+ dres), g = dn*w ;
// dw += sum_m dn*xhat ; db += sum_m dn.   stats may be NULL (recomputed from x).
// ---------------------------------------------------------------------------------------------------
template <int CJ>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dn, const float* __restrict__ x,
                                                     const float* __restrict__ stats, const float* __restrict__ w,
                                                     const float* __restrict__ dres, float* __restrict__ dx,
                                                     float* __restrict__ dw, float* __restrict__ db, int M, int C, float eps) {
    __shared__ float red[2][4][CJ * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, rg = lane >> 4;
    f4 aw[CJ], ab[CJ];
#pragma unroll
    for (int j = 0; j < CJ; ++j) { aw[j] = zero4(); ab[j] = zero4(); }
    // every global load of an iteration (x, dn, the residual gradient, the row statistics) is issued before the first use, on clamped
    // coordinates: the loop used to fetch x, then dn, then dres behind one another -- three dependent round trips per 4 rows of a wave
    f4 wv[CJ]; bool cok[CJ]; int cc[CJ];
#pragma unroll
    for (int j = 0; j < CJ; ++j) {
        const int c = 4 * i + 64 * j;
        cok[j] = c < C; cc[j] = cok[j] ? c : 0;
        wv[j] = cok[j] ? ld4(w + c) : zero4();
    }
    for (long base = ((long)blockIdx.x * 4 + wave) * 4; base < M; base += (long)gridDim.x * 16) {
        const long row = base + rg;
        const bool rok = row < M;
        const long rc = rok ? row : (long)M - 1;
        f4 xv[CJ], gv[CJ], dv[CJ], rv[CJ];
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            xv[j] = ld4(x + rc * C + cc[j]);
            dv[j] = ld4(dn + rc * C + cc[j]);
        }
        if (dres) {
#pragma unroll
            for (int j = 0; j < CJ; ++j) rv[j] = ld4(dres + rc * C + cc[j]);
        }
        float mean = 0.f, rstd = 0.f;
        if (stats) { mean = stats[2 * rc]; rstd = stats[2 * rc + 1]; }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            if (!(rok && cok[j])) { xv[j] = zero4(); dv[j] = zero4(); }
            sum += (xv[j].x + xv[j].y) + (xv[j].z + xv[j].w);
        }
        if (!stats) {
            mean = row16_sum(sum) / (float)C;
            float var = 0.f;
#pragma unroll
            for (int j = 0; j < CJ; ++j)
                if (cok[j]) { const f4 d = xv[j] - mean; var += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w); }
            rstd = rsqrtf(row16_sum(var) / (float)C + eps);
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            if (rok && cok[j]) {
                const f4 d = dv[j];
                const f4 xh = (xv[j] - mean) * rstd;
                xv[j] = xh;
                gv[j] = d * wv[j];
                aw[j] += d * xh; ab[j] += d;
                s1 += (gv[j].x + gv[j].y) + (gv[j].z + gv[j].w);
                const f4 gx = gv[j] * xh;
                s2 += (gx.x + gx.y) + (gx.z + gx.w);
            } else { gv[j] = zero4(); xv[j] = zero4(); }
        }
        s1 = row16_sum(s1) / (float)C;
        s2 = row16_sum(s2) / (float)C;
        if (!rok) continue;
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            if (cok[j]) {
                f4 r = (gv[j] - s1 - xv[j] * s2) * rstd;
                if (dres) r += rv[j];
                *reinterpret_cast<f4*>(dx + row * C + cc[j]) = r;
            }
        }
    }
    // column sums: over rg within the wave, then over the 4 waves through LDS, then one atomic per column
#pragma unroll
    for (int j = 0; j < CJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = quad16_sum(aw[j][e]), bsum = quad16_sum(ab[j][e]);
            if (rg == 0) { red[0][wave][j * 64 + 4 * i + e] = a; red[1][wave][j * 64 + 4 * i + e] = bsum; }
        }
    __syncthreads();
    for (int c = threadIdx.x; c < CJ * 64; c += 256) {
        if (c < C) {
            atomicAdd(dw + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
            atomicAdd(db + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// LayerScale backward: dt = gamma * dz ; dgamma += sum_m dz * t      (maxvit.py:51-53, :268-269)
// ---------------------------------------------------------------------------------------------------
template <int CJ>
__global__ __launch_bounds__(256) void ls_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ t,
                                                     const float* __restrict__ gamma, float* __restrict__ dt,
                                                     float* __restrict__ dgamma, int M, int C) {
    __shared__ float red[4][CJ * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, rg = lane >> 4;
    f4 ag[CJ];
#pragma unroll
    for (int j = 0; j < CJ; ++j) ag[j] = zero4();
    for (long base = ((long)blockIdx.x * 4 + wave) * 4; base < M; base += (long)gridDim.x * 16) {
        const long row = base + rg;
        if (row >= M) continue;
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const int c = 4 * i + 64 * j;
            if (c < C) {
                const f4 d = ld4(dz + row * C + c);
                ag[j] += d * ld4(t + row * C + c);
                *reinterpret_cast<f4*>(dt + row * C + c) = d * ld4(gamma + c);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < CJ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = quad16_sum(ag[j][e]);
            if (rg == 0) red[wave][j * 64 + 4 * i + e] = a;
        }
    __syncthreads();
    for (int c = threadIdx.x; c < CJ * 64; c += 256)
        if (c < C) atomicAdd(dgamma + c, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
}

// ---------------------------------------------------------------------------------------------------
// LayerScale gradient without the stored pre-scale tensor.  For z = res + gamma * (h W^T + b) and upstream dz:
//   G = dz^T h (the UN-scaled weight gradient, from the wgrad kernel), s = colsum(dz)
//   dW = diag(gamma) G ; db = gamma * s ; dgamma[n] = sum_m dz[m,n] t[m,n] = sum_k W[n,k] G[n,k] + b[n] s[n]
// so neither t = h W^T + b (forward store) nor dt = gamma * dz (backward store) ever exist in HBM.  One wave per row n.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ls_finalize_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                          const float* __restrict__ gamma, const float* __restrict__ G,
                                                          const float* __restrict__ s, float* __restrict__ dW,
                                                          float* __restrict__ db, float* __restrict__ dgamma, int N, int K) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;                                       // wave-uniform
    const float g = gamma[n];
    const float* wr = W + (long)n * K;
    const float* gr = G + (long)n * K;
    float* dr = dW + (long)n * K;
    float dot = 0.f;
    for (int k = 4 * lane; k < K; k += 256) {
        const f4 gv = ld4(gr + k), wv = ld4(wr + k);
        dot += (gv.x * wv.x + gv.y * wv.y) + (gv.z * wv.z + gv.w * wv.w);
        *reinterpret_cast<f4*>(dr + k) = ld4(dr + k) + g * gv;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) dot += __shfl_xor(dot, o, 64);
    if (lane == 0) {
        const float sn = s[n];
        dgamma[n] += dot + (b ? b[n] * sn : 0.f);
        if (db) db[n] += g * sn;
    }
}

// ---------------------------------------------------------------------------------------------------
// ConvLSTM gate backward (models/layers/rnn.py:58-68): from dh (total grad of h_t) and dc_next to
// pre-activation gate grads [M,4,C] and dc_prev.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_gates_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ dh2,
                                                             const float* __restrict__ dc_next,
                                                             const float* __restrict__ gates, const float* __restrict__ c_prev,
                                                             const float* __restrict__ c_t, float* __restrict__ dgates,
                                                             float* __restrict__ dc_prev, long M, int C) {
    const long n4 = M * C / 4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n4; idx += (long)gridDim.x * blockDim.x) {
        const long e = idx * 4;
        const long row = e / C; const int c = (int)(e - row * C);
        const float* gp = gates + row * 4 * C + c;
        const f4 f = ld4(gp), ig = ld4(gp + C), o = ld4(gp + 2 * C), g = ld4(gp + 3 * C);
        const f4 ct = ld4(c_t + e);
        f4 th; th.x = tanhf(ct.x); th.y = tanhf(ct.y); th.z = tanhf(ct.z); th.w = tanhf(ct.w);
        f4 dhv = dh ? ld4(dh + e) : zero4();
        if (dh2) dhv += ld4(dh2 + e);          // BPTT: gradient from the next timestep on top of the one from above
        f4 dc = dhv * o * (1.0f - th * th);
        if (dc_next) dc += ld4(dc_next + e);
        const f4 cp = c_prev ? ld4(c_prev + e) : zero4();
        float* dg = dgates + row * 4 * C + c;
        *reinterpret_cast<f4*>(dg) = dc * cp * f * (1.0f - f);
        *reinterpret_cast<f4*>(dg + C) = dc * g * ig * (1.0f - ig);
        *reinterpret_cast<f4*>(dg + 2 * C) = dhv * th * o * (1.0f - o);
        *reinterpret_cast<f4*>(dg + 3 * C) = dc * ig * (1.0f - g * g);
        if (dc_prev) *reinterpret_cast<f4*>(dc_prev + e) = dc * f;
    }
}

// ---------------------------------------------------------------------------------------------------
// BatchNorm2d (training statistics) + SiLU, channels-last rows  (network_blocks.py:29-51)
//   fwd : colstats[2][N] (double sum, sumsq from the conv epilogue) -> mean / biased var ;
//         y = silu((z-mean)*rstd*w + b) ; block 0 stores save_mean/save_rstd and updates the running buffers
//   bwd1: sums[r][0][n] += sum du, sums[r][1][n] += sum du*xhat   with du = dy * silu'(u); workgroup b adds into copy r = b mod rep
//   bwd2: dz = w*rstd*(du - sums0/M - xhat*sums1/M) with sums = the fold of the rep copies ; block 0: dw += sums1, db += sums0
// ---------------------------------------------------------------------------------------------------
// One problem of a BatchNorm launch; a launch carries up to 8 (blockIdx.y): the layers of equal depth over the head levels / branches share a
// launch (round 6).  gx: workgroups (blockIdx.x) of this problem.
struct BnFwdProb {
    const float* z; const double* colstats; const float* w; const float* b; float* y; float* save_mean; float* save_rstd; float* run_mean;
    float* run_var; const double* count_dev; long M; double count; float momentum; int rep, gx;
};
struct BnFwdGroup { BnFwdProb p[8]; };
__global__ __launch_bounds__(256) void bn_silu_fwd_kernel(BnFwdGroup grp, int N, float eps) {
    const BnFwdProb& pr = grp.p[blockIdx.y];
    const int bx = blockIdx.x, gx = pr.gx;
    if (bx >= gx) return;
    const float* __restrict__ z = pr.z; const double* __restrict__ colstats = pr.colstats; const float* __restrict__ w = pr.w;
    const float* __restrict__ b = pr.b; float* __restrict__ y = pr.y; float* __restrict__ save_mean = pr.save_mean;
    float* __restrict__ save_rstd = pr.save_rstd; float* __restrict__ run_mean = pr.run_mean; float* __restrict__ run_var = pr.run_var;
    const double* __restrict__ count_dev = pr.count_dev;
    const long M = pr.M; double count = pr.count; const float momentum = pr.momentum; const int rep = pr.rep;
    // every workgroup folds the `rep` replicas of (sum, sumsq) once and keeps (mean, rstd) of all N channels in LDS: the
    // double-precision divide / sqrt runs once per channel and workgroup instead of once per element
    extern __shared__ float sstat[];                 // [2][N]: mean, rstd
    if (count_dev) count = count * count_dev[0];     // SyncBatchNorm: rows per image (host) x images over all ranks (device)
    for (int c = threadIdx.x; c < N; c += blockDim.x) {
        double s = 0.0, ss = 0.0;
        for (int r = 0; r < rep; ++r) { s += colstats[(size_t)r * 2 * N + c]; ss += colstats[(size_t)r * 2 * N + N + c]; }
        const double mean = s / count;
        const double var = fmax(ss / count - mean * mean, 0.0);
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        sstat[c] = (float)mean; sstat[N + c] = rstd;
        if (bx == 0) {
            save_mean[c] = (float)mean;
            save_rstd[c] = rstd;
            if (run_mean) {
                const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
                run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mean;
                run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
            }
        }
    }
    __syncthreads();
    const long n4 = M * N / 4;
    for (long idx = (long)bx * blockDim.x + threadIdx.x; idx < n4; idx += (long)gx * blockDim.x) {
        const long e = idx * 4; const int c = (int)(e % N);
        const f4 v = ld4(z + e), mu = *reinterpret_cast<const f4*>(sstat + c), rs = *reinterpret_cast<const f4*>(sstat + N + c);
        const f4 ww = ld4(w + c), bb = ld4(b + c);
        f4 r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = siluf_((v[k] - mu[k]) * rs[k] * ww[k] + bb[k]);
        *reinterpret_cast<f4*>(y + e) = r;
    }
}

// RPT rows per thread, all 2 * RPT 16-byte loads of a thread issued before the first use (the row loop of the previous version paid one
// memory round trip per 4 rows: 13.7 us for a 2560-row map); the workgroup's sums go to replica blockIdx.x % rep of the
// (sum du, sum du*xhat) block, so that the double atomics on one address are gridDim.x / rep deep (bn_silu_bwd_apply folds the replicas).
struct BnBwdProb {
    const float* dy; const float* z; const float* mean; const float* rstd; const float* w; const float* b; double* sums; float* dz; float* dw;
    float* db; const double* count_dev; long M, lddy; double count; int rep, gx;
};
struct BnBwdGroup { BnBwdProb p[8]; };
template <int RPT>
__global__ __launch_bounds__(256) void bn_silu_bwd_reduce_kernel(BnBwdGroup grp, int N) {
    const BnBwdProb& pr = grp.p[blockIdx.y];
    const int bx = blockIdx.x;
    if (bx >= pr.gx) return;
    const float* __restrict__ dy = pr.dy; const float* __restrict__ z = pr.z; const float* __restrict__ mean = pr.mean;
    const float* __restrict__ rstd = pr.rstd; const float* __restrict__ w = pr.w; const float* __restrict__ b = pr.b;
    double* __restrict__ sums = pr.sums; const int rep = pr.rep; const long M = pr.M, lddy = pr.lddy;
    // thread t owns 4 consecutive channels c = 4*(t % (N/4)) and rows r0 + rstep * e; partial sums are combined in
    // LDS so that each workgroup issues ONE double atomic per (channel, statistic)
    // (fixed-order fold of the row lanes' partials in double: LDS float atomics made the sums -- and with them every gradient behind
    // this BatchNorm -- vary in their last bits from run to run)
    extern __shared__ float sred[];                 // [rstep][2*N]
    const int ncg = N / 4;
    const int cg = threadIdx.x % ncg;
    const int rlane = threadIdx.x / ncg, rstep = blockDim.x / ncg;
    if (rlane < rstep) {
        const int c = 4 * cg;
        const long r0 = (long)bx * rstep * RPT + rlane;
        f4 zv[RPT], dv[RPT];
#pragma unroll
        for (int e = 0; e < RPT; ++e) {
            const long row = min(r0 + (long)e * rstep, M - 1);
            zv[e] = ld4(z + row * N + c);
            dv[e] = ld4(dy + row * lddy + c);
        }
        const f4 mu = ld4(mean + c), rs = ld4(rstd + c), ww = ld4(w + c), bb = ld4(b + c);
        f4 s0 = zero4(), s1 = zero4();
#pragma unroll
        for (int e = 0; e < RPT; ++e) {
            const f4 xh = (zv[e] - mu) * rs;
            const f4 u = xh * ww + bb;
            f4 du = dv[e];
            const bool ok = r0 + (long)e * rstep < M;
#pragma unroll
            for (int k = 0; k < 4; ++k) du[k] = ok ? du[k] * silu_grad(u[k]) : 0.f;
            s0 += du; s1 += du * xh;
        }
        float* mine = sred + (size_t)rlane * 2 * N;
        *reinterpret_cast<f4*>(mine + c) = s0;
        *reinterpret_cast<f4*>(mine + N + c) = s1;
    }
    __syncthreads();
    double* dst = sums + (size_t)(bx % rep) * 2 * N;
    for (int c = threadIdx.x; c < 2 * N; c += blockDim.x) {
        double a = 0.0;
        for (int r = 0; r < rstep; ++r) a += (double)sred[(size_t)r * 2 * N + c];
        atomicAdd(dst + c, a);
    }
}

__global__ __launch_bounds__(256) void bn_silu_bwd_apply_kernel(BnBwdGroup grp, int N) {
    const BnBwdProb& pr = grp.p[blockIdx.y];
    const int bx = blockIdx.x, gx = pr.gx;
    if (bx >= gx) return;
    const float* __restrict__ dy = pr.dy; const float* __restrict__ z = pr.z; const float* __restrict__ mean = pr.mean;
    const float* __restrict__ rstd = pr.rstd; const float* __restrict__ w = pr.w; const float* __restrict__ b = pr.b;
    const double* __restrict__ sums = pr.sums; float* __restrict__ dz = pr.dz; float* __restrict__ dw = pr.dw; float* __restrict__ db = pr.db;
    const double* __restrict__ count_dev = pr.count_dev; const int rep = pr.rep; const long M = pr.M, lddy = pr.lddy; double count = pr.count;
    // every workgroup folds the `rep` replicas once and keeps per channel (sum du / count, sum du*xhat / count, w * rstd) in LDS:
    // the double-precision divides run once per channel and workgroup instead of twice per element
    extern __shared__ float sst[];                   // [3][N]
    if (count_dev) count = count * count_dev[0];     // SyncBatchNorm: rows per image (host) x images over all ranks (device)
    for (int c = threadIdx.x; c < N; c += blockDim.x) {
        double s0 = 0.0, s1 = 0.0;
        for (int r = 0; r < rep; ++r) { s0 += sums[(size_t)r * 2 * N + c]; s1 += sums[(size_t)r * 2 * N + N + c]; }
        sst[c] = (float)(s0 / count); sst[N + c] = (float)(s1 / count); sst[2 * N + c] = w[c] * rstd[c];
        if (bx == 0 && dw) { dw[c] += (float)s1; db[c] += (float)s0; }
    }
    __syncthreads();
    const long n4 = M * N / 4;
    for (long idx = (long)bx * blockDim.x + threadIdx.x; idx < n4; idx += (long)gx * blockDim.x) {
        const long e = idx * 4; const int c = (int)(e % N);
        const f4 mu = ld4(mean + c), rs = ld4(rstd + c), ww = ld4(w + c), bb = ld4(b + c);
        const f4 xh = (ld4(z + e) - mu) * rs;
        const f4 u = xh * ww + bb;
        f4 du = ld4(lddy == N ? dy + e : dy + (e / N) * lddy + c), r;
        const f4 m0 = *reinterpret_cast<const f4*>(sst + c), m1 = *reinterpret_cast<const f4*>(sst + N + c);
        const f4 wr = *reinterpret_cast<const f4*>(sst + 2 * N + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            du[k] *= silu_grad(u[k]);
            r[k] = wr[k] * (du[k] - m0[k] - xh[k] * m1[k]);
        }
        *reinterpret_cast<f4*>(dz + e) = r;
    }
}

// ---------------------------------------------------------------------------------------------------
#define DISPATCH_CJ(C, ...)                                              \
    do {                                                                 \
        const int cj_ = ((C) + 63) / 64;                                 \
        if (cj_ <= 1) { constexpr int CJ = 1; __VA_ARGS__; }             \
        else if (cj_ <= 2) { constexpr int CJ = 2; __VA_ARGS__; }        \
        else if (cj_ <= 3) { constexpr int CJ = 3; __VA_ARGS__; }        \
        else if (cj_ <= 4) { constexpr int CJ = 4; __VA_ARGS__; }        \
        else if (cj_ <= 6) { constexpr int CJ = 6; __VA_ARGS__; }        \
        else { constexpr int CJ = 8; __VA_ARGS__; }                      \
    } while (0)

static inline int row_grid(long M) { return (int)min((long)2048, max((long)1, (M + 15) / 16)); }
static inline int flat_grid(long n) { return (int)min((long)4096, max((long)1, (n + 255) / 256)); }

LEOD_API int leod_layernorm_fwd(const float* x, const float* w, const float* b, float* y, float* stats, int M, int C,
                                float eps, hipStream_t stream) {
    if (!x || !w || !b || !y || (C & 3) || C > 64 * MAXCJ) return LEOD_ERR_ARG;
    if (M <= 0) return LEOD_OK;
    DISPATCH_CJ(C, hipLaunchKernelGGL((ln_fwd_kernel<CJ>), dim3(row_grid(M)), dim3(256), 0, stream, x, w, b, y, stats, M, C, eps));
    return leod_launch_status();
}

LEOD_API int leod_layernorm_bwd(const float* dn, const float* x, const float* stats, const float* w, const float* dres,
                                float* dx, float* dw, float* db, int M, int C, float eps, hipStream_t stream) {
    if (!dn || !x || !w || !dx || !dw || !db || (C & 3) || C > 64 * MAXCJ) return LEOD_ERR_ARG;
    if (M <= 0) return LEOD_OK;
    const int grid = min(row_grid(M), 512);
    DISPATCH_CJ(C, hipLaunchKernelGGL((ln_bwd_kernel<CJ>), dim3(grid), dim3(256), 0, stream, dn, x, stats, w, dres, dx, dw, db, M, C, eps));
    return leod_launch_status();
}

LEOD_API int leod_layerscale_bwd(const float* dz, const float* t, const float* gamma, float* dt, float* dgamma, int M,
                                 int C, hipStream_t stream) {
    if (!dz || !t || !gamma || !dt || !dgamma || (C & 3) || C > 64 * MAXCJ) return LEOD_ERR_ARG;
    if (M <= 0) return LEOD_OK;
    const int grid = min(row_grid(M), 512);
    DISPATCH_CJ(C, hipLaunchKernelGGL((ls_bwd_kernel<CJ>), dim3(grid), dim3(256), 0, stream, dz, t, gamma, dt, dgamma, M, C));
    return leod_launch_status();
}

LEOD_API int leod_layerscale_finalize(const float* W, const float* b, const float* gamma, const float* G, const float* s,
                                      float* dW, float* db, float* dgamma, int N, int K, hipStream_t stream) {
    if (!W || !gamma || !G || !s || !dW || !dgamma || (K & 3)) return LEOD_ERR_ARG;
    if (N <= 0) return LEOD_OK;
    hipLaunchKernelGGL(ls_finalize_kernel, dim3(cdiv(N, 4)), dim3(256), 0, stream, W, b, gamma, G, s, dW, db, dgamma, N, K);
    return leod_launch_status();
}

LEOD_API int leod_convlstm_gates_bwd(const float* dh, const float* dh2, const float* dc_next, const float* gates, const float* c_prev,
                                     const float* c_t, float* dgates, float* dc_prev, int M, int C, hipStream_t stream) {
    if (!gates || !c_t || !dgates || (C & 3)) return LEOD_ERR_ARG;
    if (M <= 0) return LEOD_OK;
    hipLaunchKernelGGL(lstm_gates_bwd_kernel, dim3(flat_grid((long)M * C / 4)), dim3(256), 0, stream, dh, dh2, dc_next, gates,
                       c_prev, c_t, dgates, dc_prev, (long)M, C);
    return leod_launch_status();
}

// ---- BatchNorm + SiLU launches: n <= 8 problems of ONE channel count per launch (arrays are HOST arrays of length n) ---------------------
LEOD_API int leod_bn_silu_fwd_group(int n, const float* const* z, const double* const* colstats, const int* stat_rep, const float* const* w,
                                    const float* const* b, float* const* y, float* const* save_mean, float* const* save_rstd,
                                    float* const* run_mean, float* const* run_var, const int* M, int N, const double* count,
                                    const double* const* count_dev, float eps, const float* momentum, hipStream_t stream) {
    if (n < 1 || n > 8 || !z || !colstats || !stat_rep || !w || !b || !y || !save_mean || !save_rstd || !M || !count || !momentum || (N & 3)) return LEOD_ERR_ARG;
    BnFwdGroup g{};
    int gmax = 0;
    for (int k = 0; k < n; ++k) {
        if (!z[k] || !colstats[k] || !w[k] || !b[k] || !y[k] || !save_mean[k] || !save_rstd[k]) return LEOD_ERR_ARG;
        // every workgroup folds the stat_rep copies of (sum, sumsq): <= 1024 workgroups with a grid-stride row loop (4096 workgroups re-read
        // up to 49 KB of replicas each)
        const int gx = M[k] > 0 ? min(flat_grid((long)M[k] * N / 4), 1024) : 0;
        g.p[k] = BnFwdProb{z[k], colstats[k], w[k], b[k], y[k], save_mean[k], save_rstd[k], run_mean ? run_mean[k] : nullptr,
                           run_var ? run_var[k] : nullptr, count_dev ? count_dev[k] : nullptr, (long)M[k], count[k], momentum[k],
                           stat_rep[k] > 1 ? stat_rep[k] : 1, gx};
        gmax = max(gmax, gx);
    }
    if (gmax == 0) return LEOD_OK;
    hipLaunchKernelGGL(bn_silu_fwd_kernel, dim3(gmax, n), dim3(256), 2 * N * sizeof(float), stream, g, N, eps);
    return leod_launch_status();
}
LEOD_API int leod_bn_silu_fwd(const float* z, const double* colstats, int stat_rep, const float* w, const float* b, float* y,
                              float* save_mean, float* save_rstd, float* run_mean, float* run_var, int M, int N,
                              double count, const double* count_dev, float eps, float momentum, hipStream_t stream) {
    if (!z || !colstats || !w || !b || !y || !save_mean || !save_rstd || (N & 3)) return LEOD_ERR_ARG;
    return leod_bn_silu_fwd_group(1, &z, &colstats, &stat_rep, &w, &b, &y, &save_mean, &save_rstd, &run_mean, &run_var, &M, N, &count, &count_dev,
                                  eps, &momentum, stream);
}

static int bn_bwd_fill(BnBwdGroup& g, int n, const float* const* dy, const float* const* z, const float* const* mean, const float* const* rstd,
                       const float* const* w, const float* const* b, double* const* sums, const int* rep, float* const* dz, float* const* dw,
                       float* const* db, const int* M, int N, const double* count, const double* const* count_dev, const int* lddy) {
    if (n < 1 || n > 8 || !dy || !z || !mean || !rstd || !w || !b || !sums || !rep || !M || (N & 3) || N / 4 > 256) return LEOD_ERR_ARG;
    for (int k = 0; k < n; ++k) {
        const int ld = lddy ? lddy[k] : 0;
        if (!dy[k] || !z[k] || !mean[k] || !rstd[k] || !w[k] || !b[k] || !sums[k] || (ld && (ld < N || (ld & 3)))) return LEOD_ERR_ARG;
        g.p[k] = BnBwdProb{dy[k], z[k], mean[k], rstd[k], w[k], b[k], sums[k], dz ? dz[k] : nullptr, dw ? dw[k] : nullptr, db ? db[k] : nullptr,
                           count_dev ? count_dev[k] : nullptr, (long)M[k], (long)(ld ? ld : N), count ? count[k] : 1.0, rep[k] < 1 ? 1 : rep[k], 0};
    }
    return LEOD_OK;
}
LEOD_API int leod_bn_silu_bwd_reduce_group(int n, const float* const* dy, const float* const* z, const float* const* mean,
                                           const float* const* rstd, const float* const* w, const float* const* b, double* const* sums,
                                           const int* rep, const int* M, int N, const int* lddy, hipStream_t stream) {
    BnBwdGroup g{};
    const int rc = bn_bwd_fill(g, n, dy, z, mean, rstd, w, b, sums, rep, nullptr, nullptr, nullptr, M, N, nullptr, nullptr, lddy);
    if (rc != LEOD_OK) return rc;
    const int rstep = 256 / (N / 4);
    // 8 rows per thread (4 when that leaves every problem fewer than 128 workgroups); the same-address double atomics at the end of a workgroup
    // are spread over `rep` replicas (1024 workgroups x 192 atomics on ONE copy took 31 us for the 40960 x 96 maps, tools/kbench_bn.py)
    int rpt = 4;
    for (int k = 0; k < n; ++k) if (((long)M[k] + rstep * 8 - 1) / (rstep * 8) >= 128) rpt = 8;
    int gmax = 0;
    for (int k = 0; k < n; ++k) { g.p[k].gx = M[k] > 0 ? (int)(((long)M[k] + rstep * rpt - 1) / (rstep * rpt)) : 0; gmax = max(gmax, g.p[k].gx); }
    if (gmax == 0) return LEOD_OK;
    const size_t lds = (size_t)rstep * 2 * N * sizeof(float);
    if (rpt == 8) hipLaunchKernelGGL(bn_silu_bwd_reduce_kernel<8>, dim3(gmax, n), dim3(256), lds, stream, g, N);
    else hipLaunchKernelGGL(bn_silu_bwd_reduce_kernel<4>, dim3(gmax, n), dim3(256), lds, stream, g, N);
    return leod_launch_status();
}
LEOD_API int leod_bn_silu_bwd_reduce(const float* dy, const float* z, const float* mean, const float* rstd, const float* w,
                                     const float* b, double* sums, int rep, int M, int N, int lddy, hipStream_t stream) {
    return leod_bn_silu_bwd_reduce_group(1, &dy, &z, &mean, &rstd, &w, &b, &sums, &rep, &M, N, &lddy, stream);
}

LEOD_API int leod_bn_silu_bwd_apply_group(int n, const float* const* dy, const float* const* z, const float* const* mean,
                                          const float* const* rstd, const float* const* w, const float* const* b, double* const* sums,
                                          const int* rep, float* const* dz, float* const* dw, float* const* db, const int* M, int N,
                                          const double* count, const double* const* count_dev, const int* lddy, hipStream_t stream) {
    if (!dz || !count) return LEOD_ERR_ARG;
    BnBwdGroup g{};
    const int rc = bn_bwd_fill(g, n, dy, z, mean, rstd, w, b, sums, rep, dz, dw, db, M, N, count, count_dev, lddy);
    if (rc != LEOD_OK) return rc;
    int gmax = 0;
    for (int k = 0; k < n; ++k) {
        if (!dz[k]) return LEOD_ERR_ARG;
        g.p[k].gx = M[k] > 0 ? (int)min((long)1024, max((long)1, ((long)M[k] * N / 4 + 255) / 256)) : 0;
        gmax = max(gmax, g.p[k].gx);
    }
    if (gmax == 0) return LEOD_OK;
    hipLaunchKernelGGL(bn_silu_bwd_apply_kernel, dim3(gmax, n), dim3(256), 3 * N * sizeof(float), stream, g, N);
    return leod_launch_status();
}
LEOD_API int leod_bn_silu_bwd_apply(const float* dy, const float* z, const float* mean, const float* rstd, const float* w,
                                    const float* b, const double* sums, int rep, float* dz, float* dw, float* db, int M, int N,
                                    double count, const double* count_dev, int lddy, hipStream_t stream) {
    double* s_ = const_cast<double*>(sums);
    return leod_bn_silu_bwd_apply_group(1, &dy, &z, &mean, &rstd, &w, &b, &s_, &rep, &dz, &dw, &db, &M, N, &count, &count_dev, &lddy, stream);
}
