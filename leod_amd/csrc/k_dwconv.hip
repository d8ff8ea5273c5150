// Depthwise convolution (groups == channels) on channels-last maps, forward / input gradient / weight gradient:
//   * DWConv.dconv of the YOLOX blocks (models/detection/yolox/models/network_blocks.py:57-76: depthwise k x k BaseConv -> pointwise 1 x 1
//     BaseConv), selected by `depthwise` in YOLOPAFPN (yolo_pafpn.py:37) and YOLOXHead (yolo_head.py:52);
//   * conv3x3_dws of the ConvLSTM (models/layers/rnn.py:20-30,50-55: k x k depthwise conv with bias on h, or on cat(x, h)).
// One multiply-add per tap and channel: no contraction over channels, so there is nothing for the MFMA units -- this is a streaming VALU
// kernel bound by HBM / L2 (k*k re-reads of a row are L2 hits: a thread walks the taps of ONE output pixel, neighbouring threads the
// neighbouring channels, so every access is a coalesced 16-byte load of 4 consecutive channels).  fp32 in every precision mode.
#include "common.hpp"

namespace {

struct DwGeom {
    int B, H, W, C, Ho, Wo, ks, stride, pad;
};

// y[b,ho,wo,c] = bias[c] + sum_{ky,kx} w[c,ky,kx] * x[b, ho*s+ky-p, wo*s+kx-p, c]
// EP: 0 raw (+bias); 1 raw + (sum, sumsq) column statistics for the training BatchNorm; 2 eval BatchNorm folded + SiLU
template <int EP>
__global__ void __launch_bounds__(256)
dwconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y,
                  double* __restrict__ colstats, int stat_rep, const float* __restrict__ bn_w, const float* __restrict__ bn_b,
                  const float* __restrict__ bn_rm, const float* __restrict__ bn_rv, float bn_eps, DwGeom g) {
    extern __shared__ float lds[];                     // EP == 1: [2][C] per-block partial statistics
    const int C4 = g.C >> 2, ks2 = g.ks * g.ks;
    const long items = (long)g.B * g.Ho * g.Wo * C4;
    if (EP == 1) {
        for (int i = threadIdx.x; i < 2 * g.C; i += 256) lds[i] = 0.f;
        __syncthreads();
    }
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int cg = (int)(it % C4);
        const long p = it / C4;
        const int wo = (int)(p % g.Wo), ho = (int)((p / g.Wo) % g.Ho), b = (int)(p / ((long)g.Wo * g.Ho));
        const int c = cg << 2;
        f4 acc = bias ? ld4(bias + c) : zero4();
        const int h0 = ho * g.stride - g.pad, w0 = wo * g.stride - g.pad;
        for (int ky = 0; ky < g.ks; ++ky) {
            const int hi = h0 + ky;
            if (hi < 0 || hi >= g.H) continue;
            for (int kx = 0; kx < g.ks; ++kx) {
                const int wi = w0 + kx;
                if (wi < 0 || wi >= g.W) continue;
                const f4 xv = ld4(x + (((long)b * g.H + hi) * g.W + wi) * g.C + c);
                const int t = ky * g.ks + kx;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = fmaf(w[(long)(c + j) * ks2 + t], xv[j], acc[j]);
            }
        }
        if (EP == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float sc = bn_w[c + j] * rsqrtf(bn_rv[c + j] + bn_eps);
                acc[j] = siluf_(fmaf(acc[j] - bn_rm[c + j], sc, bn_b[c + j]));
            }
        }
        *reinterpret_cast<f4*>(y + p * g.C + c) = acc;
        if (EP == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                atomicAdd(&lds[c + j], acc[j]);
                atomicAdd(&lds[g.C + c + j], acc[j] * acc[j]);
            }
        }
    }
    if (EP == 1) {
        __syncthreads();
        double* dst = colstats + (long)(blockIdx.x & (stat_rep - 1)) * 2 * g.C;
        for (int i = threadIdx.x; i < 2 * g.C; i += 256) atomicAdd(dst + i, (double)lds[i]);
    }
}

// dx[b,h,w,c] (+)= sum_{ky,kx : (h+p-ky) % s == 0, (w+p-kx) % s == 0} w[c,ky,kx] * dy[b, (h+p-ky)/s, (w+p-kx)/s, c]
__global__ void __launch_bounds__(256)
dwconv_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int accumulate, DwGeom g) {
    const int C4 = g.C >> 2, ks2 = g.ks * g.ks;
    const long items = (long)g.B * g.H * g.W * C4;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const int cg = (int)(it % C4);
        const long p = it / C4;
        const int wi = (int)(p % g.W), hi = (int)((p / g.W) % g.H), b = (int)(p / ((long)g.W * g.H));
        const int c = cg << 2;
        f4 acc = accumulate ? ld4(dx + p * g.C + c) : zero4();
        for (int ky = 0; ky < g.ks; ++ky) {
            const int hn = hi + g.pad - ky;
            if (hn < 0 || hn % g.stride) continue;
            const int ho = hn / g.stride;
            if (ho >= g.Ho) continue;
            for (int kx = 0; kx < g.ks; ++kx) {
                const int wn = wi + g.pad - kx;
                if (wn < 0 || wn % g.stride) continue;
                const int wo = wn / g.stride;
                if (wo >= g.Wo) continue;
                const f4 dv = ld4(dy + (((long)b * g.Ho + ho) * g.Wo + wo) * g.C + c);
                const int t = ky * g.ks + kx;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = fmaf(w[(long)(c + j) * ks2 + t], dv[j], acc[j]);
            }
        }
        *reinterpret_cast<f4*>(dx + p * g.C + c) = acc;
    }
}

// dw[c,ky,kx] += sum_{b,ho,wo} dy[b,ho,wo,c] * x[b, ho*s+ky-p, wo*s+kx-p, c];   dbias[c] += sum dy
// grid (pixel chunks, taps, 256-channel tiles); block (64 channel groups, 4 pixel lanes): a thread keeps ONE f4 of partial sums for its
// tap, the 4 pixel lanes are added through LDS, one fp32 atomic per (channel, tap) and block.  The dy rows are re-read once per tap from L2.
__global__ void __launch_bounds__(256)
dwconv_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw, float* __restrict__ dbias, DwGeom g) {
    __shared__ f4 part[4][64];
    __shared__ f4 partb[4][64];
    const int ks2 = g.ks * g.ks;
    const int t = blockIdx.y, ky = t / g.ks, kx = t % g.ks;
    const int cg = blockIdx.z * 64 + threadIdx.x, c = cg << 2;
    const bool live = c < g.C;
    const bool want_b = dbias != nullptr && t == 0;
    const long M = (long)g.B * g.Ho * g.Wo;
    const long per = (M + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per, hi_p = lo + per < M ? lo + per : M;
    f4 acc = zero4(), accb = zero4();
    if (live) {
        for (long p = lo + threadIdx.y; p < hi_p; p += 4) {
            const int wo = (int)(p % g.Wo), ho = (int)((p / g.Wo) % g.Ho), b = (int)(p / ((long)g.Wo * g.Ho));
            const f4 dv = ld4(dy + p * g.C + c);
            if (want_b) accb += dv;
            const int hi = ho * g.stride - g.pad + ky, wi = wo * g.stride - g.pad + kx;
            if (hi < 0 || hi >= g.H || wi < 0 || wi >= g.W) continue;
            const f4 xv = ld4(x + (((long)b * g.H + hi) * g.W + wi) * g.C + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(dv[j], xv[j], acc[j]);
        }
    }
    part[threadIdx.y][threadIdx.x] = acc;
    partb[threadIdx.y][threadIdx.x] = accb;
    __syncthreads();
    if (threadIdx.y == 0 && live) {
        const f4 s = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(dw + (long)(c + j) * ks2 + t, s[j]);
        if (want_b) {
            const f4 sb = partb[0][threadIdx.x] + partb[1][threadIdx.x] + partb[2][threadIdx.x] + partb[3][threadIdx.x];
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(dbias + c + j, sb[j]);
        }
    }
}

static inline bool dw_geom(DwGeom& g, int B, int H, int W, int C, int ks, int stride, int pad) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || ks < 1 || ks > 15 || stride < 1 || pad < 0) return false;
    g = DwGeom{B, H, W, C, (H + 2 * pad - ks) / stride + 1, (W + 2 * pad - ks) / stride + 1, ks, stride, pad};
    return g.Ho > 0 && g.Wo > 0;
}
static inline unsigned dw_grid(long items) {
    const long blocks = (items + 255) / 256;
    return (unsigned)(blocks < 1 ? 1 : blocks > 4096 ? 4096 : blocks);
}

}  // namespace

LEOD_API int leod_dwconv_nhwc_fwd(const float* x, const float* w, const float* bias, float* y, double* colstats, int stat_rep,
                                  const float* bn_w, const float* bn_b, const float* bn_rm, const float* bn_rv, float bn_eps, int B, int H,
                                  int W, int C, int ks, int stride, int pad, hipStream_t stream) {
    DwGeom g;
    if (!x || !w || !y || !dw_geom(g, B, H, W, C, ks, stride, pad)) return LEOD_ERR_ARG;
    if (bn_w && (colstats || !bn_b || !bn_rm || !bn_rv)) return LEOD_ERR_ARG;
    if (stat_rep < 1) stat_rep = 1;
    if (colstats && (stat_rep & (stat_rep - 1))) return LEOD_ERR_ARG;
    const unsigned grid = dw_grid((long)B * g.Ho * g.Wo * (C / 4));
    if (bn_w)
        hipLaunchKernelGGL(dwconv_fwd_kernel<2>, dim3(grid), dim3(256), 0, stream, x, w, bias, y, nullptr, 1, bn_w, bn_b, bn_rm, bn_rv, bn_eps, g);
    else if (colstats) {
        if ((size_t)2 * C * sizeof(float) > 160 * 1024) return LEOD_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(dwconv_fwd_kernel<1>, dim3(grid), dim3(256), 2 * C * sizeof(float), stream, x, w, bias, y, colstats, stat_rep,
                           nullptr, nullptr, nullptr, nullptr, 0.f, g);
    } else
        hipLaunchKernelGGL(dwconv_fwd_kernel<0>, dim3(grid), dim3(256), 0, stream, x, w, bias, y, nullptr, 1, nullptr, nullptr, nullptr, nullptr, 0.f, g);
    return leod_launch_status();
}

LEOD_API int leod_dwconv_nhwc_dgrad(const float* dy, const float* w, float* dx, int accumulate, int B, int H, int W, int C, int ks,
                                    int stride, int pad, hipStream_t stream) {
    DwGeom g;
    if (!dy || !w || !dx || !dw_geom(g, B, H, W, C, ks, stride, pad)) return LEOD_ERR_ARG;
    hipLaunchKernelGGL(dwconv_dgrad_kernel, dim3(dw_grid((long)B * H * W * (C / 4))), dim3(256), 0, stream, dy, w, dx, accumulate, g);
    return leod_launch_status();
}

LEOD_API int leod_dwconv_nhwc_wgrad(const float* dy, const float* x, float* dw, float* dbias, int B, int H, int W, int C, int ks,
                                    int stride, int pad, hipStream_t stream) {
    DwGeom g;
    if (!dy || !x || !dw || !dw_geom(g, B, H, W, C, ks, stride, pad)) return LEOD_ERR_ARG;
    const long M = (long)B * g.Ho * g.Wo;
    const long chunks = (M + 63) / 64;
    hipLaunchKernelGGL(dwconv_wgrad_kernel, dim3((unsigned)(chunks < 1 ? 1 : chunks > 256 ? 256 : chunks), ks * ks, (C / 4 + 63) / 64),
                       dim3(64, 4), 0, stream, dy, x, dw, dbias, g);
    return leod_launch_status();
}
