// Optimiser and input-side kernels of the LEOD path: fused value-clip + AdamW over the flat
// parameter buffer (modules/detection.py:485-518, train.py:236-237) and the stacked-histogram event
// voxelisation (data/utils/representations.py:78-123).
#include "common.hpp"
#pragma clang fp contract(off)

// torch.optim.AdamW single-tensor update order: p *= 1 - lr*wd ; m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ;
// denom = sqrt(v)/sqrt(bc2) + eps ; p -= (lr/bc1) * m / denom.  Gradients are clipped BY VALUE first.
__global__ __launch_bounds__(256) void adamw_clip_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, long n, float lr, float beta1, float beta2,
                                                         float eps, float wd, float bc1, float bc2_sqrt, float clip,
                                                         float grad_scale, const float* __restrict__ hp) {
    if (hp) { lr = hp[0]; bc1 = hp[1]; bc2_sqrt = hp[2]; grad_scale = hp[3]; }   // per-step scalars from device memory (hipGraph replay)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gi = g[i] * grad_scale;
        if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
        float pi = p[i] * (1.f - lr * wd);
        // exp_avg.lerp_(grad, 1 - beta1) ; exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        const float mi = m[i] + (gi - m[i]) * (1.f - beta1);
        const float vi = v[i] * beta2 + (1.f - beta2) * (gi * gi);
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi = pi - (lr / bc1) * (mi / denom);
        p[i] = pi; m[i] = mi; v[i] = vi; g[i] = gi;
    }
}

static void shadow_mark_stale(const float* p, long n);
LEOD_API int leod_adamw_clip_step(float* p, float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                                  float eps, float weight_decay, int step, float clip_value, float grad_scale,
                                  const float* hp_dev, hipStream_t stream) {
    if (!p || !g || !m || !v || (step < 1 && !hp_dev)) return LEOD_ERR_ARG;
    if (n == 0) return LEOD_OK;
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2 = (float)(1.0 - pow((double)beta2, (double)step));
    const int grid = (int)min((long)2048, (n + 255) / 256);
    hipLaunchKernelGGL(adamw_clip_kernel, dim3(grid), dim3(256), 0, stream, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                       bc1, sqrtf(bc2), clip_value, grad_scale, hp_dev);
    shadow_mark_stale(p, n);                          // bf16 shadows of these parameters (below) are out of date
    return leod_launch_status();
}

// ---- bf16 shadow of the flat parameter buffer ------------------------------------------------------------------------------------
// In precision mode bf16 every GEMM rounds its fp32 weight tile to bf16 while staging it; the stage 2-4 Linear launches are bound by
// the L2 -> CU traffic of exactly those tiles (5 040 workgroups x 98 KB of fp32 weights for one stage-4 dgrad).  A caller that owns a
// flat parameter buffer registers a bf16 buffer of the same length; leod_weight_shadow_refresh() rounds the parameters into it (the
// same pack_bf16 as the loaders: results are bit-identical) and marks it fresh; leod_adamw_clip_step() on a registered buffer and
// leod_weight_shadow_invalidate() mark it stale, and stale shadows are not used (the loaders read the fp32 weights as before).
// Recording a step for replay (stream capture): the caller records a forced refresh as the first launch of the step and pins the
// shadows (leod_weight_shadow_pin) while it records, so that the recorded GEMMs read them; a replayed step then always starts from
// fresh shadows, whatever happened to the parameters in between.
#include <mutex>
#include <vector>
namespace {
struct ShadowEntry { const float* base; long n; unsigned short* sh; bool fresh; unsigned short* shf; };   // shf: fp16 copy (mode 16f) | NULL
std::mutex g_shadow_mu;
std::vector<ShadowEntry> g_shadows;
bool g_shadow_pinned = false;
}
const unsigned short* leod_shadow_of(const float* w, int of) {
    if (!w) return nullptr;
    std::lock_guard<std::mutex> lock(g_shadow_mu);
    for (const ShadowEntry& e : g_shadows)
        if ((e.fresh || g_shadow_pinned) && w >= e.base && w < e.base + e.n) {
            if (of == 2) return e.shf ? e.shf + (w - e.base) : nullptr;
            return e.sh + (w - e.base);
        }
    return nullptr;
}
// one pass over the parameters writes the bf16 copy and (mode 16f: shf != NULL) the fp16 copy the forward GEMMs read
__global__ __launch_bounds__(256) void weight_shadow_kernel(const float* __restrict__ p, unsigned short* __restrict__ sh,
                                                            unsigned short* __restrict__ shf, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f4 v = ld4(p + 4 * i);
        *reinterpret_cast<s4*>(sh + 4 * i) = pack_bf16(v);
        if (shf) *reinterpret_cast<s4*>(shf + 4 * i) = pack_h16_sat(v);
    }
}
// shadow16 == NULL (or n <= 0) withdraws the registration of `base`.  n % 4 == 0, base 16-byte and shadow16 8-byte aligned.
LEOD_API int leod_set_weight_shadow(const float* base, long n, void* shadow16) {
    if (!base) return LEOD_ERR_ARG;
    std::lock_guard<std::mutex> lock(g_shadow_mu);
    for (size_t k = 0; k < g_shadows.size(); ++k)
        if (g_shadows[k].base == base) { g_shadows.erase(g_shadows.begin() + k); break; }
    if (!shadow16 || n <= 0) return LEOD_OK;
    if ((n & 3) || (reinterpret_cast<uintptr_t>(base) & 15) || (reinterpret_cast<uintptr_t>(shadow16) & 7)) return LEOD_ERR_ARG;
    g_shadows.push_back(ShadowEntry{base, n, reinterpret_cast<unsigned short*>(shadow16), false, nullptr});
    return LEOD_OK;
}
// fp16 copy (n values, caller-owned, 8-byte aligned) of a buffer registered above: written by the refresh in precision mode 16f and read
// by the forward GEMMs of that mode; NULL withdraws it.  LEOD_ERR_ARG if `base` is not registered.
LEOD_API int leod_set_weight_shadow_f16(const float* base, void* shadow_f16) {
    if (!base || (reinterpret_cast<uintptr_t>(shadow_f16) & 7)) return LEOD_ERR_ARG;
    std::lock_guard<std::mutex> lock(g_shadow_mu);
    for (ShadowEntry& e : g_shadows)
        if (e.base == base) { e.shf = reinterpret_cast<unsigned short*>(shadow_f16); e.fresh = false; return LEOD_OK; }
    return LEOD_ERR_ARG;
}
LEOD_API int leod_weight_shadow_invalidate() {
    std::lock_guard<std::mutex> lock(g_shadow_mu);
    for (ShadowEntry& e : g_shadows) e.fresh = false;
    return LEOD_OK;
}
LEOD_API int leod_weight_shadow_pin(int on) {
    std::lock_guard<std::mutex> lock(g_shadow_mu);
    g_shadow_pinned = on != 0;
    return LEOD_OK;
}
// Rounds every registered buffer whose shadow is stale; force != 0: every buffer, and the freshness flags are left alone (for
// launches that are being RECORDED, not executed).  Returns the number of launches (>= 0) or an error code.
LEOD_API int leod_weight_shadow_refresh(int force, hipStream_t stream) {
    if (leod_precision() != 1) return 0;
    std::lock_guard<std::mutex> lock(g_shadow_mu);
    int launches = 0;
    for (ShadowEntry& e : g_shadows) {
        if (e.fresh && !force) continue;
        const long n4 = e.n / 4;
        hipLaunchKernelGGL(weight_shadow_kernel, dim3((unsigned)min((long)2048, (n4 + 255) / 256)), dim3(256), 0, stream, e.base, e.sh,
                           leod_precision_mode() == 2 ? e.shf : nullptr, n4);
        if (leod_launch_status() != LEOD_OK) return LEOD_ERR_LAUNCH;
        if (!force) e.fresh = true;
        ++launches;
    }
    return launches;
}
static void shadow_mark_stale(const float* p, long n) {
    std::lock_guard<std::mutex> lock(g_shadow_mu);
    for (ShadowEntry& e : g_shadows)
        if (p < e.base + e.n && e.base < p + n) e.fresh = false;
}

// ---- stacked histogram ---------------------------------------------------------------------------
// counts[2*bins*H*W] (int32, zeroed by the caller) <- events ; then uint8 = clamp(wrap(counts), 0, cutoff)
__global__ __launch_bounds__(256) void voxel_count_kernel(const long* __restrict__ x, const long* __restrict__ y,
                                                          const long* __restrict__ pol, const long* __restrict__ t,
                                                          int* __restrict__ counts, long n, int bins, int H, int W) {
    const long t0 = t[0], t1 = t[n - 1];
    const float denom = (float)max(t1 - t0, 1L);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        // reference: (time - t0) [int64] / max(t1-t0, 1) -> true division in fp32, * bins, floor, clamp
        float tn = (float)(t[i] - t0) / denom;
        tn = tn * (float)bins;
        long ti = (long)floorf(tn);
        if (ti > bins - 1) ti = bins - 1;
        const long idx = x[i] + (long)W * y[i] + (long)H * W * ti + (long)bins * H * W * pol[i];
        atomicAdd(counts + idx, 1);
    }
}
__global__ __launch_bounds__(256) void voxel_finalize_kernel(const int* __restrict__ counts, unsigned char* __restrict__ out,
                                                             long n, int cutoff, int fastmode) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int c = counts[i];
        if (fastmode) c &= 255;                                  // uint8 accumulation wraps
        else { c = (int)(short)(c & 0xffff); if (c < 0) c = 0; }  // int16 accumulation, clamp(min=0)
        out[i] = (unsigned char)min(c, cutoff);
    }
}

LEOD_API int leod_voxelize_u8(const long* x, const long* y, const long* pol, const long* t, long n_events, int* counts_ws,
                              unsigned char* out, int bins, int H, int W, int count_cutoff, int fastmode, hipStream_t stream) {
    if (!counts_ws || !out || bins < 1) return LEOD_ERR_ARG;
    const long n = 2L * bins * H * W;
    if (hipMemsetAsync(counts_ws, 0, n * sizeof(int), stream) != hipSuccess) return LEOD_ERR_LAUNCH;
    if (n_events > 0) {
        if (!x || !y || !pol || !t) return LEOD_ERR_ARG;
        hipLaunchKernelGGL(voxel_count_kernel, dim3((int)min((long)2048, (n_events + 255) / 256)), dim3(256), 0, stream, x, y, pol,
                           t, counts_ws, n_events, bins, H, W);
    }
    const int cutoff = count_cutoff <= 0 ? 255 : min(count_cutoff, 255);
    hipLaunchKernelGGL(voxel_finalize_kernel, dim3((int)min((long)2048, (n + 255) / 256)), dim3(256), 0, stream, counts_ws, out, n,
                       cutoff, fastmode);
    return leod_launch_status();
}

// ---- mixed-density event stack (data/utils/representations.py:125-221) --------------------------------
// counts[bins*H*W] (int32, zeroed) += 2*pol-1 at bin floor(max(bins - log(t_norm)/log(1/2), 0)); then per pixel the running sum over the
// bins, each stage wrapped to int8 (put_(accumulate) and the assignment of the int64 channel sums both happen in int8), clamped to +-cutoff
__global__ __launch_bounds__(256) void mixed_density_count_kernel(const long* __restrict__ x, const long* __restrict__ y,
                                                                  const long* __restrict__ pol, const long* __restrict__ t,
                                                                  int* __restrict__ counts, long n, int bins, int H, int W) {
    const long t0 = t[0], t1 = t[n - 1];
    const float denom = (float)max(t1 - t0, 1L);
    const float ln_half = (float)-0.6931471805599453;            // math.log(1/2) as the fp32 divisor ATen makes of the python scalar
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float tn = (float)(t[i] - t0) / denom;
        tn = fminf(fmaxf(tn, 1e-6f), 1.f - 1e-6f);
        float b = (float)bins - __fdiv_rn(logf(tn), ln_half);
        b = floorf(fmaxf(b, 0.f));
        const long idx = x[i] + (long)W * y[i] + (long)H * W * (long)b;
        atomicAdd(counts + idx, pol[i] ? 1 : -1);
    }
}
__global__ __launch_bounds__(256) void mixed_density_finalize_kernel(const int* __restrict__ counts, signed char* __restrict__ out, long hw,
                                                                     int bins, int cutoff) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (long)gridDim.x * blockDim.x) {
        int run = 0;
        for (int b = 0; b < bins; ++b) {
            run += counts[(long)b * hw + i];
            int v = (int)(signed char)(run & 255);
            if (cutoff >= 0) v = max(-cutoff, min(cutoff, v));
            out[(long)b * hw + i] = (signed char)v;
        }
    }
}

LEOD_API int leod_mixed_density_i8(const long* x, const long* y, const long* pol, const long* t, long n_events, int* counts_ws,
                                   signed char* out, int bins, int H, int W, int count_cutoff, hipStream_t stream) {
    if (!counts_ws || !out || bins < 1 || H < 1 || W < 1 || count_cutoff > 127) return LEOD_ERR_ARG;
    const long hw = (long)H * W;
    if (hipMemsetAsync(counts_ws, 0, bins * hw * sizeof(int), stream) != hipSuccess) return LEOD_ERR_LAUNCH;
    if (n_events > 0) {
        if (!x || !y || !pol || !t) return LEOD_ERR_ARG;
        hipLaunchKernelGGL(mixed_density_count_kernel, dim3((int)min((long)2048, (n_events + 255) / 256)), dim3(256), 0, stream, x, y,
                           pol, t, counts_ws, n_events, bins, H, W);
    }
    hipLaunchKernelGGL(mixed_density_finalize_kernel, dim3((int)min((long)2048, (hw + 255) / 256)), dim3(256), 0, stream, counts_ws, out, hw,
                       bins, count_cutoff);
    return leod_launch_status();
}

__global__ void set_scalars4_kernel(float* dst, float a, float b, float c, float d) { dst[0] = a; dst[1] = b; dst[2] = c; dst[3] = d; }
// dst[0..3] = (a,b,c,d): per-step scalars handed to a replayed hipGraph without touching host memory
LEOD_API int leod_set_scalars4(float* dst, float a, float b, float c, float d, hipStream_t stream) {
    if (!dst) return LEOD_ERR_ARG;
    hipLaunchKernelGGL(set_scalars4_kernel, dim3(1), dim3(1), 0, stream, dst, a, b, c, d);
    return leod_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// Recurrent-state plumbing of the time-batched step as ONE launch per call instead of one ATen kernel per tensor:
//   leod_rows_masked_zero: t_k[b, :] = 0 where mask[b], for up to 16 tensors of B rows each (RNNStates.reset: h and c of the four
//     stages, reference modules/utils/detection.py:60-75 `t[mask] = 0` per tensor);
//   leod_copy_multi: dst_k[:] = src_k[:] for up to 16 buffers (the initial (h, c) of a stage into slot 0 of its sequence buffers).
// Sizes are in 16-byte units; every pointer 16-byte aligned.
typedef unsigned u4v __attribute__((ext_vector_type(4)));
struct MultiBuf { void* dst[16]; const void* src[16]; long n16[16]; };

__global__ __launch_bounds__(256) void rows_masked_zero_kernel(MultiBuf mb, const unsigned char* __restrict__ mask, int B) {
    const int k = blockIdx.y;
    const long row16 = mb.n16[k];                              // 16-byte units per batch row
    u4v* p = reinterpret_cast<u4v*>(mb.dst[k]);
    const u4v z = {0u, 0u, 0u, 0u};
    for (int b = 0; b < B; ++b) {
        if (!mask[b]) continue;                                // block-uniform
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < row16; e += (long)gridDim.x * 256) p[b * row16 + e] = z;
    }
}

__global__ __launch_bounds__(256) void copy_multi_kernel(MultiBuf mb) {
    const int k = blockIdx.y;
    const long n = mb.n16[k];
    u4v* d = reinterpret_cast<u4v*>(mb.dst[k]);
    const u4v* s = reinterpret_cast<const u4v*>(mb.src[k]);
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) d[e] = s[e];
}

LEOD_API int leod_rows_masked_zero(void* const* tensors, const long* row_bytes, int n, const unsigned char* mask, int B, hipStream_t stream) {
    if (!tensors || !row_bytes || !mask || n < 0 || n > 16 || B < 0) return LEOD_ERR_ARG;
    if (n == 0 || B == 0) return LEOD_OK;
    MultiBuf mb{};
    long mx = 0;
    for (int k = 0; k < n; ++k) {
        if (!tensors[k] || (row_bytes[k] & 15) || ((uintptr_t)tensors[k] & 15)) return LEOD_ERR_ARG;
        mb.dst[k] = tensors[k]; mb.n16[k] = row_bytes[k] / 16;
        mx = mb.n16[k] > mx ? mb.n16[k] : mx;
    }
    const int gx = (int)((mx + 1023) / 1024 < 1 ? 1 : ((mx + 1023) / 1024 > 64 ? 64 : (mx + 1023) / 1024));
    hipLaunchKernelGGL(rows_masked_zero_kernel, dim3(gx, n), dim3(256), 0, stream, mb, mask, B);
    return leod_launch_status();
}

LEOD_API int leod_copy_multi(void* const* dst, const void* const* src, const long* nbytes, int n, hipStream_t stream) {
    if (!dst || !src || !nbytes || n < 0 || n > 16) return LEOD_ERR_ARG;
    if (n == 0) return LEOD_OK;
    MultiBuf mb{};
    long mx = 0;
    for (int k = 0; k < n; ++k) {
        if (!dst[k] || !src[k] || (nbytes[k] & 15) || ((uintptr_t)dst[k] & 15) || ((uintptr_t)src[k] & 15)) return LEOD_ERR_ARG;
        mb.dst[k] = dst[k]; mb.src[k] = src[k]; mb.n16[k] = nbytes[k] / 16;
        mx = mb.n16[k] > mx ? mb.n16[k] : mx;
    }
    const int gx = (int)((mx + 1023) / 1024 < 1 ? 1 : ((mx + 1023) / 1024 > 256 ? 256 : (mx + 1023) / 1024));
    hipLaunchKernelGGL(copy_multi_kernel, dim3(gx, n), dim3(256), 0, stream, mb);
    return leod_launch_status();
}

// hflip test-time augmentation of the pseudo-label pass (modules/pseudo_labeler.py:469-470 of the reference: ev = cat([ev, flip(ev, -1)], batch)):
// out[t, b] = frames[t][b], out[t, B + b, .., x] = frames[t][b, .., W - 1 - x] for T frame tensors [B, rows, W] of bytes, in ONE pass --
// torch ran stack (copy), flip (copy) and cat (copy of both): 2 GB of traffic for the 245 MB of a Gen1 chunk, 2.1 ms of a 26 ms chunk.
// V = bytes per thread access (16 / 4 / 1 by the divisibility of W); a flipped vector is the mirrored vector with its bytes reversed.
struct FrameTable { const unsigned char* f[32]; };
template <int V> struct FlipVec;
template <> struct FlipVec<16> { typedef u4v T; static __device__ __forceinline__ T rev(T v) { return T{__builtin_bswap32(v.w), __builtin_bswap32(v.z), __builtin_bswap32(v.y), __builtin_bswap32(v.x)}; } };
template <> struct FlipVec<4> { typedef unsigned T; static __device__ __forceinline__ T rev(T v) { return __builtin_bswap32(v); } };
template <> struct FlipVec<1> { typedef unsigned char T; static __device__ __forceinline__ T rev(T v) { return v; } };
template <int V>
__global__ __launch_bounds__(256) void stack_hflip_u8_kernel(FrameTable ft, unsigned char* __restrict__ out, long rows, int W) {
    typedef typename FlipVec<V>::T VT;
    const int WV = W / V;
    const long n = rows * WV;                                    // vectors of one frame tensor (rows = B * C * H)
    const VT* src = reinterpret_cast<const VT*>(ft.f[blockIdx.y]);
    VT* plain = reinterpret_cast<VT*>(out + (long)blockIdx.y * 2 * rows * W);
    VT* flipped = plain + n;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const long r = e / WV; const int d = (int)(e - r * WV);
        const VT v = src[e];
        plain[e] = v;
        flipped[r * WV + (WV - 1 - d)] = FlipVec<V>::rev(v);
    }
}

LEOD_API int leod_stack_hflip_u8(const void* const* frames, int T, void* out, int B, long rows_per_sample, int W, hipStream_t stream) {
    if (!frames || !out || T < 1 || B < 1 || rows_per_sample < 1 || W < 1) return LEOD_ERR_ARG;
    const long rows = (long)B * rows_per_sample;
    for (int t0 = 0; t0 < T; t0 += 32) {
        const int nt = min(32, T - t0);
        FrameTable ft{};
        bool a16 = (W % 16 == 0) && (((uintptr_t)out) % 16 == 0), a4 = (W % 4 == 0) && (((uintptr_t)out) % 4 == 0);
        for (int k = 0; k < nt; ++k) {
            if (!frames[t0 + k]) return LEOD_ERR_ARG;
            ft.f[k] = static_cast<const unsigned char*>(frames[t0 + k]);
            a16 = a16 && ((uintptr_t)ft.f[k] % 16 == 0); a4 = a4 && ((uintptr_t)ft.f[k] % 4 == 0);
        }
        unsigned char* o = static_cast<unsigned char*>(out) + (long)t0 * 2 * rows * W;
        const int V = a16 ? 16 : (a4 ? 4 : 1);
        const long n = rows * (W / V);
        const dim3 grid((unsigned)max((long)1, min((long)2048, (n + 1023) / 1024)), nt);
        if (V == 16) hipLaunchKernelGGL(stack_hflip_u8_kernel<16>, grid, dim3(256), 0, stream, ft, o, rows, W);
        else if (V == 4) hipLaunchKernelGGL(stack_hflip_u8_kernel<4>, grid, dim3(256), 0, stream, ft, o, rows, W);
        else hipLaunchKernelGGL(stack_hflip_u8_kernel<1>, grid, dim3(256), 0, stream, ft, o, rows, W);
    }
    return leod_launch_status();
}

// ---------------------------------------------------------------------------------------------------
// Channel concatenation of two NHWC maps with an optional nearest x2 upsampling of the first (PAFPN top-down path:
// torch.cat([upsample(a), b], 1), yolo_pafpn.py:113-123 of the reference; CSPLayer: cat(x_1, x_2), network_blocks.py:160-166):
//   forward : out[b, y, x, :Ca] = a[b, y >> up, x >> up, :], out[b, y, x, Ca:] = bsrc[b, y, x, :]
//   backward: da[b, y', x', :] = sum over the 2^up x 2^up pixels it was copied to of dout[..., :Ca];  db = dout[..., Ca:]
// instead of expand + reshape-copy + cat (forward) and two slice copies + a reduction (backward).  4-byte elements, Ca % 4 == Cb % 4 == 0.
__global__ __launch_bounds__(256) void cat2_up_fwd_kernel(const float* __restrict__ a, const float* __restrict__ bsrc, float* __restrict__ out,
                                                          long npix, int H, int W, int Ca, int Cb, int up) {
    const int C4 = (Ca + Cb) / 4, A4 = Ca / 4;
    const long total = npix * C4;
    const int Ha = H >> up, Wa = W >> up;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long pix = e / C4; const int c4 = (int)(e - pix * C4);
        f4 v;
        if (c4 < A4) {
            long src = pix;
            if (up) { const int x = (int)(pix % W); const long r = pix / W; const int y = (int)(r % H); const long b = r / H;
                      src = (b * Ha + (y >> 1)) * Wa + (x >> 1); }
            v = ld4(a + src * Ca + 4 * c4);
        } else v = ld4(bsrc + pix * Cb + 4 * (c4 - A4));
        *reinterpret_cast<f4*>(out + pix * (Ca + Cb) + 4 * c4) = v;
    }
}
__global__ __launch_bounds__(256) void cat2_up_bwd_kernel(const float* __restrict__ dout, float* __restrict__ da, float* __restrict__ db,
                                                          long npix, int H, int W, int Ca, int Cb, int up) {
    // one thread per float4 of da and of db: (pixels of a) * Ca / 4 + npix * Cb / 4 work items
    const int A4 = Ca / 4, B4 = Cb / 4, C = Ca + Cb;
    const int Ha = H >> up, Wa = W >> up;
    const long na = (up ? npix / 4 : npix) * A4, total = na + npix * B4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        if (e < na) {
            const long pa = e / A4; const int c4 = (int)(e - pa * A4);
            f4 v;
            if (up) {
                const int x = (int)(pa % Wa); const long r = pa / Wa; const int y = (int)(r % Ha); const long b = r / Ha;
                const float* p = dout + ((b * H + 2 * y) * W + 2 * x) * C + 4 * c4;
                v = (ld4(p) + ld4(p + C)) + (ld4(p + (long)W * C) + ld4(p + (long)W * C + C));
            } else v = ld4(dout + pa * C + 4 * c4);
            *reinterpret_cast<f4*>(da + pa * Ca + 4 * c4) = v;
        } else {
            const long f = e - na; const long pix = f / B4; const int c4 = (int)(f - pix * B4);
            *reinterpret_cast<f4*>(db + pix * Cb + 4 * c4) = ld4(dout + pix * C + Ca + 4 * c4);
        }
    }
}
LEOD_API int leod_cat2_up_fwd(const float* a, const float* b, float* out, int B, int H, int W, int Ca, int Cb, int up, hipStream_t stream) {
    if (!a || !b || !out || B <= 0 || H <= 0 || W <= 0 || Ca <= 0 || Cb <= 0 || (Ca & 3) || (Cb & 3) || up < 0 || up > 1 ||
        (up && ((H | W) & 1))) return LEOD_ERR_ARG;
    const long npix = (long)B * H * W, total = npix * ((Ca + Cb) / 4);
    hipLaunchKernelGGL(cat2_up_fwd_kernel, dim3((unsigned)min((long)4096, (total + 255) / 256)), dim3(256), 0, stream, a, b, out, npix, H, W, Ca, Cb, up);
    return leod_launch_status();
}
LEOD_API int leod_cat2_up_bwd(const float* dout, float* da, float* db, int B, int H, int W, int Ca, int Cb, int up, hipStream_t stream) {
    if (!dout || !da || !db || B <= 0 || H <= 0 || W <= 0 || Ca <= 0 || Cb <= 0 || (Ca & 3) || (Cb & 3) || up < 0 || up > 1 ||
        (up && ((H | W) & 1))) return LEOD_ERR_ARG;
    const long npix = (long)B * H * W, total = (up ? npix / 4 : npix) * (Ca / 4) + npix * (Cb / 4);
    hipLaunchKernelGGL(cat2_up_bwd_kernel, dim3((unsigned)min((long)4096, (total + 255) / 256)), dim3(256), 0, stream, dout, da, db, npix, H, W, Ca, Cb, up);
    return leod_launch_status();
}

// dst[idx[j], :] += src[j, :] for rows of n4 float4 (UNIQUE indices: no atomics) -- the labelled frames' gradient added into the gradient of a
// stage's output map (functions.ForkSelectFn; torch's index_add_ took 20 us per stage for this)
__global__ __launch_bounds__(256) void rows_index_add_kernel(float* __restrict__ dst, const float* __restrict__ src, const long* __restrict__ idx,
                                                             long n4, int nrows_dst) {
    const long j = blockIdx.y;
    const long r = idx[j];
    if (r < 0 || r >= nrows_dst) return;
    f4* d = reinterpret_cast<f4*>(dst) + r * n4;
    const f4* sp = reinterpret_cast<const f4*>(src) + j * n4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) d[e] = d[e] + sp[e];
}
LEOD_API int leod_rows_index_add(float* dst, const float* src, const long* idx, int nsel, long row_floats, int nrows_dst, hipStream_t stream) {
    if (!dst || !src || !idx || nsel < 0 || row_floats <= 0 || (row_floats & 3) || nrows_dst <= 0 ||
        ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15)) return LEOD_ERR_ARG;
    if (nsel == 0) return LEOD_OK;
    const long n4 = row_floats / 4;
    const int gx = (int)min((long)64, (n4 + 255) / 256);
    hipLaunchKernelGGL(rows_index_add_kernel, dim3(gx, nsel), dim3(256), 0, stream, dst, src, idx, n4, nrows_dst);
    return leod_launch_status();
}

// out[i] = (i == idx) ? *g : 0 for i < n: the seed gradient of ONE entry of a small loss vector as a real zero-padded vector (one launch;
// no memset / memcpy nodes in a captured step)
__global__ void onehot_scale_kernel(const float* __restrict__ g, float* __restrict__ out, int n, int idx) {
    const int i = threadIdx.x;
    if (i < n) out[i] = i == idx ? g[0] : 0.f;
}
LEOD_API int leod_onehot_scale(const float* g, float* out, int n, int idx, hipStream_t stream) {
    if (!g || !out || n <= 0 || n > 64 || idx < 0 || idx >= n) return LEOD_ERR_ARG;
    hipLaunchKernelGGL(onehot_scale_kernel, dim3(1), dim3(64), 0, stream, g, out, n, idx);
    return leod_launch_status();
}

LEOD_API const char* leod_version() { return "leod_hip 0.2 (gfx950)"; }

// precision mode of the contractions (see common.hpp): process-wide, set once before the first step
static int g_precision = 0;
static thread_local int t_fwd_depth = 0;
int leod_precision() { return g_precision ? 1 : 0; }
int leod_precision_mode() { return g_precision; }
int leod_opfmt() { return g_precision == 2 ? (t_fwd_depth > 0 ? 2 : 1) : g_precision; }
LeodFwdScope::LeodFwdScope() { ++t_fwd_depth; }
LeodFwdScope::~LeodFwdScope() { --t_fwd_depth; }
LEOD_API int leod_set_precision(int mode) {
    if (mode != 0 && mode != 1 && mode != 2) return LEOD_ERR_ARG;
    if (mode != g_precision) leod_weight_shadow_invalidate();      // the set of copies a refresh writes depends on the mode
    g_precision = mode;
    return LEOD_OK;
}
LEOD_API int leod_get_precision() { return g_precision; }


// ---------------------------------------------------------------------------------------------------------------------
// On-device spatial augmentation of uint8 event representations (data/utils/augmentor.py:216-331,390-401):
// horizontal flip, then EITHER zoom-in (crop a window, nearest-exact resize to the full frame) OR zoom-out
// (nearest-exact resize of the full frame to a window pasted at (x0, y0) on a zero canvas), per batch sample and
// identically for all timesteps / channels of that sample.  One gather pass: every output byte is computed from the
// source byte it maps to (the reference materialises the flipped tensor, the window and the resized tensor).
// Index rule = ATen's nearest-exact: src = min(int(floorf((dst + 0.5f) * (float(in) / out))), in - 1).
// params[b] = {hflip, mode (0 none, 1 zoom-in, 2 zoom-out), x0, y0, win_h, win_w, tflip}.
// tflip (data/genx_utils/sequence_base.py:207-227, time_flip_data): the sample's frames in reverse order and each frame's
// 2*bins channel planes reversed (`x.flip(0)`), i.e. out[t, b, c] = aug(in[T-1-t, b, C-1-c]).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int nearest_exact(int dst, int in_size, int out_size) {
    const float scale = (float)in_size / (float)out_size;
    return min((int)floorf(((float)dst + 0.5f) * scale), in_size - 1);
}

__global__ __launch_bounds__(256) void augment_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                         const int* __restrict__ params, int T, int B, int C, int H, int W) {
    // grid: x = pixel quads of one plane, y = (t*B + b)*C + c
    const int plane = blockIdx.y;
    const int b = (plane / C) % B;
    const int* pp = params + 7 * b;
    const int hflip = pp[0], mode = pp[1], x0 = pp[2], y0 = pp[3], wh = pp[4], ww = pp[5];
    int src_plane = plane;
    if (pp[6]) {
        const int c = plane % C, t = plane / (C * B);
        src_plane = ((T - 1 - t) * B + b) * C + (C - 1 - c);
    }
    const uint8_t* sp = src + (long)src_plane * H * W;
    uint8_t* dp = dst + (long)plane * H * W;
    const int quads = H * (W / 4);
    for (int e = blockIdx.x * 256 + threadIdx.x; e < quads; e += gridDim.x * 256) {
        const int y = e / (W / 4), xq = (e - y * (W / 4)) * 4;
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = xq + k;
            int sy = y, sx = x;
            bool live = true;
            if (mode == 1) {                                  // zoom-in: window (wh x ww) at (y0, x0) -> full frame
                sy = y0 + nearest_exact(y, wh, H);
                sx = x0 + nearest_exact(x, ww, W);
            } else if (mode == 2) {                           // zoom-out: full frame -> window at (y0, x0), zeros elsewhere
                const int j = y - y0, i = x - x0;
                live = (unsigned)j < (unsigned)wh && (unsigned)i < (unsigned)ww;
                sy = live ? nearest_exact(j, H, wh) : 0;
                sx = live ? nearest_exact(i, W, ww) : 0;
            }
            if (hflip) sx = W - 1 - sx;                       // the zooms act on the already flipped frame
            const uint32_t v = live ? sp[(long)sy * W + sx] : 0u;
            packed |= v << (8 * k);
        }
        *reinterpret_cast<uint32_t*>(dp + (long)y * W + xq) = packed;
    }
}

LEOD_API int leod_augment_u8(const unsigned char* src, unsigned char* dst, const int* params, int T, int B, int C, int H, int W,
                             hipStream_t stream) {
    if (!src || !dst || !params || src == dst || (W & 3) || T <= 0 || B <= 0 || C <= 0) return LEOD_ERR_ARG;
    const long planes = (long)T * B * C;
    if (planes > 65535) return LEOD_ERR_UNSUPPORTED;            // grid.y limit; 21 x 8 x 20 = 3360 at the bench shape
    const int quads = H * (W / 4);
    dim3 grid(min(cdiv(quads, 256), 64), (unsigned)planes);
    hipLaunchKernelGGL(augment_u8_kernel, grid, dim3(256), 0, stream, src, dst, params, T, B, C, H, W);
    return leod_launch_status();
}
