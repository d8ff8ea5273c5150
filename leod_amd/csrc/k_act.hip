// Elementwise pieces of the attention-block OPTIONS no shipped config enables (the shipped block -- GELU, non-gated, LayerScale -- is fused
// into the GEMM loaders / epilogues of k_linear*.hip and k_mlp.hip and never comes here):
//   * the MLP activation by name (`mlp_activation`, models/layers/maxvit/maxvit.py:357-365 -> timm's create_act.py:62-79) and the gated
//     linear unit  h = a * act(g)  with (a | g) = the two halves of the projection (GLU.forward, maxvit.py:80-82);
//   * token masking  x[token_mask] = mask_token  of the first stage (recurrent_backbone/maxvit_rnn.py:190-192) and its backward.
// Streaming fp32 kernels, 4 columns per thread.
#include "common.hpp"

namespace {

enum Act { A_GELU = 0, A_SILU, A_RELU, A_SIGMOID, A_TANH, A_RELU6, A_LEAKY, A_ELU, A_HSIGMOID, A_HSWISH, A_MISH, A_SELU, A_CELU, A_HMISH, A_COUNT };

// value and derivative of activation ACT at x (exact forms of the torch modules; transcendentals through the hardware exp / rcp)
template <int ACT> __device__ __forceinline__ void act_vd(float x, float& v, float& d) {
    if constexpr (ACT == A_GELU) { v = gelu_erf(x); d = gelu_erf_grad(x); }
    else if constexpr (ACT == A_SILU) { const float s = sigmoidf_(x); v = x * s; d = s * (1.f + x * (1.f - s)); }
    else if constexpr (ACT == A_RELU) { v = fmaxf(x, 0.f); d = x > 0.f ? 1.f : 0.f; }
    else if constexpr (ACT == A_SIGMOID) { const float s = sigmoidf_(x); v = s; d = s * (1.f - s); }
    else if constexpr (ACT == A_TANH) { const float t = tanhf(x); v = t; d = 1.f - t * t; }
    else if constexpr (ACT == A_RELU6) { v = fminf(fmaxf(x, 0.f), 6.f); d = (x > 0.f && x < 6.f) ? 1.f : 0.f; }
    else if constexpr (ACT == A_LEAKY) { v = x > 0.f ? x : 0.01f * x; d = x > 0.f ? 1.f : 0.01f; }
    else if constexpr (ACT == A_ELU || ACT == A_CELU) { const float e = expf(fminf(x, 0.f)); v = x > 0.f ? x : e - 1.f; d = x > 0.f ? 1.f : e; }
    else if constexpr (ACT == A_HSIGMOID) { v = fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f); d = (x > -3.f && x < 3.f) ? (1.f / 6.f) : 0.f; }
    else if constexpr (ACT == A_HSWISH) {
        const float r = fminf(fmaxf(x + 3.f, 0.f), 6.f);
        v = x * r * (1.f / 6.f);
        d = x < -3.f ? 0.f : (x <= 3.f ? (2.f * x + 3.f) * (1.f / 6.f) : 1.f);
    } else if constexpr (ACT == A_MISH) {
        const float sp = x > 20.f ? x : log1pf(expf(x));          // softplus with torch's threshold
        const float t = tanhf(sp), s = sigmoidf_(x);
        v = x * t; d = t + x * s * (1.f - t * t);
    } else if constexpr (ACT == A_SELU) {
        constexpr float al = 1.6732632423543772848170429916717f, sc = 1.0507009873554804934193349852946f;
        const float e = expf(fminf(x, 0.f));
        v = sc * (x > 0.f ? x : al * (e - 1.f)); d = sc * (x > 0.f ? 1.f : al * e);
    } else {                                                     // hard mish: 0.5 x clamp(x + 2, 0, 2)
        const float c = fminf(fmaxf(x + 2.f, 0.f), 2.f);
        v = 0.5f * x * c; d = (x > -2.f && x < 0.f) ? x + 1.f : (x >= 0.f ? 1.f : 0.f);
    }
}

// forward: gated: h[m][c] = p[m][c] * act(p[m][I + c]);  plain: h = act(p)
template <int ACT>
__global__ __launch_bounds__(256) void act_glu_fwd_kernel(const float* __restrict__ p, float* __restrict__ h, long M, int I, int gated) {
    const int I4 = I >> 2;
    const long n = M * I4;
    const long ldp = gated ? 2L * I : I;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < n; it += (long)gridDim.x * 256) {
        const long m = it / I4;
        const int c = (int)(it - m * I4) << 2;
        const f4 g = ld4(p + m * ldp + (gated ? I : 0) + c);
        f4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) { float v, d; act_vd<ACT>(g[j], v, d); o[j] = v; }
        if (gated) o = o * ld4(p + m * ldp + c);
        *reinterpret_cast<f4*>(h + m * I + c) = o;
    }
}
// backward: gated: dp[:, :I] = dh * act(g), dp[:, I:] = dh * a * act'(g);  plain: dp = dh * act'(p)
template <int ACT>
__global__ __launch_bounds__(256) void act_glu_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dh, float* __restrict__ dp, long M,
                                                          int I, int gated) {
    const int I4 = I >> 2;
    const long n = M * I4;
    const long ldp = gated ? 2L * I : I;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < n; it += (long)gridDim.x * 256) {
        const long m = it / I4;
        const int c = (int)(it - m * I4) << 2;
        const f4 g = ld4(p + m * ldp + (gated ? I : 0) + c), dv = ld4(dh + m * I + c);
        f4 av, ad;
#pragma unroll
        for (int j = 0; j < 4; ++j) { float v, d; act_vd<ACT>(g[j], v, d); av[j] = v; ad[j] = d; }
        if (gated) {
            const f4 a = ld4(p + m * ldp + c);
            *reinterpret_cast<f4*>(dp + m * ldp + c) = dv * av;
            *reinterpret_cast<f4*>(dp + m * ldp + I + c) = dv * a * ad;
        } else {
            *reinterpret_cast<f4*>(dp + m * ldp + c) = dv * ad;
        }
    }
}

// x[m][:] = token where mask[m]  (in place)
__global__ __launch_bounds__(256) void token_mask_fwd_kernel(float* __restrict__ x, const unsigned char* __restrict__ mask, const float* __restrict__ token,
                                                             long M, int C) {
    const int C4 = C >> 2;
    const long n = M * C4;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < n; it += (long)gridDim.x * 256) {
        const long m = it / C4;
        if (!mask[m]) continue;
        const int c = (int)(it - m * C4) << 2;
        *reinterpret_cast<f4*>(x + m * C + c) = ld4(token + c);
    }
}
// dtoken[c] += sum over masked rows of dx[m][c];  dx[m][:] = 0 where mask[m]  (in place).  One workgroup = 64 rows x all columns
__global__ __launch_bounds__(256) void token_mask_bwd_kernel(float* __restrict__ dx, const unsigned char* __restrict__ mask, float* __restrict__ dtoken,
                                                             long M, int C) {
    const int C4 = C >> 2;
    for (int cg = threadIdx.x; cg < C4; cg += 256) {
        f4 s = zero4();
        bool any = false;
        for (long m = (long)blockIdx.x * 64; m < min(M, (long)(blockIdx.x + 1) * 64); ++m) {
            if (!mask[m]) continue;
            f4* q = reinterpret_cast<f4*>(dx + m * C + 4 * cg);
            s += *q; *q = zero4(); any = true;
        }
        if (any) {
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(dtoken + 4 * cg + j, s[j]);
        }
    }
}

static inline unsigned act_grid(long items) { const long b = (items + 255) / 256; return (unsigned)(b < 1 ? 1 : b > 4096 ? 4096 : b); }

#define LEOD_BY_ACT(ACT, ...)                                          \
    switch (ACT) {                                                     \
        case 0: { constexpr int A = 0; __VA_ARGS__; } break;           \
        case 1: { constexpr int A = 1; __VA_ARGS__; } break;           \
        case 2: { constexpr int A = 2; __VA_ARGS__; } break;           \
        case 3: { constexpr int A = 3; __VA_ARGS__; } break;           \
        case 4: { constexpr int A = 4; __VA_ARGS__; } break;           \
        case 5: { constexpr int A = 5; __VA_ARGS__; } break;           \
        case 6: { constexpr int A = 6; __VA_ARGS__; } break;           \
        case 7: { constexpr int A = 7; __VA_ARGS__; } break;           \
        case 8: { constexpr int A = 8; __VA_ARGS__; } break;           \
        case 9: { constexpr int A = 9; __VA_ARGS__; } break;           \
        case 10: { constexpr int A = 10; __VA_ARGS__; } break;         \
        case 11: { constexpr int A = 11; __VA_ARGS__; } break;         \
        case 12: { constexpr int A = 12; __VA_ARGS__; } break;         \
        default: { constexpr int A = 13; __VA_ARGS__; } break;         \
    }

}  // namespace

LEOD_API int leod_act_glu_fwd(const float* p, float* h, long M, int inner, int act, int gated, hipStream_t stream) {
    if (!p || !h || M < 0 || inner <= 0 || (inner & 3) || act < 0 || act >= A_COUNT) return LEOD_ERR_ARG;
    if (M == 0) return LEOD_OK;
    LEOD_BY_ACT(act, hipLaunchKernelGGL(act_glu_fwd_kernel<A>, dim3(act_grid(M * (inner / 4))), dim3(256), 0, stream, p, h, M, inner, gated));
    return leod_launch_status();
}
LEOD_API int leod_act_glu_bwd(const float* p, const float* dh, float* dp, long M, int inner, int act, int gated, hipStream_t stream) {
    if (!p || !dh || !dp || M < 0 || inner <= 0 || (inner & 3) || act < 0 || act >= A_COUNT) return LEOD_ERR_ARG;
    if (M == 0) return LEOD_OK;
    LEOD_BY_ACT(act, hipLaunchKernelGGL(act_glu_bwd_kernel<A>, dim3(act_grid(M * (inner / 4))), dim3(256), 0, stream, p, dh, dp, M, inner, gated));
    return leod_launch_status();
}
LEOD_API int leod_token_mask_fwd(float* x, const unsigned char* mask, const float* token, long M, int C, hipStream_t stream) {
    if (!x || !mask || !token || M < 0 || C <= 0 || (C & 3)) return LEOD_ERR_ARG;
    if (M == 0) return LEOD_OK;
    hipLaunchKernelGGL(token_mask_fwd_kernel, dim3(act_grid(M * (C / 4))), dim3(256), 0, stream, x, mask, token, M, C);
    return leod_launch_status();
}
LEOD_API int leod_token_mask_bwd(float* dx, const unsigned char* mask, float* dtoken, long M, int C, hipStream_t stream) {
    if (!dx || !mask || !dtoken || M < 0 || C <= 0 || (C & 3)) return LEOD_ERR_ARG;
    if (M == 0) return LEOD_OK;
    hipLaunchKernelGGL(token_mask_bwd_kernel, dim3((unsigned)((M + 63) / 64)), dim3(256), 0, stream, dx, mask, dtoken, M, C);
    return leod_launch_status();
}
