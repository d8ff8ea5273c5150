// Common device helpers for the LEOD gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LEOD_OK 0
#define LEOD_ERR_ARG (-1)        // bad / unsupported argument combination
#define LEOD_ERR_LAUNCH (-2)     // hipLaunch reported an error
#define LEOD_ERR_UNSUPPORTED (-3)

#define LEOD_API extern "C" __attribute__((visibility("default")))

typedef float f4 __attribute__((ext_vector_type(4)));

static inline int leod_launch_status() {
    return hipGetLastError() == hipSuccess ? LEOD_OK : LEOD_ERR_LAUNCH;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- wave64 helpers ------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// reduce over the 4 lanes that share (lane & 15): lanes l, l^16, l^32, l^48
__device__ __forceinline__ float quad16_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float quad16_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
// reduce over the 16 lanes that share (lane >> 4)
__device__ __forceinline__ float row16_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

// Transcendentals of the hot epilogues.  The exact-GELU layers evaluate erf on 4C channels of every token in both
// passes, which made libm's erff/expf a first-order VALU cost of the memory-bound MLP kernels.  These use the hardware
// v_exp_f32 / v_rcp_f32 (1 ulp) and the Abramowitz-Stegun 7.1.26 rational form of the normal CDF (|error| <= 1.5e-7
// absolute, the size of one fp32 rounding of the reference's own 0.5*(1+erf(x/sqrt2)) evaluation).
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoidf_(float x) { return fast_rcp(1.0f + fast_exp(-x)); }
// Phi(x) = 0.5*(1+erf(x/sqrt2)); e_out = exp(-x*x/2) (shared with the Gaussian pdf of the GELU derivative)
__device__ __forceinline__ float normal_cdf(float x, float& e_out) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = fast_rcp(fmaf(0.3275911f, z, 1.0f));
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = fast_exp(-z * z);
    e_out = e;
    const float half_tail = 0.5f * poly * e;                 // = 0.5*erfc(|x|/sqrt2)
    return x >= 0.f ? 1.0f - half_tail : half_tail;
}
__device__ __forceinline__ float gelu_erf(float x) { float e; return x * normal_cdf(x, e); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    float e;
    const float cdf = normal_cdf(x, e);
    return fmaf(x * 0.39894228040143267794f, e, cdf);
}
__device__ __forceinline__ float siluf_(float x) { return x * sigmoidf_(x); }
__device__ __forceinline__ float silu_grad(float x) {
    const float s = sigmoidf_(x);
    return s * (1.0f + x * (1.0f - s));
}

__device__ __forceinline__ f4 ld4(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ f4 zero4() { f4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

// v_mfma_f32_16x16x4_f32: exact fp32 (bitwise an fmaf chain over k).
//   A: lane l holds A[i = l&15][k = l>>4];  B: lane l holds B[k = l>>4][j = l&15]
//   C/D: lane l, reg r: row = 4*(l>>4) + r, col = l&15
__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}


// ---- bf16 MFMA operands (precision mode "bf16", reference: Lightning precision=16, train.py:236-243) -----------------------------
// v_mfma_f32_16x16x16_bf16: A: lane l holds A[i = l&15][k = 4*(l>>4) .. +3] as 4 bf16; B: B[k = 4*(l>>4) .. +3][j = l&15];
// C/D as above.  One instruction contracts the same 16-k chunk the fp32 path feeds to FOUR v_mfma_f32_16x16x4_f32 (the
// K-permutation "lane (i,q) holds k0+4q..+3" IS this operand layout), at 8x the fp32 MFMA rate; fp32 accumulation.
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f2_ __attribute__((ext_vector_type(2)));
typedef __bf16 bf2_ __attribute__((ext_vector_type(2)));
typedef unsigned u2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s4 pack_bf16(f4 v) {             // 2 x v_cvt_pk_bf16_f32 (round to nearest even)
    const f2_ lo = {v.x, v.y}, hi = {v.z, v.w};
    const u2_ r = {__builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf2_)),
                   __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf2_))};
    return __builtin_bit_cast(s4, r);
}
// fp16 storage of the MLP hidden pre-activation (precision mode bf16, stages 1-2): the reference holds that tensor in fp16 under
// autocast (11-bit significand: 8x less rounding noise than bf16 for a tensor that is only stored, never an MFMA operand as is);
// values are clamped to the fp16 range so that an outlier saturates instead of becoming inf.
typedef _Float16 h2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s4 pack_h16(f4 v) {              // 2 x v_cvt_pk (round to nearest even)
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fminf(fmaxf(v[j], -65504.f), 65504.f);
    const f2_ lo = {v.x, v.y}, hi = {v.z, v.w};
    const u2_ r = {__builtin_bit_cast(unsigned, __builtin_convertvector(lo, h2_)),
                   __builtin_bit_cast(unsigned, __builtin_convertvector(hi, h2_))};
    return __builtin_bit_cast(s4, r);
}
__device__ __forceinline__ f4 unpack_h16(s4 v) {            // 4 fp16 -> 4 fp32 (exact)
    f4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (float)__builtin_bit_cast(_Float16, (unsigned short)v[j]);
    return o;
}
__device__ __forceinline__ float unpack_h16_1(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ f4 unpack_bf16(s4 v) {           // 4 bf16 -> 4 fp32 (exact)
    f4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = __builtin_bit_cast(float, (unsigned)(unsigned short)v[j] << 16);
    return r;
}
__device__ __forceinline__ f4 mfma16_bf16(s4 a, s4 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }

// v_mfma_f32_16x16x32_bf16: 8 bf16 per lane and operand (k = 8*(l>>4) .. +7); on gfx950 the 16-k form above issues at the same 16
// cycles per instruction, i.e. at half the bf16 MFMA rate -- MFMA-bound kernels use this one.
typedef short s8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f4 mfma32_bf16(s8v a, s8v b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, b), c, 0, 0, 0);
}

// ---- fp16 MFMA operands (precision mode "16f": the reference's own autocast dtype, train.py:236-243 precision=16 -> torch.float16) ----
// Same operand layouts and issue rates as the bf16 forms above; 11 significand bits instead of 8.  Forward contractions of mode 16f use
// these (activations and weights of the path are bounded: LayerNorm / BatchNorm+SiLU outputs, |h| < 1, uint8 counts, softmax weights);
// gradient contractions keep bf16 operands (the range of fp32 without a loss scaler).
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f4 mfma16_f16(s4 a, s4 b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h4v, a), __builtin_bit_cast(h4v, b), c, 0, 0, 0);
}
__device__ __forceinline__ f4 mfma32_f16(s8v a, s8v b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
}
__device__ __forceinline__ s4 pack_h16_raw(f4 v) {          // no clamp: operands known to be far inside the fp16 range (weights, LayerNorm rows, P)
    const f2_ lo = {v.x, v.y}, hi = {v.z, v.w};
    const u2_ r = {__builtin_bit_cast(unsigned, __builtin_convertvector(lo, h2_)),
                   __builtin_bit_cast(unsigned, __builtin_convertvector(hi, h2_))};
    return __builtin_bit_cast(s4, r);
}
__device__ __forceinline__ s4 pack_h16_sat(f4 v) {          // one v_med3_f32 per element: an outlier saturates instead of becoming inf
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_fmed3f(v[j], -65504.f, 65504.f);
    return pack_h16_raw(v);
}
// Operand format OF of a 16-bit contraction: 1 = bf16, 2 = fp16 (0 = fp32 kernels never call these).  Kernels carry it as an int
// template parameter (`BF` in the older skeletons: 0 / 1 / 2, every `if constexpr (BF)` reads "16-bit operands").
template <int OF> __device__ __forceinline__ s4 pack16(f4 v) { if constexpr (OF == 2) return pack_h16_sat(v); else return pack_bf16(v); }
template <int OF> __device__ __forceinline__ s4 pack16_raw(f4 v) { if constexpr (OF == 2) return pack_h16_raw(v); else return pack_bf16(v); }
template <int OF> __device__ __forceinline__ f4 unpack16(s4 v) { if constexpr (OF == 2) return unpack_h16(v); else return unpack_bf16(v); }
template <int OF> __device__ __forceinline__ f4 mfma16_16(s4 a, s4 b, f4 c) { if constexpr (OF == 2) return mfma16_f16(a, b, c); else return mfma16_bf16(a, b, c); }
template <int OF> __device__ __forceinline__ f4 mfma32_16(s8v a, s8v b, f4 c) { if constexpr (OF == 2) return mfma32_f16(a, b, c); else return mfma32_bf16(a, b, c); }
// fp16 -> bf16 re-rounding of four stored values (backward kernels of mode 16f that contract forward-stored fp16 rows with bf16 gradients)
__device__ __forceinline__ s4 h16_to_bf16(s4 v) { return pack_bf16(unpack_h16(v)); }

// Operand fragment of one 16-k chunk: OF = 0 keeps the f4 (four exact fp32 MFMAs consume it), OF = 1 / 2 packs it to 4 bf16 / fp16 once
// (one 16-bit MFMA consumes it).
template <int OF> struct Frag16 { s4 v; __device__ __forceinline__ void set(f4 x) { v = pack16<OF>(x); } };
template <> struct Frag16<0> { f4 v; __device__ __forceinline__ void set(f4 x) { v = x; } };
template <int OF> __device__ __forceinline__ f4 mfma_frag(const Frag16<OF>& a, const Frag16<OF>& b, f4 c) {
    if constexpr (OF != 0) return mfma16_16<OF>(a.v, b.v, c);
    else {
#pragma unroll
        for (int j = 0; j < 4; ++j) c = mfma16(a.v[j], b.v[j], c);
        return c;
    }
}

// Precision mode of the library (host side), set through leod_set_precision:
//   0 = fp32 end to end (bit-tight against the fp32 oracle);
//   1 = "bf16": bf16 MFMA operands everywhere, fp32 accumulation / statistics / state;
//   2 = "16f": as 1, but the FORWARD contractions take fp16 operands (the reference's autocast dtype) and the 16-bit activations the
//       forward pass leaves in HBM (qkv, attention output, MLP hidden, conv packs, weight shadow) are fp16; gradients stay bf16.
// leod_precision() is 1 in both 16-bit modes (tensor layouts, kernel families); leod_opfmt() is the operand format (0 / 1 / 2) of the
// contraction being launched: 2 only inside a LeodFwdScope of mode 2.  Forward entry points of the C ABI open such a scope.
int leod_precision();
int leod_precision_mode();
int leod_opfmt();
// Launch plans (k_plan.hip): kernels that read the step's INPUT tensor register themselves once (host function pointer, index of the input pointer
// among their arguments, argument count), so that a plan can re-point them at the batch's own buffer instead of copying the batch into a static one.
void leod_register_input_kernel(const void* func, int arg_index, int nargs);

struct LeodFwdScope { LeodFwdScope(); ~LeodFwdScope(); LeodFwdScope(const LeodFwdScope&) = delete; };
// 16-bit shadow of a registered fp32 weight buffer (k_misc.hip: leod_set_weight_shadow / leod_weight_shadow_refresh): the copy of the
// weight at `w` in operand format `of` (1 bf16 / 2 fp16) if `w` lies in a registered buffer whose shadow is fresh, else nullptr.  The GEMM
// weight loaders of the 16-bit modes read it instead of converting fp32 weights on every tile load (same rounding) -- half the L2 -> CU bytes.
const unsigned short* leod_shadow_of(const float* w, int of = 1);
#define LEOD_BY_PREC(CALL_BF, CALL_F32) (leod_precision() == 1 ? (CALL_BF) : (CALL_F32))
// run CALL with `constexpr int OF` = the operand format of the launch (leod_opfmt())
#define LEOD_BY_OPFMT(...)                                                   \
    switch (leod_opfmt()) {                                                  \
        case 2: { constexpr int OF = 2; __VA_ARGS__; } break;                \
        case 1: { constexpr int OF = 1; __VA_ARGS__; } break;                \
        default: { constexpr int OF = 0; __VA_ARGS__; } break;               \
    }
// the same where operand format 2 is only instantiated when FWD (a compile-time bool: the loader types of forward contractions)
#define LEOD_BY_OPFMT_IF(FWD, ...)                                           \
    switch (leod_opfmt()) {                                                  \
        case 2: if constexpr (FWD) { constexpr int OF = 2; __VA_ARGS__; break; }   \
        case 1: { constexpr int OF = 1; __VA_ARGS__; } break;                \
        default: { constexpr int OF = 0; __VA_ARGS__; } break;               \
    }
#define LEOD_BY_OPFMT16_IF(FWD, ...)                                         \
    switch (leod_opfmt()) {                                                  \
        case 2: if constexpr (FWD) { constexpr int OF = 2; __VA_ARGS__; break; }   \
        default: { constexpr int OF = 1; __VA_ARGS__; } break;               \
    }
#define LEOD_BY_OPFMT16(...)                                                 \
    switch (leod_opfmt()) {                                                  \
        case 2: { constexpr int OF = 2; __VA_ARGS__; } break;                \
        default: { constexpr int OF = 1; __VA_ARGS__; } break;               \
    }
