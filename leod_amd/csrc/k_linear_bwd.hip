// Token-row contractions of the RVT backbone: input gradients (dgrad) of the Linear layers, plain / through GELU / with the LayerNorm backward of the producer in the epilogue.  All tensors channels-last ("rows" = tokens of an NHWC map).
// C-ABI declared in include/leod_hip.h.
#include "linear_common.hpp"

// dx[M,K] (=|+=) (dy[M,N] * kscale[N]) @ W[N,K]          (dgrad of y = x W^T)
//   aux_u != NULL : dx *= gelu'(aux_u[M,K])                (through GELU, maxvit.py:107)
//   nsplit > 0    : columns >= nsplit go to dx2[M, K-nsplit] (ConvLSTM: [dx | dh_prev])
//   colsum != NULL: colsum[K] += column sums of the stored dx (bias gradient of the producer)
LEOD_API int leod_linear_dgrad(const float* dy, long lddy, const float* kscale, const float* W, float* dx, long lddx,
                               float* dx2, long lddx2, int nsplit, const float* aux_u, float* colsum,
                               int accumulate, const float* dres, int M, int N, int K, int dy_bf16, hipStream_t stream) {
    if (!dy || !W || !dx || (N & 3) || (lddy & 3)) return LEOD_ERR_ARG;
    const bool out16 = (dy_bf16 & 2) != 0;          // bit 1: dx is written as bf16 rows (row-epilogue kernels only)
    dy_bf16 &= 1;
    ALRows al{}; al.x = dy; al.ld = lddy; al.kscale = kscale; al.K = N; al.fmt = dy_bf16 ? 2 : 0;
    EpStore ep = ep_store(dx, lddx, K);
    if (out16) {
        if (leod_precision() != 1 || dx2 || colsum || accumulate || dres || aux_u || nsplit > 0 || (K & 3) || !use_gemm_lds(M, cdiv(K, 16 * pick_nt(K))))
            return LEOD_ERR_UNSUPPORTED;
        ep.out_fmt = 2;
    }
    ep.out2 = dx2; ep.ld2 = lddx2; ep.nsplit = nsplit; ep.accumulate = accumulate; ep.colsum = colsum; ep.addsrc = dres;
    if (dres && (nsplit > 0 || accumulate)) return LEOD_ERR_ARG;
    if (aux_u) { ep.act = ACT_MUL_GELU_GRAD; ep.aux = aux_u; ep.ldaux = K; }
    const int nt = pick_nt(K);
    int rc = LEOD_OK;
    // contraction over N in {48, 96}, K in {192, 384} output columns (dgrad of fc2, optionally through GELU): streaming kernel
    if (!dy_bf16 && !out16 && !dx2 && !colsum && !accumulate && !dres && lddy == N && lddx == K && nsplit <= 0) {
        if (!kscale && !aux_u && use_rowstream_narrow(M, N, K))
            return launch_rowstream_narrow<1>(dy, W, nullptr, nullptr, nullptr, dx, M, N, stream);
        if (const int slab = rowstream_slab(M, K, N)) {
            if (N == 48 && slab == 12) return launch_rowstream_dgrad<3, 12>(dy, lddy, kscale, W, aux_u, dx, M, K, stream);
            if (N == 96 && slab == 8) return launch_rowstream_dgrad<6, 8>(dy, lddy, kscale, W, aux_u, dx, M, K, stream);
            if (N == 64 && slab == 8) return launch_rowstream_dgrad<4, 8>(dy, lddy, kscale, W, aux_u, dx, M, K, stream);
        }
    }
    if (use_gemm_lds(M, cdiv(K, 16 * nt))) {
        DISPATCH_NT(nt, { BLTrans bl{W, (long)K, K, NT}; rc = launch_gemm_lds<NT>(al, bl, ep, M, N, cdiv(K, 16 * NT), stream); });
        return rc;
    }
    DISPATCH_NT(nt, { BLTrans bl{W, (long)K, K, NT}; rc = launch_gemm16<NT>(al, bl, ep, M, N, cdiv(K, 16 * NT), stream); });
    return rc;
}

// dx[M,K] = LayerNorm backward of (dy[M,N] @ W[N,K]) in one pass: dn = dy W stays in registers, dx = rstd (dn w - mean(dn w) -
// xhat mean(dn w xhat)) (+ dres), dgamma[K] += sum_m dn xhat, dbeta[K] += sum_m dn   (x[M,K] = the LayerNorm input, stats[M,2] =
// its saved (mean, rstd)).  Covers K = 48 with N = 144 / 192 and M >= 16384 (stage 1); LEOD_ERR_UNSUPPORTED otherwise -- the
// caller then runs leod_linear_dgrad + leod_layernorm_bwd.
LEOD_API int leod_linear_dgrad_lnbwd(const float* dy, const float* W, const float* x, const float* stats, const float* ln_w,
                                     const float* dres, float* dx, float* dgamma, float* dbeta, int M, int N, int K, int dy_bf16,
                                     hipStream_t stream) {
    if (!dy || !W || !x || !stats || !ln_w || !dx || !dgamma || !dbeta) return LEOD_ERR_ARG;
    if (dy_bf16 && use_rowstream_narrow96(M, N, K)) {
        const dim3 g96(min(cdiv(cdiv(M, 16), 8), 256));
        if (N == 384) hipLaunchKernelGGL((rowstream_narrow_kernel<24, 2, 1, 2, 6>), g96, dim3(512), 0, stream, dy, W, nullptr, ln_w, dres, dx, M, x, stats, dgamma, dbeta);
        else hipLaunchKernelGGL((rowstream_narrow_kernel<18, 2, 1, 2, 6>), g96, dim3(512), 0, stream, dy, W, nullptr, ln_w, dres, dx, M, x, stats, dgamma, dbeta);
        return leod_launch_status();
    }
    if (!use_rowstream_narrow(M, N, K)) return LEOD_ERR_UNSUPPORTED;
    const int grid = min(cdiv(cdiv(M, 16), 4), 256 * 2);
    if (dy_bf16) {
        if (leod_precision() != 1) return LEOD_ERR_ARG;
        if (N == 192) hipLaunchKernelGGL((rowstream_narrow_kernel<12, 2, 1, 2>), dim3(grid), dim3(256), 0, stream, dy, W, nullptr, ln_w, dres, dx, M, x, stats, dgamma, dbeta);
        else hipLaunchKernelGGL((rowstream_narrow_kernel<9, 2, 1, 2>), dim3(grid), dim3(256), 0, stream, dy, W, nullptr, ln_w, dres, dx, M, x, stats, dgamma, dbeta);
        return leod_launch_status();
    }
    if (leod_precision() == 1) {
        if (N == 192) hipLaunchKernelGGL((rowstream_narrow_kernel<12, 2, 1>), dim3(grid), dim3(256), 0, stream, dy, W, nullptr, ln_w, dres, dx, M, x, stats, dgamma, dbeta);
        else hipLaunchKernelGGL((rowstream_narrow_kernel<9, 2, 1>), dim3(grid), dim3(256), 0, stream, dy, W, nullptr, ln_w, dres, dx, M, x, stats, dgamma, dbeta);
        return leod_launch_status();
    }
    if (N == 192) hipLaunchKernelGGL((rowstream_narrow_kernel<12, 2>), dim3(grid), dim3(256), 0, stream, dy, W, nullptr, ln_w, dres, dx, M, x, stats, dgamma, dbeta);
    else hipLaunchKernelGGL((rowstream_narrow_kernel<9, 2>), dim3(grid), dim3(256), 0, stream, dy, W, nullptr, ln_w, dres, dx, M, x, stats, dgamma, dbeta);
    return leod_launch_status();
}

// du[M,K] = ((dy[M,N] * kscale[N]) @ W[N,K]) * gelu'(u16[M,K])      (dgrad of fc2 through GELU, fp16 pre-activation)
LEOD_API int leod_linear_dgrad_gelu16(const float* dy, const float* kscale, const float* W, const void* u16, void* dx,
                                      int M, int N, int K, int out_bf16, hipStream_t stream) {
    if (!dy || !W || !u16 || !dx || leod_precision() != 1) return LEOD_ERR_ARG;
    const int slab = rowstream_slab(M, K, N);
    if (!slab) {
        // generic shapes (stages 3-4): LDS-staged / wide-tile dgrad, gelu'(fp16 u) and the 16-bit store in the row epilogue
        static const int gen16 = 1;
        const int nt = pick_nt(K);
        if (!gen16 || (N & 3) || (K & 3) || !use_gemm_lds(M, cdiv(K, 16 * nt))) return LEOD_ERR_UNSUPPORTED;
        ALRows al{}; al.x = dy; al.ld = N; al.kscale = kscale; al.K = N;
        EpStore ep = ep_store(reinterpret_cast<float*>(dx), K, K);
        ep.act = ACT_MUL_GELU_GRAD; ep.aux = reinterpret_cast<const float*>(u16); ep.ldaux = K; ep.aux_fmt = 1; ep.out_fmt = out_bf16 ? 2 : 0;
        int rc = LEOD_OK;
        DISPATCH_NT(nt, { BLTrans bl{W, (long)K, K, NT}; rc = launch_gemm_lds<NT>(al, bl, ep, M, N, cdiv(K, 16 * NT), stream); });
        return rc;
    }
    const int slabs = K / (16 * slab);
    const int gx = min(cdiv(cdiv(M, 16), 4), max(8, (256 * 2 / slabs) & ~7));
    float* aux = reinterpret_cast<float*>(const_cast<void*>(u16));
#define DG16_CASE(KCV, NTTV)                                                                                                         \
    if (N == 16 * KCV && slab == NTTV) {                                                                                             \
        if (out_bf16) hipLaunchKernelGGL((rowstream48_kernel<KCV, NTTV, false, false, 2, 1, true, 1>), dim3(gx, slabs), dim3(256), 0, stream, \
                                         dy, (long)N, nullptr, kscale, nullptr, 0.f, W, nullptr, reinterpret_cast<float*>(dx), aux, M, K);          \
        else hipLaunchKernelGGL((rowstream48_kernel<KCV, NTTV, false, false, 2, 1, true>), dim3(gx, slabs), dim3(256), 0, stream, dy, (long)N, \
                                nullptr, kscale, nullptr, 0.f, W, nullptr, reinterpret_cast<float*>(dx), aux, M, K);                              \
        return leod_launch_status();                                                                                                 \
    }
    DG16_CASE(3, 12) DG16_CASE(6, 8) DG16_CASE(4, 8)
#undef DG16_CASE
    return LEOD_ERR_UNSUPPORTED;
}
