// Direct 3x3 convolution (stride 1; weight gradient also stride 2) (k_conv3.hip), called from the conv entry points of k_conv.hip in precision mode bf16.
#pragma once
#include <stddef.h>
#include <hip/hip_runtime.h>

bool conv3s1_supported(int H, int W, int Cin, int Cout);
size_t conv3s1_pack_bytes(int Cin, int Cout);
int conv3s1_launch(const float* x, const float* w, float* y, double* colstats, int stat_rep, int accumulate, int B, int H, int W,
                   int Cin, int Cout, int transposed, void* wpack, hipStream_t stream, int stride = 1, int packed = 0,
                   const float* bn_w = nullptr, const float* bn_b = nullptr, const float* bn_rm = nullptr, const float* bn_rv = nullptr,
                   float bn_eps = 1e-5f);      // bn_w != NULL: eval-mode BatchNorm + SiLU applied to the finished rows (forward only)
// n <= 8 independent stride-1 problems of one (Cin, Cout) geometry in one launch (forward with statistics, or transposed = 1: input gradients)
bool conv3s1_group_supported(int n, const int* H, const int* W, int Cin, int Cout);
int conv3s1_group(int n, const float* const* x, const float* const* w, float* const* y, double* const* colstats, const int* stat_rep,
                  const int* accumulate, const int* B, const int* H, const int* W, int Cin, int Cout, int transposed, void* const* wpack,
                  const int* packed, hipStream_t stream);
bool conv3s2_fwd_supported(int B, int H, int W, int Cin, int Cout);      // forward of a stride-2 conv on the same kernel (stride = 2, H, W: input size)
// weight gradient of a 3x3 / pad-1 conv of stride 1 or 2 (x [B,H,W,Cin], dy [B,H/stride,W/stride,Cout])
bool conv3_wgrad_supported(int H, int W, int Cin, int Cout, int stride);
size_t conv3_wgrad_workspace_floats(int B, int H, int W, int Cin, int Cout, int stride);
int conv3_wgrad_launch(const float* dy, const float* x, float* dW, float* ws, int B, int H, int W, int Cin, int Cout, int stride, hipStream_t stream);
// n <= 8 weight gradients of one (Cin, Cout, stride) geometry on the nine-tap kernel in one launch (+ one reduce launch); ws[k]:
// conv3_wgrad_group_workspace_floats(B[k], H[k], W[k], ..) floats
size_t conv3_wgrad_group_workspace_floats(int B, int H, int W, int Cin, int Cout, int stride);
int conv3_wgrad_group_launch(int n, const float* const* dy, const float* const* x, float* const* dW, float* const* ws, const int* B, const int* H,
                             const int* W, int Cin, int Cout, int stride, hipStream_t stream, bool force = true);
// input gradient of a 3x3 / stride-2 / pad-1 conv: dx [B,H,W,Cin] from dy [B,H/2,W/2,N], w [N][Cin][3][3]; wpack: conv3s1_pack_bytes(Cin, N)
bool conv3s2_dgrad_supported(int H, int W, int Cin, int N);
int conv3s2_dgrad_launch(const float* dy, const float* w, float* dx, int accumulate, int B, int H, int W, int Cin, int N, void* wpack, hipStream_t stream, int packed = 0);
