// Stem (7x7 / stride 4 / pad 3 over raw uint8 voxels) kernels of precision mode bf16 (k_stem.hip), called from k_conv.hip.
#pragma once
#include <hip/hip_runtime.h>

bool stem_wgrad_bf16_supported(const void* x, int Cin, int H, int W, int N, int stride, int pad);
int stem_wgrad_bf16_launch(const float* dy, const void* x, float* dW, int B, int Cin, int H, int W, int Ho, int Wo, int N,
                           hipStream_t s);
bool stem_fwd_bf16_supported(const void* x, int Cin, int H, int W, int N, int stride, int pad);
int stem_fwd_bf16_launch(const void* x, const float* w, float* y, int B, int Cin, int H, int W, int Ho, int Wo, int N, hipStream_t s);
