#!/usr/bin/env python
"""Stem convolution (7x7 / stride 4 over raw uint8 voxels) forward + weight gradient at the benchmarked shape (T*B = 168 frames of
20 x 240 x 304, padded 256 x 320 -> 64 x 80 x 48), both precision modes.  usage: python tools/kbench_stem.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from leod_amd import ops

dev = torch.device('cuda', 0)
B, C, H, W, Hp, Wp, N = 168, 20, 240, 304, 256, 320, 48
g = torch.Generator().manual_seed(0)
x = ((torch.rand((B, C, H, W), generator=g) < 0.08) * torch.randint(1, 10, (B, C, H, W), generator=g)).to(torch.uint8).to(dev)
w = (torch.randn((N, C, 7, 7), generator=g) * 0.05).to(dev)
dy = torch.randn((B, Hp // 4, Wp // 4, N), generator=g).to(dev)
flops = 2 * B * (Hp // 4) * (Wp // 4) * N * C * 49


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


ref = None
for mode in ('f32', 'bf16'):
    ops.set_precision(mode)
    dw = torch.zeros_like(w)
    us_f = timeit(lambda: ops.stem_conv_fwd(x, w, (Hp, Wp), 4, 3))
    us_w = timeit(lambda: ops.stem_conv_wgrad(dy, x, dw, (Hp, Wp), 4, 3))
    dw.zero_(); ops.stem_conv_wgrad(dy, x, dw, (Hp, Wp), 4, 3); torch.cuda.synchronize()
    if ref is None:
        ref = dw.clone()
    err = float((dw - ref).norm() / ref.norm())
    print(f'{mode}: fwd {us_f:7.1f} us ({flops / us_f / 1e6:6.1f} TFLOP/s)   wgrad {us_w:7.1f} us ({flops / us_w / 1e6:6.1f} TFLOP/s)   '
          f'dW rel. deviation from the fp32 mode {err:.2e}')
