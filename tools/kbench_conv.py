#!/usr/bin/env python
"""Per-conv micro-benchmark at the shapes of the RVT-S training step (time-batched backbone downsampling convs on
T*B = 168 frames; PAFPN / head convs on the 32 labelled frames).  usage: python tools/kbench_conv.py"""
import os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
from leod_amd import ops  # noqa: E402
from kbench import timeit  # noqa: E402

DEV = 'cuda'
CASES = [  # name, B, H, W, Cin, N, ks, stride
    ('down2 3x3/s2', 168, 64, 80, 48, 96, 3, 2), ('down3 3x3/s2', 168, 32, 40, 96, 192, 3, 2), ('down4 3x3/s2', 168, 16, 20, 192, 384, 3, 2),
    ('fpn 1x1 s32', 32, 8, 10, 384, 192, 1, 1), ('fpn 3x3 s16', 32, 16, 20, 192, 192, 3, 1), ('fpn 3x3/s2 s8', 32, 32, 40, 96, 96, 3, 2), ('fpn 3x3/s2 s16', 32, 16, 20, 192, 192, 3, 2),
    ('fpn 3x3 s8', 32, 32, 40, 96, 96, 3, 1), ('1x1 s8 96', 32, 32, 40, 96, 96, 1, 1), ('1x1 s8 192->48', 32, 32, 40, 192, 48, 1, 1), ('1x1 s16 384->96', 32, 16, 20, 384, 96, 1, 1), ('head 3x3 s8', 32, 32, 40, 96, 96, 3, 1), ('head 3x3 s32', 32, 8, 10, 96, 96, 3, 1),
]
if os.environ.get('LEOD_PRECISION'):
    ops.set_precision(os.environ['LEOD_PRECISION'])
print('precision mode', ops.get_precision())
print(f'{"conv":<16} {"fwd us":>8} {"TF/s":>6} {"+stats us":>9} {"dgrad us":>9} {"TF/s":>6} {"wgrad us":>9} {"TF/s":>6}')
for name, B, H, W, Cin, N, ks, st in CASES:
    x = torch.randn(B, H, W, Cin, device=DEV)
    w = torch.randn(N, Cin, ks, ks, device=DEV) * 0.05
    y = ops.conv_nhwc_fwd(x, w, None, stride=st)
    dy = torch.randn_like(y)
    dw = torch.zeros_like(w)
    fl = 2.0 * y.numel() // N * N * Cin * ks * ks
    tf = timeit(lambda: ops.conv_nhwc_fwd(x, w, None, stride=st))
    st64 = torch.zeros(ops.STAT_REPLICAS, 2, N, dtype=torch.float64, device=DEV)
    ts = timeit(lambda: ops.conv_nhwc_fwd(x, w, None, stride=st, colstats=st64))
    td = timeit(lambda: ops.conv_nhwc_dgrad(dy, w, x.shape, stride=st))
    tw = timeit(lambda: ops.conv_nhwc_wgrad(dy, x, dw, None, stride=st))
    print(f'{name:<16} {tf:8.1f} {fl / tf / 1e6:6.1f} {ts:9.1f} {td:9.1f} {fl / td / 1e6:6.1f} {tw:9.1f} {fl / tw / 1e6:6.1f}')
