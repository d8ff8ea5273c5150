#!/usr/bin/env python
"""BatchNorm + SiLU kernels at the PAFPN / head shapes of the RVT-S training step (32 labelled frames).
usage: python tools/kbench_bn.py"""
import os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
from leod_amd import ops  # noqa: E402
from kbench import timeit  # noqa: E402

DEV = 'cuda'
if os.environ.get('LEOD_PRECISION'):
    ops.set_precision(os.environ['LEOD_PRECISION'])
print(f'{"shape":<22} {"bwd reduce us":>14} {"bwd apply us":>13}')
for name, M, N in [('s8 96', 32 * 32 * 40, 96), ('s8 48', 32 * 32 * 40, 48), ('s16 96', 32 * 16 * 20, 96), ('s16 192', 32 * 16 * 20, 192),
                   ('s32 192', 32 * 8 * 10, 192), ('s32 384', 32 * 8 * 10, 384)]:
    z, dy = torch.randn(M, N, device=DEV), torch.randn(M, N, device=DEV)
    mean, rstd, w, b = torch.zeros(N, device=DEV), torch.ones(N, device=DEV), torch.ones(N, device=DEV), torch.zeros(N, device=DEV)
    sums = torch.zeros(ops.bn_bwd_replicas(M), 2, N, dtype=torch.float64, device=DEV)
    dw, db = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV)
    tr = timeit(lambda: ops.bn_silu_bwd_reduce(dy, z, mean, rstd, w, b, out=sums))
    ta = timeit(lambda: ops.bn_silu_bwd_apply(dy, z, mean, rstd, w, b, sums, dw, db, float(M)))
    print(f'{name:<22} {tr:14.1f} {ta:13.1f}')
