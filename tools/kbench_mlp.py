#!/usr/bin/env python
"""Fused stage-1 MLP kernels (csrc/k_mlp.hip) against the launches they replace, RVT-S step shapes, precision mode bf16: us per launch, GB/s."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from leod_amd import ops
ops.set_precision('bf16')
dev = 'cuda'
M, C = int(sys.argv[1]) if len(sys.argv) > 1 else 860160, 48
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s, sc=1.0: sc * torch.randn(*s, generator=g, device=dev)
y, dz = r(M, C), r(M, C)
lw, lb = 1 + 0.1 * r(C), 0.1 * r(C)
W1, b1, W2, b2, ga = r(4 * C, C, sc=0.2), r(4 * C, sc=0.1), r(C, 4 * C, sc=0.1), r(C, sc=0.1), 0.5 + 0.1 * r(C)
dlw, dlb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)


def timeit(name, fn, nbytes, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / n
    print(f'{name:58s} {us:8.1f} us  {nbytes / us / 1e3:7.0f} GB/s', flush=True)


MB = M * C * 4
timeit('mlp_fwd_fused (inference: no u16 / stats)', lambda: ops.mlp_fwd_fused(y, lw, lb, W1, b1, W2, b2, ga, want_saved=False), 2 * MB)
timeit('mlp_fwd_fused (training: + fp16 u, stats)', lambda: ops.mlp_fwd_fused(y, lw, lb, W1, b1, W2, b2, ga, want_saved=True), 2 * MB + M * 4 * C * 2 + M * 8)
os.environ['X'] = '1'
_, u16, st = ops.mlp_fwd_fused(y, lw, lb, W1, b1, W2, b2, ga, want_saved=True)
timeit('  unfused: ln_linear_fwd (fp16 u)', lambda: ops.ln_linear_fwd(y, lw, lb, W1, b1, want_act=True, want_stats=True), MB + M * 4 * C * 2 + M * 8)
timeit('  unfused: linear_lsres_fwd from fp16 u', lambda: ops.linear_lsres_fwd(u16, W2, b2, ga, y, want_t=False), 2 * MB + M * 4 * C * 2)
timeit('mlp_bwd_dgrad_fused (+ bf16 du)', lambda: ops.mlp_bwd_dgrad_fused(dz, y, st, lw, lb, W1, b1, W2, ga, dlw, dlb), 3 * MB + M * 4 * C * 2)
timeit('mlp_bwd_dgrad_fused (no du)', lambda: ops.mlp_bwd_dgrad_fused(dz, y, st, lw, lb, W1, b1, W2, ga, dlw, dlb, want_du=False), 3 * MB)
du = ops.linear_dgrad(dz, W2, kscale=ga, aux_u=u16)
timeit('  unfused: linear_dgrad through GELU', lambda: ops.linear_dgrad(dz, W2, kscale=ga, aux_u=u16), MB + 2 * M * 4 * C * 2)
timeit('  unfused: linear_dgrad_ln_bwd', lambda: ops.linear_dgrad_ln_bwd(du, W1, y, st, lw, dz, dlw, dlb), 3 * MB + M * 4 * C * 2)
