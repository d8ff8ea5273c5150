#!/usr/bin/env python
"""Weight-gradient launches of stages 3-4 (RVT-S, Gen1, bs 8, T 21: 53 760 / 13 440 rows) in precision mode 16f, graph-timed
(tools/kbench.py timeit): us per launch of the five Linear weight gradients of a stage with the operand formats of the training step."""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
from leod_amd import ops  # noqa: E402
from kbench import timeit  # noqa: E402

DEV = 'cuda'


def main():
    ops.set_precision(os.environ.get('LEOD_PRECISION', '16f'))
    r = lambda *s: torch.randn(*s, device=DEV)  # noqa
    tot = 0.0
    shapes = {'gen1': ((53760, 192), (13440, 384)), 'gen1s2': ((215040, 96),), '1mpx': ((1351680, 64), (337920, 128), (84480, 256), (21120, 512))}[sys.argv[1] if len(sys.argv) > 1 else 'gen1']
    for M, C in shapes:
        x, lw, lb = r(M, C), r(C), r(C)
        _, st = ops.layernorm_fwd(x, lw, lb, want_stats=True)
        dq, du = r(M, 3 * C).to(torch.bfloat16), r(M, 4 * C).to(torch.bfloat16)
        dz = r(M, C)
        o16, u16, h = r(M, C).to(ops.act16_dtype()), r(M, 4 * C).to(torch.float16), r(M, C)
        dg = r(M, 4 * C).to(torch.bfloat16)
        Wq, W1, W2, Wp, Wl = (torch.zeros(a, b, device=DEV) for a, b in ((3 * C, C), (4 * C, C), (C, 4 * C), (C, C), (4 * C, 2 * C)))
        bq, b1, b2, bp, bl = (torch.zeros(a, device=DEV) for a in (3 * C, 4 * C, C, C, 4 * C))
        cases = [('qkv (bf16 dY, LN x)', lambda: ops.linear_wgrad(dq, x, Wq, bq, stats=st, ln_w=lw, ln_b=lb), 3 * C, C),
                 ('proj (fp32 dY, 16-bit O)', lambda: ops.linear_wgrad(dz, o16, Wp, bp, x_gelu=False), C, C),
                 ('fc1 (bf16 dY, LN x)', lambda: ops.linear_wgrad(du, x, W1, b1, stats=st, ln_w=lw, ln_b=lb), 4 * C, C),
                 ('fc2 (fp32 dY, gelu(fp16 u))', lambda: ops.linear_wgrad(dz, u16, W2, b2), C, 4 * C),
                 ('lstm (bf16 dY, [x | h])', lambda: ops.linear_wgrad(dg, x, Wl, bl, x2=h), 4 * C, 2 * C)]
        for name, fn, N, K in cases:
            us = timeit(fn)
            tot += us
            print(f'{M:7d} {name:<30s} {N:5d} x {K:5d}  {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s  {2.0 * M * (N + K) / us / 1e3:7.0f} GB/s (16-bit operands)')
    print('sum us', round(tot, 1))


if __name__ == '__main__':
    main()
