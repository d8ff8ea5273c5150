#!/usr/bin/env python
"""Linear forward / dgrad GEMMs of one RVT stage at the time-batched shapes, timed with HIP events over repeated eager launches
(no graph): us, TFLOP/s and algorithmic GB/s per op.  usage: python tools/kbench_gemm.py [stages=3,4] [reps=20]   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
from leod_amd import ops  # noqa: E402

DEV = 'cuda'
KEEP = []
TB = int(os.environ.get('KBENCH_T', '21'))
STAGES = {1: (40960 * TB, 48), 2: (10240 * TB, 96), 3: (2560 * TB, 192), 4: (640 * TB, 384)}


def timeit(fn, reps):
    """KBENCH_GRAPH=1: `reps` calls captured in one hipGraph and replayed (GPU time per call without the host's launch rate: repeated
    eager launches of a < 25 us kernel measure the ~20 us per call of the Python / ctypes launch path instead)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if os.environ.get('KBENCH_GRAPH') == '1':
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (3 * reps)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ops.set_precision(os.environ.get('LEOD_PRECISION', 'bf16'))
    stages = [int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else '3,4').split(',')]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    r = lambda *s: torch.randn(*s, device=DEV)  # noqa
    print(f'precision mode {ops.get_precision()}; {"stage":>5} {"op":<18} {"us":>8} {"TFLOP/s":>8} {"GB/s":>8}')
    tot = 0.
    for si in stages:
        M, C = STAGES[si]
        x, lw, lb, g = r(M, C), r(C), r(C), r(C)
        # the weights live in one flat buffer, as the parameters of a model do (parallel.FlatParams); KBENCH_SHADOW=1 registers its bf16 shadow
        flat = torch.randn(12 * C * C, device=DEV) * .1
        Wqkv, Wp = flat[:3 * C * C].view(3 * C, C), flat[3 * C * C:4 * C * C].view(C, C)
        W1, W2 = flat[4 * C * C:8 * C * C].view(4 * C, C), flat[8 * C * C:12 * C * C].view(C, 4 * C)
        bqkv, bp, b1, b2 = r(3 * C), r(C), r(4 * C), r(C)
        if os.environ.get('KBENCH_SHADOW') == '1':
            shadow = torch.empty(flat.numel(), dtype=torch.bfloat16, device=DEV)
            ops.set_weight_shadow(flat, shadow)
            shadow16 = torch.empty(flat.numel(), dtype=torch.float16, device=DEV)
            ops.set_weight_shadow_f16(flat, shadow16)
            ops.weight_shadow_refresh()
            KEEP.append((flat, shadow, shadow16))
        u, dyC, dy3, dy4 = r(M, 4 * C), r(M, C), r(M, 3 * C), r(M, 4 * C)
        u16, dy4b, dy3b = u.to(torch.float16), dy4.to(torch.bfloat16), dy3.to(torch.bfloat16)
        Wl = r(4 * C, 2 * C) * .1
        cases = [
            ('ln_qkv', lambda: ops.ln_linear_fwd(x, lw, lb, Wqkv, bqkv, want_stats=True), 4 * M * 4 * C, 2 * M * C * 3 * C),
            ('qkv_noln', lambda: ops.ln_linear_fwd(x, None, None, Wqkv, bqkv), 4 * M * 4 * C, 2 * M * C * 3 * C),
            ('ln_fc1_gelu', lambda: ops.ln_linear_fwd(x, lw, lb, W1, b1, want_act=True, want_stats=True), 4 * M * 9 * C, 2 * M * C * 4 * C),
            ('proj_lsres', lambda: ops.linear_lsres_fwd(x, Wp, bp, g, x, want_t=False), 4 * M * 3 * C, 2 * M * C * C),
            ('fc2_lsres', lambda: ops.linear_lsres_fwd(u, W2, b2, g, x, want_t=False), 4 * M * 6 * C, 2 * M * C * 4 * C),
            ('fc2_lsres(u16)', lambda: ops.linear_lsres_fwd(u16, W2, b2, g, x, want_t=False), 4 * M * 4 * C, 2 * M * C * 4 * C),
            ('dgrad_fc2(gelu)', lambda: ops.linear_dgrad(dyC, W2, kscale=g, aux_u=u), 4 * M * 9 * C, 2 * M * C * 4 * C),
            ('dgrad_fc1', lambda: ops.linear_dgrad(dy4, W1), 4 * M * 5 * C, 2 * M * C * 4 * C),
            ('dgrad_fc1(bf16)', lambda: ops.linear_dgrad(dy4b, W1), 4 * M * 3 * C, 2 * M * C * 4 * C),
            ('dgrad_qkv', lambda: ops.linear_dgrad(dy3, Wqkv), 4 * M * 4 * C, 2 * M * C * 3 * C),
            ('dgrad_qkv(bf16)', lambda: ops.linear_dgrad(dy3b, Wqkv), 4 * M * 2.5 * C, 2 * M * C * 3 * C),
            ('dgrad_proj', lambda: ops.linear_dgrad(dyC, Wp, kscale=g), 4 * M * 2 * C, 2 * M * C * C),
            ('lstm_xproj', lambda: ops.ln_linear_fwd(x, None, None, Wl[:, :C].contiguous(), b1), 4 * M * 5 * C, 2 * M * C * 4 * C),
            ('lstm_dx', lambda: ops.linear_dgrad(dy4, Wl[:, :C].contiguous()), 4 * M * 5 * C, 2 * M * C * 4 * C),
        ]
        # the same plain contractions through torch.mm (hipBLASLt / rocBLAS, bf16 operands, fp32 accumulation): what a library GEMM does here
        W1b, Wqb, W2b, Wpb = W1.to(torch.bfloat16), Wqkv.to(torch.bfloat16), W2.to(torch.bfloat16), Wp.to(torch.bfloat16)
        xb, dyCb = x.to(torch.bfloat16), dyC.to(torch.bfloat16)
        cases += [
            ('mm dgrad_fc1', lambda: torch.mm(dy4b, W1b), 4 * M * 3 * C, 2 * M * C * 4 * C),
            ('mm dgrad_qkv', lambda: torch.mm(dy3b, Wqb), 4 * M * 2.5 * C, 2 * M * C * 3 * C),
            ('mm dgrad_fc2', lambda: torch.mm(dyCb, W2b), 4 * M * 3 * C, 2 * M * C * 4 * C),
            ('mm fwd_fc1', lambda: torch.mm(xb, W1b.t()), 4 * M * 3 * C, 2 * M * C * 4 * C),
            ('mm fwd_qkv', lambda: torch.mm(xb, Wqb.t()), 4 * M * 2.5 * C, 2 * M * C * 3 * C),
            ('mm fwd_proj', lambda: torch.mm(xb, Wpb.t()), 4 * M * 1 * C, 2 * M * C * C),
        ]
        flt = os.environ.get('KBENCH_FILTER', '')
        for name, fn, nbytes, flops in cases:
            if flt and not any(f in name for f in flt.split(',')):
                continue
            us = timeit(fn, reps)
            tot += us
            print(f'{si:>5} {name:<18} {us:8.1f} {flops / us / 1e6:8.1f} {nbytes / us / 1e3:8.0f}')
    print('sum us', round(tot, 1))


if __name__ == '__main__':
    main()
