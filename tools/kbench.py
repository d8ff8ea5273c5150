#!/usr/bin/env python
"""Per-op micro-benchmark at the RVT-S / Gen1 / bs 8 shapes (GPU box): prints us, GB/s (algorithmic bytes) and TFLOP/s
per op and stage.  KBENCH_T=21 (default) = the time-batched shapes of the training step (T*B samples per launch),
KBENCH_T=1 = one timestep.  usage: python tools/kbench.py [filter[,filter..]]"""
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
from leod_amd import ops  # noqa: E402

DEV = 'cuda'
TB = int(os.environ.get('KBENCH_T', '21'))
STAGES = [(40960 * TB, 48, 2, 64, 80), (10240 * TB, 96, 4, 32, 40), (2560 * TB, 192, 8, 16, 20), (640 * TB, 384, 16, 8, 10)]


def timeit(fn, n=int(os.environ.get('KBENCH_N', '6'))):
    if os.environ.get("KBENCH_EAGER"):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        return 1.0                                     # (placeholder time: eager mode only runs the kernels, e.g. under rocprofv3 --pmc)
    """GPU time per call: n calls captured in one hipGraph (no host launch overhead), replayed 3x."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * n)


def main():
    ops.set_precision(os.environ.get('LEOD_PRECISION', 'f32'))
    print('precision mode', ops.get_precision())
    flt = sys.argv[1] if len(sys.argv) > 1 else ''
    r = lambda *s: torch.randn(*s, device=DEV)  # noqa
    rows = []
    for si, (M, C, heads, H, W) in enumerate(STAGES):
        B = 8 * TB
        x, lw, lb = r(M, C), r(C), r(C)
        Wqkv, bqkv, Wp, bp, g = r(3 * C, C) * .1, r(3 * C), r(C, C) * .1, r(C), r(C)
        W1, b1, W2, b2 = r(4 * C, C) * .1, r(4 * C), r(C, 4 * C) * .1, r(C)
        qkv4 = r(B, H, W, 3 * C)
        qkv16 = qkv4.to(ops.act16_dtype())
        u, dyC, dy3, dy4 = r(M, 4 * C), r(M, C), r(M, 3 * C), r(M, 4 * C)
        dy3b, dy4b, u16 = dy3.to(torch.bfloat16), dy4.to(torch.bfloat16), u.to(torch.float16)
        _, _, st = ops.ln_linear_fwd(x, lw, lb, Wqkv, bqkv, want_stats=True)
        o, lse = ops.partition_attn_fwd(qkv4, heads, (8, 10), True, want_lse=True)
        Wl, bl_, h0, c0 = r(4 * C, 2 * C) * .1, r(4 * C), r(M, C), r(M, C)
        dWqkv, dbqkv, dW1, db1, dW2, db2, dWl, dbl = (torch.zeros_like(t) for t in (Wqkv, bqkv, W1, b1, W2, b2, Wl, bl_))
        cases = [
            ('ln_qkv_fwd', lambda: ops.ln_linear_fwd(x, lw, lb, Wqkv, bqkv, want_stats=True), 4 * M * 4 * C, 2 * M * C * 3 * C),
            ('ln_fc1_gelu_fwd', lambda: ops.ln_linear_fwd(x, lw, lb, W1, b1, want_act=True, want_stats=True), 4 * M * 9 * C, 2 * M * C * 4 * C),
            ('proj_lsres_fwd', lambda: ops.linear_lsres_fwd(x, Wp, bp, g, x, want_t=False), 4 * M * 3 * C, 2 * M * C * C),
            ('fc2_lsres_fwd', lambda: ops.linear_lsres_fwd(u, W2, b2, g, x, want_t=False), 4 * M * 6 * C, 2 * M * C * 4 * C),
            ('attn_fwd_window', lambda: ops.partition_attn_fwd(qkv4, heads, (8, 10), True, want_lse=True), 4 * M * 4 * C, 4 * M * 80 * C),
            ('attn_fwd_grid', lambda: ops.partition_attn_fwd(qkv4, heads, (8, 10), False, want_lse=True), 4 * M * 4 * C, 4 * M * 80 * C),
            ('attn_bwd_window', lambda: ops.partition_attn_bwd(qkv4, o, lse, heads, (8, 10), True), 4 * M * 8 * C, 10 * M * 80 * C),
            ('attn_fwd_window16', lambda: ops.partition_attn_fwd(qkv16, heads, (8, 10), True, want_lse=True), M * (2 * 3 * C + 4 * C), 4 * M * 80 * C),
            ('attn_bwd_window16', lambda: ops.partition_attn_bwd(qkv16, o, lse, heads, (8, 10), True), M * (2 * 6 * C + 4 * C), 10 * M * 80 * C),
            ('attn_bwd_grid16', lambda: ops.partition_attn_bwd(qkv16, o, lse, heads, (8, 10), False), M * (2 * 6 * C + 4 * C), 10 * M * 80 * C),
            ('convlstm_fwd', lambda: ops.convlstm_fwd(x, h0, c0, Wl, bl_, want_gates=True), 4 * M * 9 * C, 2 * M * 2 * C * 4 * C),
            ('dgrad_fc2(gelu)', lambda: ops.linear_dgrad(dyC, W2, aux_u=u), 4 * M * 9 * C, 2 * M * C * 4 * C),
            ('dgrad_fc1', lambda: ops.linear_dgrad(dy4, W1), 4 * M * 5 * C, 2 * M * C * 4 * C),
            ('dgrad_qkv', lambda: ops.linear_dgrad(dy3, Wqkv), 4 * M * 4 * C, 2 * M * C * 3 * C),
            ('dgrad_lstm', lambda: ops.linear_dgrad(dy4, Wl, split=C), 4 * M * 6 * C, 2 * M * 2 * C * 4 * C),
            ('wgrad_qkv(LN)', lambda: ops.linear_wgrad(dy3, x, dWqkv, dbqkv, stats=st, ln_w=lw, ln_b=lb), 4 * M * 4 * C, 2 * M * C * 3 * C),
            ('wgrad_fc1(LN)', lambda: ops.linear_wgrad(dy4, x, dW1, db1, stats=st, ln_w=lw, ln_b=lb), 4 * M * 5 * C, 2 * M * C * 4 * C),
            ('wgrad_fc2', lambda: ops.linear_wgrad(dyC, u, dW2, db2), 4 * M * 5 * C, 2 * M * C * 4 * C),
            ('wgrad16_qkv(LN)', lambda: ops.linear_wgrad(dy3.to(torch.bfloat16) if False else dy3b, x, dWqkv, dbqkv, stats=st, ln_w=lw, ln_b=lb), M * (2 * 3 * C + 4 * C), 2 * M * C * 3 * C),
            ('wgrad16_fc1(LN)', lambda: ops.linear_wgrad(dy4b, x, dW1, db1, stats=st, ln_w=lw, ln_b=lb), M * (2 * 4 * C + 4 * C), 2 * M * C * 4 * C),
            ('wgrad16_fc2(u16)', lambda: ops.linear_wgrad(dyC, u16, dW2, db2), M * (4 * C + 2 * 4 * C), 2 * M * C * 4 * C),
            ('wgrad_lstm', lambda: ops.linear_wgrad(dy4, x, dWl, dbl, x2=h0), 4 * M * 6 * C, 2 * M * 2 * C * 4 * C),
            ('ln_bwd', lambda: ops.layernorm_bwd(dyC, x, st, lw, dyC, dbqkv[:C], dbqkv[C:2 * C]), 4 * M * 4 * C, 0),
            ('ls_bwd', lambda: ops.layerscale_bwd(dyC, x, g, dbqkv[:C]), 4 * M * 3 * C, 0),
        ]
        for name, fn, nbytes, flops in cases:
            if flt and not any(f in name for f in flt.split(',')):
                continue
            us = timeit(fn)
            rows.append((si + 1, name, us, nbytes / us / 1e3, flops / us / 1e6))
    print(f'{"stage":>5} {"op":<20} {"us":>8} {"GB/s":>8} {"TFLOP/s":>8}')
    tot = 0
    for s, n, us, gbs, tf in rows:
        tot += us
        print(f'{s:>5} {n:<20} {us:8.1f} {gbs:8.0f} {tf:8.2f}')
    print('sum us', round(tot, 1))


if __name__ == '__main__':
    main()
