#!/usr/bin/env python
"""Diagnosis: fused LDS attention backward on one-head workgroups with padded partitions (LEOD_ATTN_LDS_PAD1=1) against torch."""
import os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
from leod_amd import ops
torch.manual_seed(0)
dev = 'cuda'
for (B, H, W, C, heads, part) in [(1, 12, 20, 32, 1, (6, 10)), (1, 16, 16, 32, 1, (8, 8)), (1, 14, 16, 32, 1, (7, 8)), (1, 12, 20, 24, 1, (6, 10))]:
    d = C // heads
    qkv = torch.randn(B, H, W, 3 * C, device=dev)
    dout = torch.randn(B, H, W, C, device=dev)
    out, lse = ops.partition_attn_fwd(qkv, heads, part, True, want_lse=True)
    dqkv = ops.partition_attn_bwd(qkv, dout, lse, heads, part, True)
    # torch reference, window partition
    q = qkv.detach().clone().requires_grad_(True)
    x = q.view(B, H // part[0], part[0], W // part[1], part[1], 3, heads, d).permute(0, 1, 3, 5, 6, 2, 4, 7).reshape(B, H // part[0], W // part[1], 3, heads, part[0] * part[1], d)
    qq, kk, vv = x[:, :, :, 0], x[:, :, :, 1], x[:, :, :, 2]
    o = torch.softmax(qq @ kk.transpose(-1, -2) / d ** 0.5, -1) @ vv          # [B, nh, nw, heads, P, d]
    o = o.view(B, H // part[0], W // part[1], heads, part[0], part[1], d).permute(0, 1, 4, 2, 5, 3, 6).reshape(B, H, W, C)
    print('fwd err', (o - out).abs().max().item())
    o.backward(dout)
    err = (dqkv - q.grad).abs().view(B, H, W, 3, heads, d)
    print((B, H, W, C, heads, part), 'max err dq/dk/dv', [err[:, :, :, j].max().item() for j in range(3)])
    e = err[0, :part[0], :part[1]].reshape(part[0] * part[1], 3, heads, d)      # first partition, token-major
    for j, nm in enumerate(('dq', 'dk', 'dv')):
        bad = (e[:, j].amax((1, 2)) > 1e-3).nonzero().flatten().tolist()
        print('  ', nm, 'bad tokens of partition 0:', bad[:70])
        badc = (e[:, j].amax((0, 1)) > 1e-3).nonzero().flatten().tolist()
        print('  ', nm, 'bad head-dim columns:', badc)
    g = q.grad.view(B, H, W, 3, heads, d)
    k = dqkv.view(B, H, W, 3, heads, d)
    if err.max() > 1e-3:
        torch.set_printoptions(precision=3, linewidth=250, sci_mode=False)
        for tok in (1, 2):
            ty, tx = tok // part[1], tok % part[1]
            for j, nm in enumerate(('dq', 'dk', 'dv')):
                print('   token', tok, nm, 'kernel', k[0, ty, tx, j, 0].cpu())
                print('   token', tok, nm, 'ref   ', g[0, ty, tx, j, 0].cpu())
