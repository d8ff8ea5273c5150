#!/usr/bin/env python
"""Practical HBM bandwidth of the box with stock torch kernels (copy = read+write, sum = read only, fill = write only)."""
import torch
dev = 'cuda'
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (64, 256, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    a = torch.randn(n, device=dev); b = torch.empty_like(a)
    tc = t(lambda: b.copy_(a)); ts = t(lambda: a.sum()); tf = t(lambda: b.fill_(1.0)); tm = t(lambda: torch.mul(a, 2.0, out=b))
    print(f'{mb:5d} MiB: copy {2*n*4/tc/1e12:.2f} TB/s  sum(read) {n*4/ts/1e12:.2f} TB/s  fill(write) {n*4/tf/1e12:.2f} TB/s  mul {2*n*4/tm/1e12:.2f} TB/s')
