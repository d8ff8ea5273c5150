#!/usr/bin/env python
"""Host cost per call of the Python -> ctypes -> HIP launch chain (what bounds the eager step): raw ctypes launch, ops wrapper, autograd node."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from leod_amd import ops, functions as Fn
from leod_amd.models.detection.yolox.models.network_blocks import BaseConv

dev = 'cuda'
ops.set_precision('bf16')
N = 2000


def timeit(name, fn, n=N):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{name:46s} host {1e6 * (t1 - t0) / n:7.2f} us/call   (drained {1e6 * (t2 - t0) / n:7.2f})', flush=True)


x = torch.randn(64, 48, device=dev)
w = torch.ones(48, device=dev)
b = torch.zeros(48, device=dev)
y = torch.empty_like(x)
l = ops._l()
s = ops._stream()
timeit('raw ctypes leod_layernorm_fwd (64 rows)', lambda: l.leod_layernorm_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), None, 64, 48, 1e-5, s))
px, pw, pb, py = x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr()
timeit('raw ctypes, pointers precomputed', lambda: l.leod_layernorm_fwd(px, pw, pb, py, None, 64, 48, 1e-5, s))
timeit('ops.layernorm_fwd', lambda: ops.layernorm_fwd(x, w, b))
timeit('torch.empty', lambda: torch.empty((64, 48), dtype=torch.float32, device=dev))
timeit('torch add (ATen launch)', lambda: torch.add(x, x))
timeit('ops._stream()', lambda: ops._stream())
e = torch.cuda.Event()
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
timeit('side.wait_stream(main)', lambda: side.wait_stream(main))
def ctxsw():
    with torch.cuda.stream(side):
        pass
timeit('with torch.cuda.stream(side): pass', ctxsw)
conv = BaseConv(48, 48, 3, 1).to(dev).train()
xi = torch.randn(2, 8, 10, 48, device=dev, requires_grad=True)
ops.StatArena.begin_step(dev)
def f():
    ops.StatArena.off = 0
    return conv.forward_nhwc(xi)
timeit('BaseConv.forward_nhwc (autograd node, 2 launches+pack)', f)
yy = f()
g = torch.ones_like(yy)
def fb():
    ops.StatArena.off = 0
    yy = conv.forward_nhwc(xi)
    yy.backward(g)
timeit('BaseConv fwd + bwd', fb, 1000)
Fn.WgradSide.active = True
def fb2():
    ops.StatArena.off = 0
    yy = conv.forward_nhwc(xi)
    yy.backward(g)
    Fn.WgradSide.join()
timeit('BaseConv fwd + bwd, wgrad side stream', fb2, 1000)
