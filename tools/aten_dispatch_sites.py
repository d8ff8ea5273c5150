#!/usr/bin/env python
"""Which source lines issue the ATen device ops of one training step?  A TorchDispatchMode sees every aten op the step dispatches
(autograd's worker threads are switched off so that the backward pass runs on this thread too); ops that only make views are dropped,
the rest are grouped by (op, innermost leod_amd frame).  usage: python tools/aten_dispatch_sites.py   (GPU box)"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.argv = [sys.argv[0], '3']
import runpy
import torch
from torch.utils._python_dispatch import TorchDispatchMode
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'step_drift.py'), run_name='drift')
fit_step, mod, opt, sched, batch = ns['fit_step'], ns['mod'], ns['opt'], ns['sched'], ns['batch']
torch.autograd.set_multithreading_enabled(False)
for s in range(2):
    fit_step(mod, opt, sched, batch(), s)
torch.cuda.synchronize()
VIEWS = ('view', 'as_strided', 'reshape', 'permute', 'transpose', 'slice', 'select', 'unsqueeze', 'squeeze', 'expand', 'detach', 'alias',
         't.default', 'unbind', 'split', 'narrow', '_unsafe_view', 'size', 'stride', 'numel', 'is_', 'dim', 'sym_', 'lift_fresh',
         'empty', 'unfold', 'chunk', 'record_stream', '_local_scalar_dense', 'set_', 'resize_', 'contiguous')
sites = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(v in name.split('.')[1] if '.' in name else False for v in VIEWS):
            return out
        dev = any(isinstance(a, torch.Tensor) and a.is_cuda for a in list(args) + list((kwargs or {}).values())) or \
            (isinstance(out, torch.Tensor) and out.is_cuda)
        if not dev:
            return out
        fr = [f for f in traceback.extract_stack() if 'leod_amd' in f.filename or 'bench.py' in f.filename]
        site = f'{os.path.relpath(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].line}' if fr else '?'
        shp = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), ())
        sites[(name, site[:120], str(shp)[:40])] += 1
        return out


STEPS = 2
with Spy():
    for s in range(STEPS):
        fit_step(mod, opt, sched, batch(), 2 + s)
torch.cuda.synchronize()
tot = 0
for (name, site, shp), n in sorted(sites.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f'{n / STEPS:6.1f}  {name:32s} {shp:40s} {site}')
    tot += n
print('ops per step:', tot / STEPS)
