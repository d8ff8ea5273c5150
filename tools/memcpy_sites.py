#!/usr/bin/env python
"""Memcpy / memset activities of one training step by kind and by the ATen op that issued them.  usage: python tools/memcpy_sites.py"""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.argv = [sys.argv[0], '3']
import runpy
import torch
from torch.profiler import profile, ProfilerActivity
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'step_drift.py'), run_name='drift')
fit_step, mod, opt, sched, batch = ns['fit_step'], ns['mod'], ns['opt'], ns['sched'], ns['batch']
for s in range(2):
    fit_step(mod, opt, sched, batch(), s)
torch.cuda.synchronize()
STEPS = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for s in range(STEPS):
        fit_step(mod, opt, sched, batch(), 2 + s)
    torch.cuda.synchronize()
kinds = collections.Counter()
for e in prof.events():
    n = e.name
    if 'emcpy' in n or 'emset' in n:
        kinds[n] += 1
print('device-side memcpy / memset activities per step:')
for k, v in kinds.most_common():
    print(f'  {v / STEPS:7.1f}  {k}')
ops_ = collections.Counter()
for e in prof.events():
    if e.name in ('aten::copy_', 'aten::_to_copy', 'aten::zero_', 'aten::fill_', 'aten::clone', 'aten::contiguous', 'aten::index_select', 'aten::cat'):
        ops_[(e.name, str(e.input_shapes)[:90])] += 1
print('issuing ATen ops per step (name, input shapes):')
for (n, sh), v in ops_.most_common(40):
    print(f'  {v / STEPS:7.1f}  {n:18s} {sh}')
