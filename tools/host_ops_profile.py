#!/usr/bin/env python
"""Which ATen ops does the launch thread run during one training step, and do any of them touch large tensors (candidates for an
OpenMP parallel region on the host)?  usage: python tools/host_ops_profile.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.argv = [sys.argv[0], '3']
import runpy
import torch
from torch.profiler import profile, ProfilerActivity
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'step_drift.py'), run_name='drift')
fit_step, mod, opt, sched, batch = ns['fit_step'], ns['mod'], ns['opt'], ns['sched'], ns['batch']
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    for s in range(2):
        fit_step(mod, opt, sched, batch(), s)
    torch.cuda.synchronize()
rows = {}
for e in prof.events():
    shapes = [s for s in (e.input_shapes or []) if s]
    big = max([int(torch.tensor(s).prod()) for s in shapes if all(isinstance(d, int) for d in s)] or [0])
    k = (e.name, str(shapes)[:120])
    r = rows.setdefault(k, [0, 0.0, big])
    r[0] += 1; r[1] += e.self_cpu_time_total
print('ops with an input of >= 16384 elements (name, shapes, calls, self cpu us):')
for k, r in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    if r[2] >= 16384 and k[0].startswith('aten::'):
        print(f'  {k[0]:40s} {k[1]:120s} {r[0]:5d} {r[1]:10.0f}')
print('top 25 ops by self cpu time:')
agg = {}
for k, r in rows.items():
    a = agg.setdefault(k[0], [0, 0.0]); a[0] += r[0]; a[1] += r[1]
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f'  {n:50s} {a[0]:6d} {a[1]:10.0f}')
