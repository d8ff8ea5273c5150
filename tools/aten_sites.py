#!/usr/bin/env python
"""Where do the ATen device ops of one training step come from?  Groups every aten:: op that launches device work (copy_, fill_,
zero_, add, cat, index_select, ...) by the innermost leod_amd / tests source line that issued it.  usage: python tools/aten_sites.py"""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for s in range(STEPS):
        fit_step(mod, opt, sched, batch(), 2 + s)
    torch.cuda.synchronize()
sites = collections.Counter()
for e in prof.events():
    if not e.name.startswith('aten::'):
        continue
    # leaf ops that own device kernels
    if not e.kernels:
        continue
    site = next((f for f in (e.stack or []) if 'leod_amd' in f or 'bench.py' in f or 'tools/' in f), '?')
    sites[(e.name, site.strip()[-110:], len(e.kernels))] += 1
print(f'{"per step":>8}  op / kernels / site')
tot = 0
for (name, site, nk), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    tot += n * nk
    print(f'{n / STEPS:8.1f}  {name:28s} k={nk}  {site}')
print('device kernels / copies launched by ATen ops per step:', tot / STEPS)
