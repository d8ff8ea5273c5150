#!/usr/bin/env python
"""Per-step wall time of the benchmarked training step over a long run (one synchronize per step): does the step time drift
(clock / power management) or is it flat?  Prints every step's ms plus the sclk / power rocm-smi reports every 10 steps.
usage: python tools/step_drift.py [steps] [bf16|f32]"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
import bench
from leod_amd.config import full_config, dynamically_modify_train_config
from leod_amd.data.genx_utils.labels import ObjectLabels, SparselyBatchedObjectLabels
from leod_amd.data.utils.types import DataType
from leod_amd.modules.utils.detection import DATA_KEY, WORKER_ID_KEY
from leod_amd.modules.utils.fetch import fetch_model_module
from leod_amd.optim import fit_step

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dtype = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
dev = torch.device('cuda', 0)
cfg = dynamically_modify_train_config(full_config('gen1', 'small'))
cfg.training.precision = 16 if dtype == 'bf16' else 32
torch.manual_seed(0)
mod = fetch_model_module(cfg).to(dev)
mod.setup('fit'); mod.train()
oc = mod.configure_optimizers()
opt, sched = oc['optimizer'], oc['lr_scheduler']['scheduler']
T, B, hw = 21, 8, (240, 304)
ev, _, label_tb, labs = bench.make_batch(T, B, hw, 2, 0, dev, (4, 9, 14, 19))
lab8 = [np.concatenate([np.ones((len(l), 1), np.float32), l[:, 1:2] - l[:, 3:4] / 2, l[:, 2:3] - l[:, 4:5] / 2, l[:, 3:5],
                        l[:, 0:1], l[:, 6:7], l[:, 5:6]], 1) for l in labs]
first = torch.ones(B, dtype=torch.bool, device=dev)


def batch():
    it = iter(lab8); seq = []
    for t in range(T):
        row = [None] * B
        for b in label_tb[t]:
            row[b] = ObjectLabels(torch.from_numpy(next(it).copy()), hw)
        seq.append(SparselyBatchedObjectLabels(row))
    return {WORKER_ID_KEY: 0, DATA_KEY: {DataType.EV_REPR: [ev[t] for t in range(T)], DataType.OBJLABELS_SEQ: seq, DataType.IS_FIRST_SAMPLE: first}}


def smi():
    try:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp'], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in out.splitlines() if any(k in l for k in ('sclk', 'Power', 'junction', 'mclk'))]
        return ' | '.join(keep)
    except Exception as e:   # noqa: BLE001
        return f'rocm-smi failed: {e}'


import gc
gcmode = os.environ.get('GCMODE', 'default')
if gcmode == 'off':
    gc.disable()
elif gcmode == 'freeze':
    gc.collect(); gc.freeze()
print('gc mode', gcmode, 'thresholds', gc.get_threshold(), 'tracked objects', len(gc.get_objects()))
g0 = [dict(d) for d in gc.get_stats()]
def cpustat():
    try:
        d = dict(l.split() for l in open('/sys/fs/cgroup/cpu.stat').read().splitlines())
        return {k: int(d[k]) for k in ('usage_usec', 'nr_periods', 'nr_throttled', 'throttled_usec') if k in d}
    except OSError as e:
        return {'err': str(e)}
print('cpu.max', open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else None,
      'affinity', len(os.sched_getaffinity(0)), 'torch threads', torch.get_num_threads())
c0, p0, w0 = cpustat(), os.times(), time.perf_counter()
times, host = [], []
for s in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fit_step(mod, opt, sched, batch(), s)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    times.append(1e3 * (t2 - t0)); host.append(1e3 * (t1 - t0))
    if s % 10 == 9:
        print(f'step {s}: ', smi(), flush=True)
c1, p1, w1 = cpustat(), os.times(), time.perf_counter()
print('cgroup cpu.stat delta:', {k: c1[k] - c0[k] for k in c0 if k in c1 and isinstance(c0[k], int)}, f'wall {w1 - w0:.2f} s, process user {p1.user - p0.user:.2f} s sys {p1.system - p0.system:.2f} s')
print('gc collections during the run (gen0, gen1, gen2):', [b['collections'] - a['collections'] for a, b in zip(g0, gc.get_stats())])
print(dtype, 'per-step ms (synced):', ' '.join(f'{t:.1f}' for t in times))
print(dtype, 'host enqueue ms     :', ' '.join(f'{t:.1f}' for t in host))
torch.cuda.synchronize(); t0 = time.perf_counter()
for s in range(20):
    fit_step(mod, opt, sched, batch(), s)
torch.cuda.synchronize()
print(dtype, f'free-running 20 steps: {1e3 * (time.perf_counter() - t0) / 20:.2f} ms/step')
