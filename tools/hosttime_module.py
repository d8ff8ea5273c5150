#!/usr/bin/env python
"""Host-side time of one training step through the Module surface, phase by phase (launch-thread milliseconds; the GPU runs
asynchronously underneath, one synchronize per step): is the step launch-bound or GPU-bound, and where does the host go?
usage: python tools/hosttime_module.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
import bench
from leod_amd.config import full_config, dynamically_modify_train_config
from leod_amd.data.genx_utils.labels import ObjectLabels, SparselyBatchedObjectLabels
from leod_amd.data.utils.types import DataType
from leod_amd.modules.utils.detection import DATA_KEY, WORKER_ID_KEY
from leod_amd.modules.utils.fetch import fetch_model_module

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda', 0)
cfg = dynamically_modify_train_config(full_config('gen1', 'small'))
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


acc = np.zeros(7)
for s in range(steps + 2):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    b = batch(); t.append(time.perf_counter())
    opt.zero_grad(); t.append(time.perf_counter())
    out = mod.training_step(b, s); t.append(time.perf_counter())
    out['loss'].backward(); t.append(time.perf_counter())
    opt.step(); sched.step(); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    if s >= 2:
        acc[:6] += np.diff(t); acc[6] += t[-1] - t[0]
acc *= 1e3 / steps
print('host ms/step: batch %.2f | zero_grad %.2f | training_step %.2f | backward %.2f | opt+sched %.2f | drain %.2f | total %.2f'
      % tuple(acc))

# free-running (no per-step synchronize): what bench.py times
from leod_amd.optim import fit_step
for mode in ('fit_step', 'manual'):
    torch.cuda.synchronize(); t0 = time.perf_counter(); marks = []
    for s in range(steps):
        if mode == 'fit_step':
            fit_step(mod, opt, sched, batch(), s)
        else:
            b = batch(); opt.zero_grad(); out = mod.training_step(b, s); out['loss'].backward(); opt.step(); sched.step()
        marks.append(time.perf_counter())
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f'free-running {mode}: {1e3 * (t1 - t0) / steps:.2f} ms/step; host per step (ms):',
          ' '.join(f'{1e3 * d:.1f}' for d in np.diff([t0] + marks)))
