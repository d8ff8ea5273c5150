#!/usr/bin/env python
"""The N > 1 code path on a single-GPU box: LEOD_FORCE_COLLECTIVES=1 makes a one-rank job create the RCCL communicator and issue
every collective of a data-parallel training step (parameter broadcast, the SyncBatchNorm statistic exchanges, the five per-stage
gradient buckets released under the backward pass -- LEOD_DP_BUCKETS=0: one flat all-reduce after it) through torch.distributed's 'nccl' backend (= RCCL on ROCm).  Each collective is bracketed with HIP events on the launch
stream; the table goes to stdout (kept under profiles/).  With one rank the collectives move no data between GPUs: what this run
shows is that the RCCL calls are issued, ordered correctly against the HIP kernels (losses equal the plain run) and how many of
them a step contains -- not xGMI timings.

    LEOD_FORCE_COLLECTIVES=1 python tools/rccl_force_log.py [steps]"""
import os, sys, time
os.environ.setdefault('LEOD_FORCE_COLLECTIVES', '1')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
import torch.distributed as dist
import bench
from leod_amd import functions as Fn
from leod_amd.parallel import init_distributed

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rank, local, world = init_distributed()
dev = torch.device('cuda', local)
print(f'backend={dist.get_backend()} world_size={dist.get_world_size()} rank={rank} (LEOD_FORCE_COLLECTIVES={os.environ["LEOD_FORCE_COLLECTIVES"]})')
log = []
_orig = dist.all_reduce


def timed_all_reduce(t, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = _orig(t, *a, **k); e1.record()
    log.append((t.numel() * t.element_size(), e0, e1))
    return r


dist.all_reduce = timed_all_reduce
from leod_amd.config import full_config, dynamically_modify_train_config
from leod_amd.data.genx_utils.labels import ObjectLabels, SparselyBatchedObjectLabels
from leod_amd.data.utils.types import DataType
from leod_amd.modules.utils.detection import DATA_KEY, WORKER_ID_KEY
from leod_amd.modules.utils.fetch import fetch_model_module
from leod_amd.optim import fit_step
cfg = dynamically_modify_train_config(full_config('gen1', 'small'))
torch.manual_seed(0)
mod = fetch_model_module(cfg).to(dev); mod.setup('fit'); mod.train()
oc = mod.configure_optimizers(); opt, sched = oc['optimizer'], oc['lr_scheduler']['scheduler']
assert opt.dp.force and Fn._sync_bn_on()
T, B, hw = 21, 8, (240, 304)
ev, _, label_tb, labs = bench.make_batch(T, B, hw, 2, 0, dev, (4, 9, 14, 19))
lab8 = [np.concatenate([np.ones((len(l), 1), np.float32), l[:, 1:2] - l[:, 3:4] / 2, l[:, 2:3] - l[:, 4:5] / 2, l[:, 3:5], l[:, 0:1], l[:, 6:7], l[:, 5:6]], 1) for l in labs]
first = torch.ones(B, dtype=torch.bool, device=dev)


def batch():
    it = iter(lab8); seq = []
    for t in range(T):
        row = [None] * B
        for b in label_tb[t]:
            row[b] = ObjectLabels(torch.from_numpy(next(it).copy()), hw)
        seq.append(SparselyBatchedObjectLabels(row))
    return {WORKER_ID_KEY: 0, DATA_KEY: {DataType.EV_REPR: [ev[t] for t in range(T)], DataType.OBJLABELS_SEQ: seq, DataType.IS_FIRST_SAMPLE: first}}


losses = []
for s in range(steps + 2):
    if s == 2:
        torch.cuda.synchronize(); log.clear(); Fn._SYNC_BN['n_collectives'] = 0; t0 = time.perf_counter()
    out = fit_step(mod, opt, sched, batch(), s)
    losses.append(out['loss'].detach())
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
us = np.array([a.elapsed_time(b) * 1e3 for _, a, b in log]); size = np.array([n for n, _, _ in log])
print(f'{steps} steps, {1e3 * dt:.2f} ms/step with every collective issued; losses: ' + ' '.join(f'{float(l):.4f}' for l in losses))
print(f'collectives per step: {len(log) / steps:.1f} all-reduce ({Fn._SYNC_BN["n_collectives"] / steps:.1f} SyncBatchNorm + {(len(log) / steps - Fn._SYNC_BN["n_collectives"] / steps):.0f} gradient)')
bsizes = set(4 * (r[1] - r[0]) for r in opt.dp.buckets.ranges) if opt.dp.buckets is not None else {4 * opt.flat.numel}
isg = np.array([n in bsizes for n in size])
for name, m in (('SyncBatchNorm statistics', ~isg), ('gradient buckets (head+PAFPN, stage 4..1)' if opt.dp.buckets is not None else 'flat gradient', isg)):
    if m.any():
        print(f'  {name:38s} n/step {m.sum() / steps:5.1f}  bytes {size[m].min()}..{size[m].max()}  event-bracketed us: mean {us[m].mean():.1f} '
              f'min {us[m].min():.1f} max {us[m].max():.1f}  total/step {us[m].sum() / steps / 1e3:.3f} ms')
b = opt.dp.buckets
if b is not None:
    print('bucket release order of the last step:', b.order, ' bytes per bucket:', [4 * (r[1] - r[0]) for r in b.ranges])
dist.barrier(); dist.destroy_process_group()
