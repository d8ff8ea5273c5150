#!/usr/bin/env python
"""Throughput of the pseudo-label inference pass (BASELINE.json configs[4] shape on one GPU): RVT-S, Gen1 240x304,
T=21, bs=8 streams + hflip TTA (16 frame streams), backbone -> head -> batched NMS -> pseudo-label filters, all on the
device.  Prints event-frames/s (source frames, i.e. without counting the flipped copies)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from leod_amd.config import full_config, dynamically_modify_train_config
from leod_amd.models.detection.yolox_extension.models.detector import YoloXDetector
from leod_amd.engine import PseudoLabelEngine

from leod_amd import ops
ops.set_precision(os.environ.get('LEOD_PRECISION', 'bf16'))      # the engine-level driver does not go through Module.setup
print('precision mode', ops.get_precision())
dev = torch.device('cuda', 0)
cfg = dynamically_modify_train_config(full_config('gen1', 'small'))
torch.manual_seed(0)
det = YoloXDetector(cfg.model).to(dev)
T, B = 21, 8
ev, _, _, _ = bench.make_batch(T, B, (240, 304), 2, 0, dev, (4, 9, 14, 19))
for mode in (False, True):
    eng = PseudoLabelEngine(det, 2, conf_thre=0.01, hflip=True, max_det=256)
    eng.time_batched = mode
    first = torch.zeros(B, dtype=torch.bool, device=dev)
    for _ in range(3):
        eng.step(ev, first)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        out = eng.step(ev, first)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f'time_batched={mode}: {1e3 * dt:.1f} ms per [T=21, B=8 (+hflip)] chunk = {T * B / dt:.0f} event-frames/s '
          f'({int(out[3].sum())} raw detections, {int(out[1].sum())} pseudo labels)')
