#!/usr/bin/env python
"""Host enqueue time vs GPU completion time of one eager training step (is the step launch-bound or GPU-bound?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from leod_amd.config import full_config, dynamically_modify_train_config
from leod_amd.models.detection.yolox_extension.models.detector import YoloXDetector
from leod_amd.engine import TrainEngine

dev = torch.device('cuda', 0)
cfg = dynamically_modify_train_config(full_config('gen1', 'small'))
torch.manual_seed(0)
eng = TrainEngine(YoloXDetector(cfg.model).to(dev), lr=cfg.training.learning_rate)
T, B = 21, 8
ev, labels, label_tb, _ = bench.make_batch(T, B, (240, 304), 2, 0, dev, (4, 9, 14, 19))
first = torch.zeros(B, dtype=torch.bool, device=dev)
for _ in range(3):
    eng.step(ev, labels, label_tb, first)
torch.cuda.synchronize()
for it in range(4):
    t0 = time.perf_counter()
    eng.step(ev, labels, label_tb, first)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'enqueue {1e3 * (t1 - t0):.1f} ms, drain {1e3 * (t2 - t1):.1f} ms, total {1e3 * (t2 - t0):.1f} ms')
