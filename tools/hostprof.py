#!/usr/bin/env python
"""cProfile of the host side of one eager training step (where do the ~15 us per launch go?)."""
import cProfile, os, pstats, sys
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
ev, labels, label_tb, _ = bench.make_batch(21, 8, (240, 304), 2, 0, dev, (4, 9, 14, 19))
first = torch.zeros(8, dtype=torch.bool, device=dev)
for _ in range(3):
    eng.step(ev, labels, label_tb, first)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    eng.step(ev, labels, label_tb, first)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(45)
