#!/usr/bin/env python
"""cProfile of PseudoLabeler.predict_step on the bench workload (16 source streams + hflip, L = 21): where does the host time go?"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from leod_amd.config import full_config, dynamically_modify_train_config
from leod_amd.data.genx_utils.labels import SparselyBatchedObjectLabels
from leod_amd.data.utils.types import DataType
from leod_amd.modules.pseudo_labeler import PseudoLabeler
from leod_amd.modules.utils.detection import DATA_KEY, WORKER_ID_KEY
L, B = 21, 16
dev = torch.device('cuda', 0)
cfg = dynamically_modify_train_config(full_config('gen1', 'small', model='pseudo_labeler', overrides=dict(dataset=dict(sequence_length=L), tta=dict(enable=True, hflip=True, tflip=False))))
cfg.training.precision = 16
cfg.model.postprocess.confidence_threshold = 0.01
torch.manual_seed(0)
mod = PseudoLabeler(cfg).to(dev).eval(); mod.setup('predict'); mod.pipelined = True
with torch.no_grad():
    for k in range(3):
        mod.mdl.yolox_head.obj_preds[k].bias += 4.0; mod.mdl.yolox_head.cls_preds[k].bias += 4.0
ev, _, _, _ = bench.make_batch(L, B, (240, 304), 2, 0, dev, ())
none_seq = lambda: [SparselyBatchedObjectLabels([None] * B) for _ in range(L)]
step = [0]
def batch():
    s = step[0]; step[0] += 1
    return {WORKER_ID_KEY: 0, DATA_KEY: {DataType.EV_REPR: [ev[t] for t in range(L)], DataType.OBJLABELS_SEQ: none_seq(), DataType.SKIPPED_OBJLABELS_SEQ: none_seq(),
            DataType.IS_FIRST_SAMPLE: torch.full((B,), s == 0), DataType.IS_LAST_SAMPLE: torch.zeros(B, dtype=torch.bool), DataType.IS_REVERSED: torch.zeros(B, dtype=torch.bool),
            DataType.EV_IDX: [torch.full((B,), L * s + t, dtype=torch.long) for t in range(L)], DataType.IS_PADDED_MASK: [torch.zeros(B, dtype=torch.bool) for _ in range(L)],
            DataType.PATH: [f'train/rec_{b}' for b in range(B)]}}
for _ in range(3):
    mod.predict_step(batch(), 0)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(3):
    mod.predict_step(batch(), 0)
torch.cuda.synchronize()
print('ms per step', 1e3 * (time.perf_counter() - t0) / 3)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
pstats.Stats(pr).sort_stats('tottime').print_stats(25)
