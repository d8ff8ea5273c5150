#!/usr/bin/env python
"""Time the REFERENCE itself (/root/reference, imported on CPU in the build container) on the cpu_baseline workload and check
that the oracle -- the "port" that bench.py times on the GPU box, where the reference does not exist -- runs within +-15 % of it.

Workload: RVT-S Gen1, T = 21, bs = 8, 4 labelled frames per sequence, one full training step driven exactly like
Module.training_step (modules/detection.py:150-298: RNNStates / BackboneFeatureSelector / ObjectLabels, forward_detect, backward,
clip by value, AdamW, OneCycleLR), same synthetic weights and inputs for both.  1 warm-up + 2 timed steps each, best-of.
Also cross-checks the first step's loss (the oracle is pinned to the reference by tests/golden; this is the full-size version).

usage: python tools/time_reference.py [threads=8] [out=profiles/r02_m_reference_vs_oracle_cpu.txt]   (build container only)
"""
import json
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

if not os.path.isdir('/root/reference'):
    raise SystemExit('tools/time_reference.py runs in the build container only (needs /root/reference)')
import make_golden as mg  # noqa: E402  (imports the reference with the third-party stand-ins on sys.path)
import bench  # noqa: E402
from oracle import train_step as ot  # noqa: E402
from oracle.synth import synth_state_dict  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, 'profiles', 'r02_m_reference_vs_oracle_cpu.txt')
torch.set_num_threads(threads)
T, B, hw = 21, 8, (240, 304)
ev, _, label_tb, labs = bench.make_batch(T, B, hw, 2, 7, 'cpu', (4, 9, 14, 19))
it = iter(labs)
labels = []
for t in range(T):
    row = [None] * B
    for b in label_tb[t]:
        l = next(it)
        row[b] = torch.from_numpy(np.concatenate([np.ones((len(l), 1), np.float32), l[:, 1:2] - l[:, 3:4] / 2, l[:, 2:3] - l[:, 4:5] / 2,
                                                  l[:, 3:5], l[:, 0:1], l[:, 6:7], l[:, 5:6]], 1))
    labels.append(row)
first = torch.ones(B, dtype=torch.bool)
man = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'g11_manifest.json')))['small_gen1']
sd = synth_state_dict(man, 0)


class RefTrainer:
    def __init__(self):
        torch.manual_seed(0)
        self.det = mg.YoloXDetector(mg.make_cfg(embed_dim=48, dim_head=24, fpn_depth=0.33, in_hw=(256, 320), part=(8, 10)))
        self.det.load_state_dict(sd, strict=True)
        self.det.train()
        self.opt = torch.optim.AdamW(self.det.parameters(), lr=2e-4, weight_decay=0)
        self.sch = torch.optim.lr_scheduler.OneCycleLR(self.opt, max_lr=2e-4, div_factor=20, final_div_factor=500, total_steps=400000,
                                                       pct_start=0.005, cycle_momentum=False, anneal_strategy='linear')
        self.padder = mg.InputPadderFromShape(desired_hw=(256, 320))
        self.rnn = mg.RNNStates()

    def step(self):
        det = self.det
        x = self.padder.pad_tensor_ev_repr(ev.to(torch.float32))            # detection.py:132-135
        self.rnn.reset(worker_id=0, indices_or_bool_tensor=first)
        prev = self.rnn.get_states(worker_id=0)
        sel = mg.BackboneFeatureSelector()
        obj = []
        for t in range(T):
            feats, prev = det.forward_backbone(x=x[t], previous_states=prev)
            idx = [b for b in range(B) if labels[t][b] is not None]
            if idx:
                sel.add_backbone_features(backbone_features=feats, selected_indices=idx)
                obj.extend(mg.ObjectLabels(labels[t][b], hw) for b in idx)
        self.rnn.save_states_and_detach(worker_id=0, states=prev)
        targets = mg.ObjectLabels.get_labels_as_batched_tensor(obj_label_list=obj, format_='yolox')
        _, losses = det.forward_detect(backbone_features=sel.get_batched_backbone_features(), targets=targets)
        self.opt.zero_grad(set_to_none=True)
        losses['loss'].backward()
        torch.nn.utils.clip_grad_value_(det.parameters(), 1.0)
        self.opt.step()
        self.sch.step()
        return float(losses['loss'])


def best_of(fn, warm=1, n=2):
    vals, best = [], float('inf')
    for i in range(warm + n):
        t0 = time.time()
        vals.append(fn())
        if i >= warm:
            best = min(best, time.time() - t0)
    return best, vals[0]


ref = RefTrainer()
t_ref, loss_ref = best_of(ref.step)
orc = ot.OracleTrainer(sd, ot.model_cfg(48, 24, 0.33, (8, 10)))
t_orc, loss_orc = best_of(lambda: orc.step(ev, labels, first)[0]['loss'])
ratio = t_orc / t_ref
lines = [f'# tools/time_reference.py: reference (imported from /root/reference) vs oracle/ on this container\'s CPU, {threads} threads',
         f'workload: RVT-S Gen1 240x304 T={T} bs={B}, one full training step ({T * B} event-frames), 1 warm-up + 2 timed, best-of',
         f'reference: {t_ref:.2f} s/step = {T * B / t_ref:.1f} event-frames/s   first-step loss {loss_ref:.6f}',
         f'oracle   : {t_orc:.2f} s/step = {T * B / t_orc:.1f} event-frames/s   first-step loss {loss_orc:.6f}',
         f'oracle / reference step time = {ratio:.3f} (accepted band 0.85 .. 1.15); |loss diff| / loss = {abs(loss_ref - loss_orc) / abs(loss_ref):.2e}']
print('\n'.join(lines))
open(out_path, 'w').write('\n'.join(lines) + '\n')
assert abs(loss_ref - loss_orc) <= 2e-5 * abs(loss_ref), 'oracle and reference disagree on the first-step loss'
assert 0.85 <= ratio <= 1.15, f'oracle step time is {ratio:.2f}x the reference: cpu_baseline kind "port" is not representative'
