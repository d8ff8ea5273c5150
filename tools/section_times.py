#!/usr/bin/env python
"""GPU time of the sections of one training step on the launch stream (HIP events): backbone forward, PAFPN + head + loss forward,
PAFPN + head backward, backbone backward (incl. the join of the weight-gradient stream), optimiser.  usage: python tools/section_times.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.argv = [sys.argv[0], '3']
import runpy
import torch
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'step_drift.py'), run_name='drift')
fit_step, mod, opt, sched, batch = ns['fit_step'], ns['mod'], ns['opt'], ns['sched'], ns['batch']
ev = {}


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    ev.setdefault(name, []).append(e)


orig_fd = mod.mdl.forward_detect


def forward_detect(backbone_features, targets=None, soft_targets=None):
    mark('fd_begin')
    feats = {}
    for k, v in backbone_features.items():
        if v.requires_grad:
            v = v.view_as(v)
            v.register_hook(lambda g, k=k: mark(f'bwd_feat_{k}'))
        feats[k] = v
    out = orig_fd(feats, targets, soft_targets)
    mark('fd_end')
    return out


mod.mdl.forward_detect = forward_detect
orig_bwd = mod.backward


def backward(loss, *a, **k):
    mark('bwd_begin')
    orig_bwd(loss, *a, **k)
    mark('bwd_end')


mod.backward = backward
for s in range(8):
    ev.clear()
    mark('step_begin')
    fit_step(mod, opt, sched, batch(), s)
    mark('step_end')
torch.cuda.synchronize()
t = lambda a, b: ev[a][0].elapsed_time(ev[b][-1])
last_feat = max((k for k in ev if k.startswith('bwd_feat_')), key=lambda k: ev['bwd_begin'][0].elapsed_time(ev[k][-1]))
print(f'step {t("step_begin", "step_end"):.2f} ms = backbone fwd {t("step_begin", "fd_begin"):.2f} + PAFPN/head/loss fwd {t("fd_begin", "fd_end"):.2f} '
      f'+ PAFPN/head bwd {ev["bwd_begin"][0].elapsed_time(ev[last_feat][-1]):.2f} + backbone bwd {ev[last_feat][-1].elapsed_time(ev["bwd_end"][0]):.2f} '
      f'+ join/all-reduce/AdamW {t("bwd_end", "step_end"):.2f}')
