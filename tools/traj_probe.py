#!/usr/bin/env python
"""200-step OneCycle trajectory of the benchmark workload (as tests/test_bf16_class_gpu.py::test_loss_trajectory_200_steps...) in ONE
precision mode: prints the last-20-step mean loss.  usage: traj_probe.py [bf16|f32]   (environment switches select the variant)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import test_bf16_class_gpu as tc
import test_engine_gpu as te
from leod_amd.config import full_config, dynamically_modify_train_config
from leod_amd.modules.utils.fetch import fetch_model_module
from leod_amd.optim import fit_step
mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
T, B, steps = 21, 8, 200
batches = [tc._device_batch(T, B, 300 + i) for i in range(16)]
g = torch.Generator().manual_seed(6)
firsts = [torch.ones(B, dtype=torch.bool)]
for s in range(1, steps):
    m = torch.ones(B, dtype=torch.bool); m[:B // 2] = torch.rand(B // 2, generator=g) < 0.05; firsts.append(m)
cfg = dynamically_modify_train_config(full_config('gen1', 'small'))
cfg.training.max_steps = steps; cfg.training.lr_scheduler.total_steps = steps; cfg.training.lr_scheduler.pct_start = 0.1
torch.manual_seed(0)
mod = fetch_model_module(cfg).to('cuda'); mod.setup('fit'); mod.train()
oc = mod.configure_optimizers(); opt, lrs = oc['optimizer'], oc['lr_scheduler']['scheduler']
# TRAJ_GRAD_NOISE=<rel>: before every optimiser step each parameter tensor's gradient gets isotropic Gaussian noise of relative norm <rel>
# (the bf16 mode's gradient differs from the fp32 one by 1 - cos = 0.06-0.08, i.e. a relative error norm of ~0.37, at this size: is the lower
# end-of-schedule loss of the bf16 mode what unbiased gradient noise of that size does to these 200 steps?)
rel = float(os.environ.get('TRAJ_GRAD_NOISE', '0'))
if rel > 0:
    gen = torch.Generator(device='cuda').manual_seed(int(os.environ.get('TRAJ_NOISE_SEED', '1')))
    flat = opt.flat
    sizes = [p.numel() for p in flat.params]
    orig_step = opt.step

    def noisy_step(*a, **k):
        for o, n in zip(flat.offsets, sizes):
            g_ = flat.grad[o:o + n]
            g_.add_(torch.randn(n, generator=gen, device='cuda') * (rel * float(g_.norm()) / max(n, 1) ** 0.5))
        return orig_step(*a, **k)
    opt.step = noisy_step
out = []
with tc.precision(mode):
    for s in range(steps):
        ev, lab, label_tb = batches[s % 16]
        res = fit_step(mod, opt, lrs, te._loader_batch(ev, lab, label_tb, firsts[s].to('cuda')), s)
        out.append(res['log_dict']['train/loss'].detach())
a = torch.stack(out).cpu().numpy()
# the trained weights judged in ONE arithmetic: mean training-mode loss of the 16 batches in fp32 mode (fresh LSTM state for every batch)
ev_losses = []
mod.plan_mode = False
with tc.precision('f32'), torch.no_grad():
    for i in range(16):
        ev, lab, label_tb = batches[i]
        res = mod.training_step(te._loader_batch(ev, lab, label_tb, torch.ones(B, dtype=torch.bool).to('cuda')), i, log=False)
        ev_losses.append(float(res['loss']))
print(f'   final weights evaluated in fp32 mode on the 16 batches (fresh state): mean loss {np.mean(ev_losses):.3f}', flush=True)
print(f'{mode} noise={rel} plan={mod.plan_mode} replays={mod._plans.replays} lanes={mod._plans.max_lanes}: last-20 mean {a[-20:].mean():.3f}  steps 100-120 {a[100:120].mean():.3f}  140-160 {a[140:160].mean():.3f}', flush=True)
