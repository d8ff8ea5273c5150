#!/usr/bin/env python
"""Does the REFERENCE's own 16-bit arithmetic class move a training trajectory the way the HIP 16-bit modes do?

tests/test_bf16_class_gpu.py::test_loss_trajectory_200_steps_onecycle_bf16_vs_f32 finds that the 16-bit modes follow the fp32 curve
through warm-up and most of the decay and then, while the cycled batches are being fitted, tend to end LOWER with a wider run-to-run
spread.  The GPU cannot run the reference; this script runs the reference detector itself (imported from /root/reference, build
container only -- nothing here travels to the GPU box) on the CPU at the micro geometry of the golden fixtures, through a whole
OneCycle schedule of 200 AdamW steps on 16 cycled batches, in

  fp32            the reference as it is
  fp32 + eps      the same from initial weights perturbed once by relative Gaussian noise of 2^-9 / 2^-12 (the chaos control:
                  how far does a perturbation the size of one 16-bit rounding move the END of the schedule)
  h16f            torch.autocast(float16) with CUDA autocast's fp32-op placement and GradScaler-style loss scaling -- the class the
                  reference trains in (Lightning precision=16, train.py:236-243)
  acf             the same placement in bfloat16

and prints the loss every 20 steps, the last-20-step means and the smoothed relative differences to the fp32 run.
usage: python tools/ref_trajectory_cpu.py [steps=200] [out.txt]"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', 'tests', 'golden'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import make_golden as mg  # noqa: E402  (sets up the reference imports and the stand-ins of absent third-party packages)

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 200
OUT = sys.argv[2] if len(sys.argv) > 2 else None
T, B, NB = 5, 2, 16
HW, PAD = (60, 90), (64, 96)


def run(mode, _data, perturb=0.0, seed=0):
    return mg._ref_trajectory(mode, STEPS, perturb, seed, T=T, B=B, nb=NB, hw=HW, pad=PAD)


def main():
    torch.set_num_threads(8)
    data = None
    runs = [('fp32', 'fp32', 0.0, 0), ('fp32 + 2^-12 (a)', 'fp32', 2.0 ** -12, 1), ('fp32 + 2^-12 (b)', 'fp32', 2.0 ** -12, 2),
            ('fp32 + 2^-9 (a)', 'fp32', 2.0 ** -9, 3), ('fp32 + 2^-9 (b)', 'fp32', 2.0 ** -9, 4),
            ('h16f', 'h16f', 0.0, 0), ('h16f + 2^-12', 'h16f', 2.0 ** -12, 1), ('acf (bf16)', 'acf', 0.0, 0), ('acf + 2^-12', 'acf', 2.0 ** -12, 1)]
    lines = [f'# reference detector (micro geometry {PAD}, T={T}, B={B}), {STEPS} AdamW steps of one OneCycle schedule (peak 2e-4, pct_start 0.1) on {NB} cycled batches, CPU']
    traj = {}
    for name, mode, eps, seed in runs:
        t0 = time.time()
        traj[name] = run(mode, data, eps, seed)
        lines.append(f'{name:<20s} every 20th: {np.round(traj[name][::20], 3).tolist()}  first/last-20 mean {traj[name][:20].mean():.3f} / {traj[name][-20:].mean():.3f}   ({time.time() - t0:.0f} s)')
        print(lines[-1], flush=True)
    sm = lambda x: np.convolve(x, np.ones(10) / 10, mode='valid')  # noqa: E731
    base = sm(traj['fp32'])
    lines.append('# smoothed (10-step) relative difference to the fp32 run: max before step 100 / max overall / at the end')
    for name in traj:
        if name == 'fp32':
            continue
        d = np.abs(sm(traj[name]) - base) / base
        lines.append(f'{name:<20s} {d[:90].max():.4f} / {d.max():.4f} / {d[-1]:.4f}   final-20 mean relative to fp32: {traj[name][-20:].mean() / traj["fp32"][-20:].mean():.4f}')
        print(lines[-1], flush=True)
    if OUT:
        with open(OUT, 'w') as f:
            f.write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()
