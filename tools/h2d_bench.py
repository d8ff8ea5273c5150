#!/usr/bin/env python
"""PCIe-inclusive rate of the training step: every step consumes a NEW uint8 batch that starts in host
memory.  Compares (a) inputs resident in HBM (bench.py's contract), (b) a blocking pageable .to(device) per step (what Lightning
does for the reference), (c) leod_amd.engine.HostFeeder (pinned double buffer, copy stream overlapped with the previous step).
usage: python tools/h2d_bench.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

import bench  # noqa: E402
from leod_amd.config import full_config, dynamically_modify_train_config  # noqa: E402
from leod_amd.engine import TrainEngine, HostFeeder  # noqa: E402
from leod_amd.models.detection.yolox_extension.models.detector import YoloXDetector  # noqa: E402


def main(steps=8):
    dev = torch.device('cuda', 0)
    cfg = dynamically_modify_train_config(full_config('gen1', 'small'))
    torch.manual_seed(0)
    eng = TrainEngine(YoloXDetector(cfg.model).to(dev), lr=cfg.training.learning_rate, total_steps=400000)
    T, B = 21, 8
    ev, labels, label_tb, _ = bench.make_batch(T, B, (240, 304), 2, 0, dev, (4, 9, 14, 19))
    first = torch.ones(B, dtype=torch.bool, device=dev)
    host_batches = [ev.cpu().roll(k, dims=1).contiguous() for k in range(2)]          # pageable host tensors
    mb = ev.numel() / 1e6
    pinned = host_batches[0].pin_memory()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ev.copy_(pinned, non_blocking=True)
    torch.cuda.synchronize()
    print(f'batch {mb:.1f} MB uint8; pinned H2D {5 * mb / 1e3 / (time.perf_counter() - t0):.1f} GB/s')

    def timed(fn, name):
        for s in range(2):
            fn(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(steps):
            fn(2 + s)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(f'{name:58s} {1e3 * dt:7.2f} ms/step  {T * B / dt:8.1f} event-frames/s')
        return dt

    timed(lambda s: eng.step(ev, labels, label_tb, first), '(a) inputs resident in HBM')
    timed(lambda s: eng.step(host_batches[s % 2].to(dev), labels, label_tb, first), '(b) blocking pageable .to(device) every step')
    feeder = HostFeeder(ev.shape, dev)
    feeder.put(host_batches[0])

    def fed(s):
        x = feeder.get()
        feeder.put(host_batches[(s + 1) % 2])            # stage the next batch while this step runs
        eng.step(x, labels, label_tb, first)
        feeder.done()
    timed(fed, '(c) HostFeeder, pageable batches (staging memcpy on the launch thread)')
    pinned_batches = [b.pin_memory() for b in host_batches]
    feeder = HostFeeder(ev.shape, dev)
    feeder.put(pinned_batches[0])

    def fed_pinned(s):
        x = feeder.get()
        feeder.put(pinned_batches[(s + 1) % 2])
        eng.step(x, labels, label_tb, first)
        feeder.done()
    timed(fed_pinned, '(d) HostFeeder, pinned batches (DataLoader pin_memory): copy stream overlapped')


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
