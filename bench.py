#!/usr/bin/env python
"""LEOD hot-path benchmark on MI355X: RVT-S training step on synthetic Gen1-shaped voxel sequences
(BASELINE.json configs[1]: 240x304 uint8 voxels padded to 256x320, T=21, bs=8 per GPU, 4 labelled
frames per sequence), event-frames/s aggregated over all ranks.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one full training step with inputs already resident in HBM: LSTM-row reset, 21 backbone
timesteps, PAFPN + head + SimOTA + losses on the 32 labelled frames, backward, gradient all-reduce (RCCL),
value-clip + AdamW, OneCycle LR.  Nothing is skipped or cached.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_HBM_GBS = 8000.0           # MI355X HBM3E peak (MI355X_MICROARCH.md)
PEAK_F32_MFMA_TFLOPS = 157.3    # dense fp32 MFMA peak: 256 FLOP/clk/CU x 256 CUs x 2.4 GHz (MI355X_MICROARCH.md)
# SURVEY 8d / DESIGN.md: algorithmic HBM traffic of one RVT-S training event-frame at fp32 activations
# = 2 x the bf16 figure for activations (2 x (68.61 + 32/168*29.01)) + 395/168 MB optimiser traffic
ALGO_MB_PER_FRAME_FP32 = 2 * (68.61 + 32.0 / 168.0 * 29.01) + 395.0 / 168.0


def make_batch(T, B, hw, num_classes, seed, device, label_ts):
    """Synthetic inputs of SURVEY 8d: sparse small-count uint8 voxels, 1-6 boxes on labelled frames."""
    g = torch.Generator(device='cpu').manual_seed(1000 + seed)
    H, W = hw
    mask = torch.rand((T, B, 20, H, W), generator=g) < 0.08
    ev = (mask * torch.randint(1, 10, (T, B, 20, H, W), generator=g)).to(torch.uint8)
    rng = np.random.RandomState(2000 + seed)
    labs, label_tb = [], []
    for t in range(T):
        if t in label_ts:
            label_tb.append(list(range(B)))
            for _ in range(B):
                n = rng.randint(1, 7)
                w = rng.uniform(10, 90, n)
                h = rng.uniform(10, 70, n)
                x = rng.uniform(0, W - 1 - w)
                y = rng.uniform(0, H - 1 - h)
                c = rng.randint(0, num_classes, n)
                labs.append(np.stack([c, x + w / 2, y + h / 2, w, h, np.ones(n), np.ones(n)], 1).astype(np.float32))
        else:
            label_tb.append([])
    nmax = max(len(l) for l in labs)
    tg = np.zeros((len(labs), nmax, 7), np.float32)
    for i, l in enumerate(labs):
        tg[i, :len(l)] = l
    return ev.to(device), torch.from_numpy(tg).to(device), label_tb, labs


def usable_cores():
    """Host cores this process may actually use: min(affinity mask, cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(sample_B=8, T=21, threads=None):
    """The oracle (CPU restatement of the reference, torch fp32 autograd) on a bounded sample of the same workload."""
    from oracle import train_step as ot
    from oracle.synth import synth_state_dict
    import json as _json
    threads = threads or usable_cores()
    torch.set_num_threads(threads)
    man = _json.load(open(os.path.join(ROOT, 'tests', 'golden', 'g11_manifest.json')))['small_gen1']
    tr = ot.OracleTrainer(synth_state_dict(man, 0), ot.model_cfg(48, 24, 0.33, (8, 10)))
    ev, _, label_tb, labs = make_batch(T, sample_B, (240, 304), 2, 7, 'cpu', (4, 9, 14, 19))
    it = iter(labs)
    labels = []
    for t in range(T):
        row = [None] * sample_B
        for b in label_tb[t]:
            l = next(it)
            row[b] = torch.from_numpy(np.concatenate([np.ones((len(l), 1), np.float32), l[:, 1:2] - l[:, 3:4] / 2,
                                                      l[:, 2:3] - l[:, 4:5] / 2, l[:, 3:5], l[:, 0:1], l[:, 6:7], l[:, 5:6]], 1))
        labels.append(row)
    t0 = time.time()
    tr.step(ev, labels, torch.ones(sample_B, dtype=torch.bool))
    dt = time.time() - t0
    return dict(value=round(sample_B * T / dt, 3), unit='event-frames/s', cores=threads, kind='port',
                sample=f'oracle (PyTorch-CPU fp32 restatement of the reference) RVT-S Gen1 T={T} bs={sample_B}, '
                       f'1 full training step = {sample_B * T} event-frames in {dt:.1f} s')


def cpu_baseline_bounded(timeout_s=240):
    """Run the CPU leg in a child process with a hard time limit so that bench.py always finishes within minutes."""
    import subprocess
    code = ('import json, sys; sys.path.insert(0, %r); import bench; '
            'print("CPUBASE " + json.dumps(bench.cpu_baseline()))' % ROOT)
    try:
        out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=timeout_s,
                             env=dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')).stdout
        for line in out.splitlines():
            if line.startswith('CPUBASE '):
                return json.loads(line[8:])
    except subprocess.TimeoutExpired:
        pass
    return dict(value=None, unit='event-frames/s', cores=usable_cores(), kind='port',
                sample=f'CPU leg did not finish within {timeout_s} s on this host')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--seq-len', type=int, default=21)
    ap.add_argument('--size', default='small')
    ap.add_argument('--dataset', choices=('gen1', 'gen4'), default='gen1', help='gen4: 3 classes, 360x640 frames (downsampled by 2)')
    ap.add_argument('--full-res', action='store_true', help='gen4 at 720x1280 -> 768x1280, 240-token partitions '
                    '(BASELINE configs[3]: --dataset gen4 --full-res --size base --seq-len 11 --batch 2)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--graph', action='store_true', help='replay one single-stream hipGraph per step instead of the default eager '
                    'launch with the 4-stream stage wavefront (LEOD_GRAPH=1 does the same)')
    ap.add_argument('--no-graph', action='store_true', help=argparse.SUPPRESS)        # old flag, now the default
    ap.add_argument('--launch', choices=('cells', 'eager', 'graph'), default=os.environ.get('LEOD_LAUNCH', 'eager'),
                    help='cells: per-(stage,timestep) hipGraphs replayed as a 4-stream wavefront (leod_amd/cellgraph.py); '
                         'eager: one Python launch per kernel, same wavefront; graph: one single-stream hipGraph per step')
    args = ap.parse_args()

    from leod_amd.parallel import init_distributed
    rank, local, world = init_distributed()
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback for the HIP path)')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    import torch.distributed as dist
    from leod_amd.config import full_config, dynamically_modify_train_config
    from leod_amd.models.detection.yolox_extension.models.detector import YoloXDetector
    from leod_amd.cellgraph import CellGraphEngine as TrainEngine      # TrainEngine + the per-cell hipGraph scheduler
    from leod_amd import ops

    over = dict(dataset=dict(downsample_by_factor_2=not args.full_res)) if args.dataset == 'gen4' else {}
    cfg = dynamically_modify_train_config(full_config(args.dataset, args.size, overrides=over))
    hw = (240, 304) if args.dataset == 'gen1' else ((720, 1280) if args.full_res else (360, 640))
    in_hw = tuple(cfg.model.backbone.in_res_hw)
    headline = args.dataset == 'gen1' and args.size == 'small'       # the configuration BASELINE.json's metric is quoted on
    torch.manual_seed(0)                                  # identical random-init weights on every rank
    det = YoloXDetector(cfg.model).to(dev)
    eng = TrainEngine(det, lr=cfg.training.learning_rate, weight_decay=cfg.training.weight_decay,
                      total_steps=cfg.training.lr_scheduler.total_steps, pct_start=cfg.training.lr_scheduler.pct_start,
                      div_factor=cfg.training.lr_scheduler.div_factor,
                      final_div_factor=cfg.training.lr_scheduler.final_div_factor,
                      clip_value=cfg.training.gradient_clip_val)
    T, B = args.seq_len, args.batch
    label_ts = tuple(t for t in (4, 9, 14, 19) if t < T) or (T - 1,)
    ev, labels, label_tb, _ = make_batch(T, B, hw, cfg.model.head.num_classes, rank, dev, label_ts)
    g = torch.Generator(device='cpu').manual_seed(77 + rank)

    def first_mask(step):
        m = torch.ones(B, dtype=torch.bool)
        if step > 0:                                      # stream half carries state, random half always restarts
            m[:B // 2] = torch.rand(B // 2, generator=g) < 0.05
        return m.to(dev)

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    # Schedule (engine.schedule, default 'batched'): stage-major, every stage processes all T timesteps per launch; only
    # the ConvLSTM recurrence is unrolled over t.  Launch modes on top of it (identical kernels, parity between them is
    # tested in tests/test_engine_gpu.py):
    #   eager (default): one Python launch per kernel (~900 launches per step: GPU-bound).
    #   graph          : the whole step as one hipGraph (same speed on one GPU; no RCCL inside).
    #   cells          : per-(stage,timestep) hipGraphs replayed as a 4-stream wavefront (leod_amd/cellgraph.py) -- the
    #                    best launch mode for the timestep-major schedule, kept for comparison.
    launch = 'graph' if (args.graph or os.environ.get('LEOD_GRAPH') == '1') else ('eager' if args.no_graph else args.launch)
    use_graph = launch == 'graph'
    if launch == 'graph':
        eng.step(ev, labels, label_tb, first_mask(0))                  # one eager step builds the LSTM states
        eng.capture(ev, labels, label_tb, first_mask(1))
        run = lambda m: eng.step_graph(None, None, m)                  # noqa: E731  (inputs already in the static buffers)
    elif launch == 'cells':
        eng.step(ev, labels, label_tb, first_mask(0))
        eng.build(ev, labels, label_tb, first_mask(1))
        run = lambda m: eng.step_cells(None, None, m)                  # noqa: E731
    else:
        run = lambda m: eng.step(ev, labels, label_tb, m)              # noqa: E731
    for s in range(args.warmup):
        run(first_mask(1 + s))
    masks = [first_mask(1 + args.warmup + s) for s in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        losses = run(masks[s])
    barrier()
    dt = time.perf_counter() - t0
    t_max = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist.is_initialized():
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    dt = float(t_max)
    # roofline of the dominant kernel: a few extra eager steps with HIP events around each of its launches
    roofline = None
    if not args.no_roofline:
        # EVERY rank runs the two probe steps (a step contains collectives: a rank-0-only step would dead-lock N > 1);
        # only rank 0 brackets the kernel with events and reports
        probe = ops.KernelProbe() if rank == 0 else None
        # isolated launches: no co-running kernels inside the event bracket (single stream, wgrad on the launch stream)
        n_streams, eng.n_streams = eng.n_streams, 1
        side, eng.wgrad_side = eng.wgrad_side, False
        for s in range(2):
            eng.step(ev, labels, label_tb, first_mask(1))
        eng.n_streams, eng.wgrad_side = n_streams, side
        if probe is not None:
            roofline = probe.finish(PEAK_HBM_GBS, PEAK_F32_MFMA_TFLOPS)
        # HBM bytes per launch of the same kernel from the PMC passes kept under profiles/ (FETCH_SIZE x2 + WRITE_SIZE,
        # separate rocprofv3 --pmc runs; bench.py itself cannot sample hardware counters)
        tpath = os.path.join(ROOT, 'profiles', 'roofline_traffic.json')
        if rank == 0 and roofline is not None and os.path.exists(tpath):
            roofline['traffic'] = json.load(open(tpath)).get('hbm_bytes_per_launch')
    barrier()
    loss_val = float(losses['loss'])

    if rank == 0:
        frames = world * B * T * args.steps
        fps = frames / dt
        out = {
            'metric': 'event-frames/sec/GPU (RVT-S train, Gen1 T=21) at 1/2/4/8 GPUs; mAP@0.5 parity',
            'value': round(fps, 2), 'unit': 'event-frames/s (whole job)', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(1000 * dt / args.steps, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'RVT-{args.size} {args.dataset} {hw[0]}x{hw[1]} (pad {in_hw[0]}x{in_hw[1]}) T={T} bs={B}/GPU fully-supervised train step, '
                                   f'{len(label_ts)} labelled frames/sequence, random-init weights',
                       'global_batch': world * B, 'seq_len': T, 'parallelism': f'dp{world}', 'launch': {'graph': 'one single-stream hipGraph per step', 'eager': f'eager, schedule={eng.schedule}, wgrad side stream {"on" if eng.wgrad_side else "off"}',
                                  'cells': f'per-cell hipGraphs, {eng.n_streams}-stream stage wavefront'}[launch],
                       'per_gpu_event_frames_per_s': round(fps / world, 2), 'final_loss': round(loss_val, 4),
                       'whole_step_hbm_frac_of_peak': round(ALGO_MB_PER_FRAME_FP32 * 1e6 * fps / world / (PEAK_HBM_GBS * 1e9), 5)
                       if headline else None},
            'roofline': roofline,
        }
        if not args.no_cpu_baseline and world == 1 and headline:
            out['cpu_baseline'] = cpu_baseline_bounded()
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
