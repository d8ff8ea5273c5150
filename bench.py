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

os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')    # kernel-argument blocks in device memory (see leod_amd/__init__.py); before HIP initialises
import numpy as np  # noqa: E402
import torch  # noqa: E402

PROBE_STEPS = 3                 # bracketed single-stream eager steps behind roofline / family_ms_per_step (median step), after one warm step
PEAK_HBM_GBS = 8000.0           # MI355X HBM3E peak (MI355X_MICROARCH.md)
PEAK_F32_MFMA_TFLOPS = 157.3    # dense fp32 MFMA peak: 256 FLOP/clk/CU x 256 CUs x 2.4 GHz (MI355X_MICROARCH.md)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md; AMD's 5 PF figure includes 2:1 sparsity)
# SURVEY 8d / DESIGN.md: algorithmic HBM traffic of one RVT-S training event-frame: 16-bit activations (68.61 MB backbone + 32/168 of the
# 29.01 MB of PAFPN + head per labelled frame) + 395/168 MB of optimiser traffic = 76.5 MB; with fp32 activations twice the activation part
CFG_PRECISION = {'16f': 16, 'bf16': 'bf16', 'f32': 32}      # training.precision per bench mode (16 = the reference's fp16 autocast -> mode 16f)
DTYPE_NAME = {'16f': 'fp16', 'bf16': 'bf16', 'f32': 'f32'}   # the JSON line's dtype: the operand type of the forward contractions
ALGO_MB_PER_FRAME = {'bf16': 68.61 + 32.0 / 168.0 * 29.01 + 395.0 / 168.0,
                     'f32': 2 * (68.61 + 32.0 / 168.0 * 29.01) + 395.0 / 168.0}


def _native_comm_active():
    from leod_amd.comm import NativeComm
    return NativeComm.active


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
    nmax = max([len(l) for l in labs] or [1])
    tg = np.zeros((len(labs), nmax, 7), np.float32)
    for i, l in enumerate(labs):
        tg[i, :len(l)] = l
    return ev.to(device), torch.from_numpy(tg).to(device), label_tb, labs


from leod_amd.utils.host import usable_cores  # noqa: E402  (min(affinity mask, cgroup CPU quota))


def cpu_baseline(sample_B=8, T=21, threads=None, timed=3):
    """The oracle (CPU restatement of the reference, torch fp32 autograd) on a bounded sample of the same workload:
    1 warm-up step + ``timed`` timed steps, best-of, on all usable cores; plus a best-of-2 figure at 8 threads (the core
    count of the build container, where tools/time_reference.py times the reference itself against this oracle)."""
    from oracle import train_step as ot
    from oracle.synth import synth_state_dict
    import json as _json
    threads = threads or usable_cores()
    man = _json.load(open(os.path.join(ROOT, 'tests', 'golden', 'g11_manifest.json')))['small_gen1']
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
    first = torch.ones(sample_B, dtype=torch.bool)

    def best_of(n_threads, warm, n):
        torch.set_num_threads(n_threads)
        tr = ot.OracleTrainer(synth_state_dict(man, 0), ot.model_cfg(48, 24, 0.33, (8, 10)))
        best = float('inf')
        for i in range(warm + n):
            t0 = time.time()
            tr.step(ev, labels, first)
            if i >= warm:
                best = min(best, time.time() - t0)
        return best

    dt = best_of(threads, 1, timed)
    out = dict(value=round(sample_B * T / dt, 3), unit='event-frames/s', cores=threads, kind='port',
               sample=f'oracle (PyTorch-CPU fp32 restatement of the reference) RVT-S Gen1 T={T} bs={sample_B}: 1 warm-up + {timed} '
                      f'timed full training steps of {sample_B * T} event-frames, best step {dt:.2f} s')
    if threads != 8:
        dt8 = best_of(min(8, threads), 1, 2)
        out['value_8_threads'] = round(sample_B * T / dt8, 3)
    return out


def cpu_baseline_bounded(timeout_s=300):
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


def pseudo_cpu_baseline(L=21, B=2, timed=2):
    """The oracle's pseudo-label inference (backbone over L, head, NMS; hflip copy included) on a bounded sample: B source streams."""
    from oracle import train_step as ot
    from oracle.synth import synth_state_dict
    import json as _json
    threads = usable_cores()
    torch.set_num_threads(threads)
    man = _json.load(open(os.path.join(ROOT, 'tests', 'golden', 'g11_manifest.json')))['small_gen1']
    sd = synth_state_dict(man, 0)
    for k in sd:
        if ('obj_preds' in k or 'cls_preds' in k) and k.endswith('bias'):
            sd[k] = sd[k] + 4.0
    ev, _, _, _ = make_batch(L, B, (240, 304), 2, 7, 'cpu', ())
    cfg = ot.model_cfg(48, 24, 0.33, (8, 10))
    best = float('inf')
    with torch.no_grad():
        for i in range(1 + timed):
            t0 = time.time()
            ot.infer_sequence(sd, cfg, ev, None, conf_thre=0.01, hflip=True)
            if i:
                best = min(best, time.time() - t0)
    return dict(value=round(L * B / best, 3), unit='source event-frames/s', cores=threads, kind='port',
                sample=f'oracle pseudo-label inference (hflip TTA, NMS) RVT-S Gen1 L={L}, {B} source streams: 1 warm-up + {timed} timed passes of '
                       f'{L * B} source frames, best {best:.2f} s')


def pseudo_main(args):
    """BASELINE.json configs[4], single-GPU leg: the pseudo-label inference loop through the product class -- ``PseudoLabeler.predict_step``
    (modules/pseudo_labeler.py:622-770) on loader-shaped streaming batches of B source recordings x L frames with horizontal-flip TTA
    (2B frame streams through the backbone), head on every frame, batched NMS, pred2label and the per-recording bookkeeping.  Weights
    are random-init plus the +4.0 objectness / class bias bump of SURVEY 8d so that NMS sees O(100) candidates per frame."""
    from leod_amd.parallel import init_distributed
    rank, local, world = init_distributed()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    import torch.distributed as dist
    from leod_amd.config import full_config, dynamically_modify_train_config
    from leod_amd.data.genx_utils.labels import SparselyBatchedObjectLabels
    from leod_amd.data.utils.types import DataType
    from leod_amd.modules.pseudo_labeler import PseudoLabeler
    from leod_amd.modules.utils.detection import DATA_KEY, WORKER_ID_KEY
    from leod_amd import ops
    L, B = args.seq_len, args.batch
    over = dict(dataset=dict(sequence_length=L), tta=dict(enable=True, hflip=True, tflip=False))
    cfg = dynamically_modify_train_config(full_config('gen1', args.size, model='pseudo_labeler', overrides=over))
    cfg.training.precision = CFG_PRECISION[args.dtype]
    cfg.model.postprocess.confidence_threshold = 0.01
    torch.manual_seed(0)
    mod = PseudoLabeler(cfg).to(dev).eval()
    mod.setup('predict')
    with torch.no_grad():
        for k in range(3):
            mod.mdl.yolox_head.obj_preds[k].bias += 4.0
            mod.mdl.yolox_head.cls_preds[k].bias += 4.0
    hw = (240, 304)
    ev, _, _, _ = make_batch(L, B, hw, 2, rank, dev, ())
    none_seq = lambda: [SparselyBatchedObjectLabels([None] * B) for _ in range(L)]   # noqa: E731
    step_no = [0]

    def batch():
        s = step_no[0]
        step_no[0] += 1
        first = torch.full((B,), s == 0)
        return {WORKER_ID_KEY: 0, DATA_KEY: {
            DataType.EV_REPR: [ev[t] for t in range(L)], DataType.OBJLABELS_SEQ: none_seq(), DataType.SKIPPED_OBJLABELS_SEQ: none_seq(),
            DataType.IS_FIRST_SAMPLE: first, DataType.IS_LAST_SAMPLE: torch.zeros(B, dtype=torch.bool),
            DataType.IS_REVERSED: torch.zeros(B, dtype=torch.bool), DataType.EV_IDX: [torch.full((B,), L * s + t, dtype=torch.long) for t in range(L)],
            DataType.IS_PADDED_MASK: [torch.zeros(B, dtype=torch.bool) for _ in range(L)], DataType.PATH: [f'train/rec{rank}_{b}' for b in range(B)]}}

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    mod.pipelined = True                       # as leod_amd.predict.run_pseudo_labeling drives it: host half of chunk i - 1 under chunk i
    for _ in range(args.warmup):
        mod.predict_step(batch(), 0)
    mod.flush_predictions()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mod.predict_step(batch(), 0)
    mod.flush_predictions()
    barrier()
    dt = time.perf_counter() - t0
    t_max = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist.is_initialized():
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    dt = float(t_max)
    n_lab = sum(len(l) for e in mod.ev_path_2_ev_data.values() for l in e.frame_idx_2_labels.values() if l is not None)
    roofline = family_ms = None
    if not args.no_roofline and rank == 0:
        probe = ops.KernelProbe()
        for _ in range(PROBE_STEPS):
            mod.predict_step(batch(), 0)
            probe.mark_step()
        mod.flush_predictions()
        roofline = probe.finish(PEAK_HBM_GBS, PEAK_BF16_MFMA_TFLOPS if args.dtype != 'f32' else PEAK_F32_MFMA_TFLOPS, target='linear_gemm')
        family_ms = probe.family_ms()
        if args.dump_calls:                               # every C launch of the two probe chunks, in order
            with open(args.dump_calls, 'w') as f:
                for n, ints, us in probe.call_table():
                    f.write(f'{us:9.1f}  {n:<34s} {ints}\n')
    if rank == 0:
        fps = world * B * L * args.steps / dt
        # inference-forward bytes per PROCESSED frame (backbone 15.22 MB + PAFPN / head forward 9.67 MB at 16-bit activations;
        # twice that in fp32 mode); a source frame is processed twice (hflip copy)
        mb_frame = (15.22 + 9.67) * (1 if args.dtype != 'f32' else 2)
        out = {'metric': 'source event-frames/sec (pseudo-label inference: RVT-S backbone + head + batched NMS + pred2label, hflip TTA), whole job',
               'value': round(fps, 2), 'unit': 'source event-frames/s (whole job)', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': round(1e3 * dt / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': DTYPE_NAME[args.dtype], 'precision_mode': args.dtype, 'data': 'synthetic',
               'config': {'workload': f'LEOD pseudo-label pass (BASELINE configs[4], one shard): RVT-{args.size} gen1 {hw[0]}x{hw[1]} L={L}, {B} source streams '
                                      f'/GPU + hflip TTA = {2 * B} frame streams, conf 0.01 / NMS 0.45, random-init weights + 4.0 obj / cls bias bump',
                          'driver': 'PseudoLabeler.predict_step, pipelined as leod_amd.predict.run_pseudo_labeling drives it (host bookkeeping of chunk i - 1 under the device work of chunk i), eager launches',
                          'processed_frames_per_s': round(2 * fps, 2), 'pseudo_labels_stored': n_lab,
                          'algorithmic_MB_per_processed_frame': mb_frame,
                          'whole_pass_hbm_frac_of_peak': round(mb_frame * 1e6 * 2 * fps / world / (PEAK_HBM_GBS * 1e9), 5),
                          'family_ms_per_step': family_ms},
               'roofline': roofline}
        if not args.no_cpu_baseline and world == 1:
            import subprocess
            code = ('import json, sys; sys.path.insert(0, %r); import bench; print("CPUBASE " + json.dumps(bench.pseudo_cpu_baseline()))' % ROOT)
            try:
                so = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300,
                                    env=dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')).stdout
                out['cpu_baseline'] = next(json.loads(l[8:]) for l in so.splitlines() if l.startswith('CPUBASE '))
            except Exception as e:                           # noqa: BLE001
                out['cpu_baseline'] = dict(value=None, unit='source event-frames/s', cores=usable_cores(), kind='port', sample=f'CPU leg failed: {e}')
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        from leod_amd.comm import NativeComm
        NativeComm.shutdown()
        dist.destroy_process_group()



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--seq-len', type=int, default=21)
    ap.add_argument('--size', default='small')
    ap.add_argument('--dataset', choices=('gen1', 'gen4'), default='gen1', help='gen4: 3 classes, 360x640 frames (downsampled by 2)')
    ap.add_argument('--full-res', action='store_true', help='gen4 at 720x1280 -> 768x1280, 240-token partitions '
                    '(BASELINE configs[3]: --dataset gen4 --full-res --size base --seq-len 11 --batch 2)')
    ap.add_argument('--dtype', choices=('16f', 'bf16', 'f32'), default='16f',
                    help='precision mode of the contractions (leod_set_precision): 16f = the reference\'s precision=16 (fp16 MFMA operands and fp16 '
                         'activation rows in the forward pass, bf16 operands for the gradients, fp32 accumulation / statistics / state / optimiser), '
                         'bf16 = bf16 operands in both directions, f32 = fp32 end to end (the bit-tight parity mode)')
    ap.add_argument('--no-second-dtype', action='store_true', help='skip the secondary line measured in the other precision (N=1 only)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-plan', action='store_true', help='eager Python launches for every step (LEOD_PLAN=0): no launch plans')
    ap.add_argument('--single-stream', action='store_true', help='profiling aid: every kernel on the launch stream (no weight-gradient side stream, '
                    'no per-level head streams); use with --no-plan')
    ap.add_argument('--vary-labels', default='', metavar='LO:HI', help='draw the labelled frames of every step at random: B\' ~ U{LO..HI} distinct '
                    '(t, b) positions per step (the reference\'s loaders deliver a data-dependent number of labelled frames per step, '
                    'modules/detection.py:209-224); default: the fixed frames t in {4, 9, 14, 19} of every sequence (B\' = 32).  NOT the BASELINE '
                    'headline workload: reported with config.vary_labels set and the plan cache\'s hit rate')
    ap.add_argument('--dump-calls', default='', help='file for the per-launch table (C entry point, shape arguments, us) of the probe steps')
    ap.add_argument('--pseudo', action='store_true', help='the pseudo-label inference pass of BASELINE configs[4] (single-GPU leg) instead of '
                    'the training step: PseudoLabeler.predict_step, --batch source streams + hflip TTA, --seq-len frames per chunk')
    args = ap.parse_args()
    if args.pseudo:
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs an MI355X (no CPU fallback for the HIP path)')
        return pseudo_main(args)

    from leod_amd.parallel import init_distributed
    rank, local, world = init_distributed()
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback for the HIP path)')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    import torch.distributed as dist
    from leod_amd.config import full_config, dynamically_modify_train_config
    from leod_amd.data.genx_utils.labels import ObjectLabels, SparselyBatchedObjectLabels
    from leod_amd.data.utils.types import DataType
    from leod_amd.modules.utils.detection import DATA_KEY, WORKER_ID_KEY
    from leod_amd.modules.utils.fetch import fetch_model_module
    from leod_amd.optim import fit_step
    from leod_amd import ops

    over = dict(dataset=dict(downsample_by_factor_2=not args.full_res)) if args.dataset == 'gen4' else {}
    over.setdefault('dataset', {})['sequence_length'] = args.seq_len
    cfg = dynamically_modify_train_config(full_config(args.dataset, args.size, overrides=over))
    hw = (240, 304) if args.dataset == 'gen1' else ((720, 1280) if args.full_res else (360, 640))
    in_hw = tuple(cfg.model.backbone.in_res_hw)
    headline = args.dataset == 'gen1' and args.size == 'small'       # the configuration BASELINE.json's metric is quoted on
    T, B = args.seq_len, args.batch
    label_ts = tuple(t for t in (4, 9, 14, 19) if t < T) or (T - 1,)

    def measure(dtype, steps, warmup, want_roofline):
        """One full measurement in precision mode ``dtype``: fresh module (same seed), warm-up, K timed steps between barriers,
        optional roofline probe.  -> dict(dt, loss, roofline, module)."""
        # The product path, through the reference's own surface (train.py:131-133,228-250): fetch_model_module(config) ->
        # Module.setup('fit') -> configure_optimizers() -> per batch what Lightning's automatic optimisation does
        # (optimizer.step(closure: zero_grad, training_step, backward), scheduler.step()).
        cfg.training.precision = CFG_PRECISION[dtype]
        os.environ.pop('LEOD_PRECISION', None)                # the config decides (Module.setup -> leod_set_precision)
        torch.manual_seed(0)                                  # identical random-init weights on every rank
        module = fetch_model_module(cfg).to(dev)
        module.setup('fit')
        module.train()
        oc = module.configure_optimizers()                    # FlatAdamW (+ flat all-reduce / SyncBN when world > 1) and OneCycleLR
        opt, sched = (oc['optimizer'], oc['lr_scheduler']['scheduler']) if isinstance(oc, dict) else (oc, None)
        ev, _, label_tb, labs = make_batch(T, B, hw, cfg.model.head.num_classes, rank, dev, label_ts)
        g = torch.Generator(device='cpu').manual_seed(77 + rank)
        # host-side box labels as the loader delivers them: [n, 8] = (t, x, y, w, h, class_id, class_confidence, objectness)
        lab8 = [np.concatenate([np.ones((len(l), 1), np.float32), l[:, 1:2] - l[:, 3:4] / 2, l[:, 2:3] - l[:, 4:5] / 2, l[:, 3:5],
                                l[:, 0:1], l[:, 6:7], l[:, 5:6]], 1) for l in labs]

        def first_mask(step):
            m = torch.ones(B, dtype=torch.bool)
            if step > 0:                                      # stream half carries state, random half always restarts
                m[:B // 2] = torch.rand(B // 2, generator=g) < 0.05
            return m.to(dev)

        vary = tuple(int(v) for v in args.vary_labels.split(':')) if args.vary_labels else None
        vrng = np.random.RandomState(4242 + rank)
        if vary:                                              # a pool of label sets to draw from: one per possible labelled frame of a step
            prng = np.random.RandomState(2000 + rank)
            while len(lab8) < vary[1]:
                lab8.append(lab8[prng.randint(0, len(lab8))].copy())

        def loader_batch(mask):
            """The dictionary the reference's loaders emit (modules/data/genx.py:120-144): a list of L frame tensors [B,20,H,W]
            (uint8, device-resident: consecutive views of one buffer), L SparselyBatchedObjectLabels built from host arrays on
            every step, the is_first_sample flags, the worker id that keys the LSTM state."""
            tb = label_tb
            if vary:                                          # this step's labelled frames: B' ~ U{lo..hi} distinct (t, b) positions
                n = int(vrng.randint(vary[0], vary[1] + 1))
                pos = sorted(vrng.choice(T * B, size=n, replace=False).tolist())
                tb = [[p % B for p in pos if p // B == t] for t in range(T)]
            it = iter(lab8)
            seq = []
            for t in range(T):
                row = [None] * B
                for b in tb[t]:
                    row[b] = ObjectLabels(torch.from_numpy(next(it).copy()), hw)
                seq.append(SparselyBatchedObjectLabels(row))
            return {WORKER_ID_KEY: 0, DATA_KEY: {DataType.EV_REPR: [ev[t] for t in range(T)], DataType.OBJLABELS_SEQ: seq,
                                                 DataType.IS_FIRST_SAMPLE: mask}}

        def barrier():
            if dist.is_initialized():
                dist.barrier()
            torch.cuda.synchronize()

        step_no = [0]

        def run(mask):
            out = fit_step(module, opt, sched, loader_batch(mask), step_no[0])
            step_no[0] += 1
            return out

        if args.no_plan:
            module.plan_mode = False
        if args.single_stream:
            module.wgrad_side = False
        for s in range(args.warmup):
            run(first_mask(s))
        masks = [first_mask(1 + args.warmup + s) for s in range(args.steps)]
        pl0 = (module._plans.steps, module._plans.replays, module._plans.eager_steps, module._plans.captures, module._plans.head_captures)
        barrier()
        t0 = time.perf_counter()
        for s in range(args.steps):
            out = run(masks[s])
        barrier()
        dt = time.perf_counter() - t0
        pl_t = (module._plans.steps, module._plans.replays, module._plans.eager_steps, module._plans.captures, module._plans.head_captures)
        t_max = torch.tensor([dt], dtype=torch.float64, device=dev)
        if dist.is_initialized():
            dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dt = float(t_max)
        # host side of one step: enqueue time of single steps launched into an IDLE device (inside the timed loop the host runs ahead of
        # the GPU and its launch calls block on the full hardware queue, so the loop's own host time says nothing)
        host_ms = []
        for s in range(5):
            m = first_mask(2 + s)
            b = loader_batch(m)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            fit_step(module, opt, sched, b, step_no[0])
            host_ms.append(1e3 * (time.perf_counter() - t1))
            step_no[0] += 1
        torch.cuda.synchronize()
        host_ms = sorted(host_ms)[len(host_ms) // 2]
        plan_info = module._plans.info() if module.plan_mode else None
        if plan_info is not None:
            # the plan cache over the TIMED steps alone (warm-up and the probe steps excluded)
            pl = module._plans
            plan_info['timed_region'] = {'planned_steps': pl_t[0] - pl0[0], 'replays': pl_t[1] - pl0[1], 'eager_steps': pl_t[2] - pl0[2],
                                         'backbone_captures': pl_t[3] - pl0[3], 'head_captures': pl_t[4] - pl0[4],
                                         'plan_hit_rate': round((pl_t[1] - pl0[1]) / max((pl_t[0] - pl0[0]) + (pl_t[2] - pl0[2]), 1), 4)}
        # roofline of the dominant kernel: two extra steps with HIP events around each of its launches
        roofline = roofline_gemm = family_ms = probe = family_all = probe_note = None
        if not args.no_roofline:
            # EVERY rank runs the probe steps (a step contains collectives: a rank-0-only step would dead-lock N > 1);
            # only rank 0 brackets the kernels with events and reports
            # isolated launches: no co-running kernels inside the event bracket (wgrad on the launch stream), eager launches so that
            # every C entry point is bracketed
            side, module.wgrad_side = module.wgrad_side, False
            planned, module.plan_mode = module.plan_mode, False
            run(first_mask(1))                            # the eager single-stream path once UN-bracketed (allocator pools, code objects
            torch.cuda.synchronize()                      # and autograd buffers of this path warm), then PROBE_STEPS bracketed steps
            probe = ops.KernelProbe() if rank == 0 else None
            for s in range(PROBE_STEPS):
                run(first_mask(1))
                if probe is not None:
                    probe.mark_step()
            module.wgrad_side, module.plan_mode = side, planned
            if probe is not None:
                peak_t = PEAK_BF16_MFMA_TFLOPS if dtype != 'f32' else PEAK_F32_MFMA_TFLOPS      # fp16 and bf16 MFMA share one dense peak
                roofline = probe.finish(PEAK_HBM_GBS, peak_t, target='linear_wgrad')
                roofline_gemm = probe.finish(PEAK_HBM_GBS, peak_t, target='linear_gemm')
                # every C entry point bracketed with events during the same single-stream steps (median step per family): the line audits itself
                family_ms = probe.family_ms(top=8)
                family_all = probe.family_ms(top=1000)
                # self-check: the bracketed single-stream kernel time of a step must be of the order of the measured step (two lanes overlap,
                # brackets add gaps: 0.5x .. 2x); otherwise a bracket caught a host stall and the table is not evidence
                kms = sum(family_all.values())
                step_ms = 1e3 * dt / args.steps
                if not (0.5 * step_ms <= kms <= 2.0 * step_ms):
                    probe_note = f'bracketed kernel time {kms:.2f} ms per step is outside 0.5x..2x of ms_per_step {step_ms:.2f}: family table and secondary roofline withheld'
                    family_ms = family_all = roofline_gemm = None
                if args.dump_calls:                           # every C launch of the first probe step, in order (tools / profiles)
                    with open(args.dump_calls, 'w') as f:
                        for n, ints, us in probe.call_table():
                            f.write(f'{us:9.1f}  {n:<34s} {ints}\n')
            # HBM bytes per launch of the same kernel from the PMC passes kept under profiles/ (FETCH_SIZE x2 + WRITE_SIZE,
            # separate rocprofv3 --pmc runs; bench.py itself cannot sample hardware counters)
            tpath = os.path.join(ROOT, 'profiles', 'roofline_traffic.json')
            if rank == 0 and roofline is not None and headline and os.path.exists(tpath):      # measured on the headline workload only
                tj = json.load(open(tpath)).get(dtype, {})
                roofline['traffic'] = tj.get('hbm_bytes_per_launch')
                roofline['traffic_source'] = tj.get('source')
        barrier()
        loss_val = float(out['loss'].detach())

        how = ('launch plans (step captured once, replayed as plain stream launches from C: modules/step_plan.py)' if plan_info else 'eager')
        launch = (f'{how}, schedule={"batched" if module.time_batched else "timestep"}, wgrad side stream {"on" if module.wgrad_side else "off"}, '
                  f'precision mode {ops.get_precision()}')
        return dict(dt=dt, loss=loss_val, roofline=roofline, roofline_gemm=roofline_gemm, family_ms=family_ms, launch=launch, family_ms_all=family_all,
                    host_ms=host_ms, plan_info=plan_info, probe_note=probe_note,
                    calls=(len(probe.calls) // probe.steps if probe is not None else None))

    main_run = measure(args.dtype, args.steps, args.warmup, not args.no_roofline)
    dt, loss_val, roofline, launch = main_run['dt'], main_run['loss'], main_run['roofline'], main_run['launch']
    other = None
    if world == 1 and headline and not args.no_second_dtype:
        torch.cuda.empty_cache()
        od = 'f32' if args.dtype != 'f32' else '16f'
        r2 = measure(od, args.steps, args.warmup, not args.no_roofline)
        other = {'dtype': od, 'value': round(B * T * args.steps / r2['dt'], 2), 'unit': 'event-frames/s (whole job)',
                 'ms_per_step': round(1000 * r2['dt'] / args.steps, 3), 'final_loss': round(r2['loss'], 4), 'roofline': r2['roofline'],
                 'roofline_linear_gemm': r2['roofline_gemm']}
        del r2
    if rank == 0:
        frames = world * B * T * args.steps
        fps = frames / dt
        out = {
            'metric': 'event-frames/sec/GPU (RVT-S train, Gen1 T=21) at 1/2/4/8 GPUs; mAP@0.5 parity',
            'value': round(fps, 2), 'unit': 'event-frames/s (whole job)', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(1000 * dt / args.steps, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPE_NAME[args.dtype], 'precision_mode': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'RVT-{args.size} {args.dataset} {hw[0]}x{hw[1]} (pad {in_hw[0]}x{in_hw[1]}) T={T} bs={B}/GPU fully-supervised train step, ' +
                                   (f'{len(label_ts)} labelled frames/sequence' if not args.vary_labels else f'labelled frames drawn per step, B\' ~ U{{{args.vary_labels}}}') + ', random-init weights',
                       'global_batch': world * B, 'seq_len': T, 'parallelism': f'dp{world}',
                       'vary_labels': args.vary_labels or None,
                       'driver': 'fetch_model_module(cfg) -> Module.training_step + FlatAdamW.step + OneCycleLR.step (leod_amd.optim.fit_step)',
                       'launch': launch,
                       'precision': {'16f': 'mode 16f: fp16 MFMA operands and fp16 activation rows in the forward contractions (GEMM / conv / attention / ConvLSTM / '
                                            'stem: the reference\'s fp16 autocast, train.py:236-243), bf16 MFMA operands for the gradient contractions (no loss '
                                            'scaler), fp32 accumulation, fp32 LayerNorm / BatchNorm statistics, softmax, residual stream, LSTM state, SimOTA cost, '
                                            'losses, master weights and AdamW',
                                     'bf16': 'mode bf16: bf16 MFMA operands in both directions, fp32 accumulation / statistics / state / optimiser',
                                     'f32': 'fp32 end to end'}[args.dtype],
                       'collective_backend': (('rccl (library communicator, csrc/k_comm.hip)' if _native_comm_active() else dist.get_backend()) if dist.is_initialized() else None),
                       'collective_world_size': dist.get_world_size() if dist.is_initialized() else 1,
                       'per_gpu_event_frames_per_s': round(fps / world, 2), 'final_loss': round(loss_val, 4),
                       # algorithmic bytes of the MEASURED precision mode: 76.5 MB per event-frame with 16-bit activations
                       'whole_step_hbm_frac_of_peak': round(ALGO_MB_PER_FRAME['f32' if args.dtype == 'f32' else 'bf16'] * 1e6 * fps / world / (PEAK_HBM_GBS * 1e9), 5)
                       if headline else None,
                       'algorithmic_MB_per_event_frame': round(ALGO_MB_PER_FRAME['f32' if args.dtype == 'f32' else 'bf16'], 2) if headline else None,
                       # GPU time per C entry point (= kernel family) and step, HIP events around every launch of two single-stream steps
                       'family_ms_per_step': main_run['family_ms'],
                       # host side: wall time of the Python call chain of ONE step launched into an idle device (median of 5), the kernels
                       # of the step's launch plans (+ ~12 eager launches: zero-grad, copy-ins, optimiser) or, eager, the C calls of a step
                       'host_enqueue_ms_per_step': round(main_run['host_ms'], 3),
                       'kernel_ms_per_step': round(sum((main_run['family_ms_all'] or {}).values()), 3) if main_run.get('family_ms_all') else None,
                       'launches_per_step': (main_run['plan_info']['forward']['kernels'] + main_run['plan_info']['backward']['kernels'] + 12)
                       if main_run['plan_info'] else None,
                       'c_calls_per_eager_step': main_run['calls'], 'probe_steps': PROBE_STEPS, 'probe_note': main_run['probe_note'],
                       'launch_plans': main_run['plan_info']},
            'roofline': roofline,
            'roofline_linear_gemm': main_run['roofline_gemm'],
        }
        if other is not None:
            out['other_precision'] = other          # the same workload and step in the other precision mode, measured in this run
        if not args.no_cpu_baseline and world == 1 and headline:
            out['cpu_baseline'] = cpu_baseline_bounded()
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        from leod_amd.comm import NativeComm
        NativeComm.shutdown()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
