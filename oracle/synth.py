"""Deterministic synthetic weights / inputs shared by make_golden, the tests and the smoke check.
TEST INFRASTRUCTURE (see oracle/__init__.py).

Weights are a pure function of (key, shape, seed) so that golden fixtures only need to store
*outputs*: the generator script loads ``synth_state_dict`` into the reference model, the tests
load the very same tensors into the oracle and into the HIP-backed modules.

LayerScale gammas are drawn O(0.5) instead of the reference's 1e-5 initial value so that the
attention / MLP branches actually contribute to the checked outputs.
"""
import zlib

import numpy as np
import torch


def _gen(key: str, seed: int) -> torch.Generator:
    g = torch.Generator(device='cpu')
    g.manual_seed((zlib.crc32(key.encode()) * 2654435761 + seed * 97 + 12345) % (2 ** 63 - 1))
    return g


def synth_tensor(key: str, shape, seed: int = 0) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    g = _gen(key, seed)
    if key.endswith('num_batches_tracked'):
        return torch.zeros(shape, dtype=torch.int64)
    if key.endswith('running_mean'):
        return 0.2 * torch.randn(shape, generator=g)
    if key.endswith('running_var'):
        return 0.5 + torch.rand(shape, generator=g)
    if key.endswith('.gamma'):                                   # LayerScale
        return 0.25 + 0.5 * torch.rand(shape, generator=g)
    last = key.rsplit('.', 2)
    is_norm = any(t in key for t in ('.norm', '.bn.', 'norm1', 'norm2'))
    if key.endswith('.weight') and len(shape) == 1:              # LN / BN scale
        return 1.0 + 0.2 * torch.randn(shape, generator=g)
    if key.endswith('.bias') and is_norm:
        return 0.1 * torch.randn(shape, generator=g)
    if key.endswith('.bias'):
        if 'cls_preds' in key or 'obj_preds' in key:
            # around the reference's prior-prob bias, raised so that detections exist
            return -2.0 + 0.5 * torch.randn(shape, generator=g)
        return 0.1 * torch.randn(shape, generator=g)
    if key.endswith('.weight'):
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        return torch.randn(shape, generator=g) * (1.2 / max(fan_in, 1) ** 0.5)
    if key.endswith('mask_token'):
        return 0.02 * torch.randn(shape, generator=g)
    del last
    return torch.randn(shape, generator=g)


def synth_state_dict(manifest, seed: int = 0):
    """manifest: {key: shape}"""
    return {k: synth_tensor(k, s, seed) for k, s in manifest.items()}


def synth_events(T, B, C, H, W, seed=0, density=0.08, as_uint8=True):
    """Stacked-histogram-like sparse small counts (SURVEY 8d synthetic input)."""
    g = torch.Generator(device='cpu')
    g.manual_seed(1000 + seed)
    mask = torch.rand((T, B, C, H, W), generator=g) < density
    vals = torch.randint(1, 10, (T, B, C, H, W), generator=g)
    ev = (mask * vals)
    return ev.to(torch.uint8) if as_uint8 else ev.to(torch.float32)


def synth_labels(n_frames, hw, num_classes, seed=0, max_boxes=6, min_boxes=1):
    """Per labelled frame 1..6 boxes, w~U(10,90), h~U(10,70), fully inside the frame.
    Returns list of float32 [n,8] = (t,x,y,w,h,class_id,class_confidence,objectness), corner xy."""
    H, W = hw
    rng = np.random.RandomState(2000 + seed)
    out = []
    for _ in range(n_frames):
        n = rng.randint(min_boxes, max_boxes + 1)
        w = rng.uniform(10, min(90, W - 2), size=n)
        h = rng.uniform(10, min(70, H - 2), size=n)
        x = rng.uniform(0, W - 1 - w)
        y = rng.uniform(0, H - 1 - h)
        cls = rng.randint(0, num_classes, size=n)
        lab = np.stack([np.full(n, 1.0), x, y, w, h, cls, np.ones(n), np.ones(n)], axis=1)
        out.append(torch.from_numpy(lab.astype(np.float32)))
    return out


def synth_augment_sample(seed: int, H: int, W: int, T: int = 4):
    """One synthetic loader sample for the augmentation fixtures (tests/golden/g14_augment.npz): T uint8 voxel frames
    [20,H,W] and per-frame label rows [n,8] (t,x,y,w,h,class,cls_conf,obj) or None on unlabelled frames."""
    import torch
    g = torch.Generator().manual_seed(4000 + seed)
    ev = [((torch.rand((20, H, W), generator=g) < 0.15) * torch.randint(1, 12, (20, H, W), generator=g)).to(torch.uint8)
          for _ in range(T)]
    labels = []
    for t in range(T):
        if t % 2 == 0 and not (seed == 3 and t == 0):
            labels.append(None)
            continue
        n = int(torch.randint(1, 4, (1,), generator=g))
        w = torch.rand(n, generator=g) * 0.4 * W + 6
        h = torch.rand(n, generator=g) * 0.4 * H + 6
        x = torch.rand(n, generator=g) * (W - 1 - w)
        y = torch.rand(n, generator=g) * (H - 1 - h)
        labels.append(torch.stack([torch.full((n,), 1000. * (t + 1)), x, y, w, h,
                                   torch.randint(0, 2, (n,), generator=g).float(), torch.ones(n), torch.ones(n)], 1))
    return ev, labels


AUGMENT_CASES = [(s, 60, 76) for s in range(8)] + [(100, 48, 64), (101, 48, 64)]
AUGMENT_CFG = dict(prob_hflip=0.5, prob_tflip=0, rotate=dict(prob=0, min_angle_deg=2, max_angle_deg=6),
                   zoom=dict(prob=0.8, zoom_in=dict(weight=8, factor=dict(min=1, max=1.5)),
                             zoom_out=dict(weight=2, factor=dict(min=1, max=1.2))))


EVAL_BBOX_DTYPE = np.dtype({'names': ['t', 'x', 'y', 'w', 'h', 'class_id', 'track_id', 'class_confidence'],
                            'formats': ['<i8', '<f4', '<f4', '<f4', '<f4', '<u4', '<u4', '<f4'],
                            'offsets': [0, 8, 12, 16, 20, 24, 28, 32], 'itemsize': 40})


def synth_eval_sequences(seed: int, hw=(240, 304), n_cls: int = 2, n_seq: int = 3, n_frames: int = 12, dt_per_gt=(0, 4),
                         jitter_us: int = 0, quantize_scores: bool = False):
    """Label / detection records for the evaluator fixtures: ``n_seq`` recordings, labelled frames every 100 ms from 0.3 s
    (so the first ones fall under the 0.5 s filter), boxes from tiny to large, detections = jittered copies of the labels
    (hits of varying IoU), duplicates and clutter; ``jitter_us`` moves detection timestamps around the label time (some beyond
    the +-50 ms tolerance); ``quantize_scores`` produces score ties.  Returns (labels, detections): lists of structured arrays
    sorted by time."""
    rng = np.random.RandomState(7000 + seed)
    H, W = hw
    labels, dets = [], []
    for _ in range(n_seq):
        g_rows, d_rows = [], []
        for f in range(n_frames):
            t = 300000 + 100000 * f
            for _ in range(rng.randint(0, 5)):
                w, h = rng.uniform(4, 0.6 * W), rng.uniform(4, 0.6 * H)
                if rng.rand() < 0.3:
                    w, h = rng.uniform(4, 40), rng.uniform(4, 30)
                x, y = rng.uniform(0, W - w), rng.uniform(0, H - h)
                c = rng.randint(0, n_cls)
                g_rows.append((t, x, y, w, h, c, 0, 1.0))
                for _ in range(rng.randint(dt_per_gt[0], dt_per_gt[1] + 1)):
                    s = rng.uniform(0.02, 0.25)
                    dw, dh = w * (1 + rng.uniform(-s, s)), h * (1 + rng.uniform(-s, s))
                    dx, dy = x + rng.uniform(-s, s) * w, y + rng.uniform(-s, s) * h
                    dc = c if rng.rand() < 0.9 else rng.randint(0, n_cls)
                    td = t + (int(rng.randint(-jitter_us, jitter_us + 1)) if jitter_us else 0)
                    d_rows.append((td, dx, dy, dw, dh, dc, 0, rng.uniform(0.05, 1.0)))
            for _ in range(rng.randint(0, 3)):                       # clutter, also on frames without labels
                w, h = rng.uniform(8, 80), rng.uniform(8, 60)
                td = t + (int(rng.randint(-jitter_us, jitter_us + 1)) if jitter_us else 0)
                d_rows.append((td, rng.uniform(0, W - w), rng.uniform(0, H - h), w, h, rng.randint(0, n_cls), 0,
                               rng.uniform(0.05, 0.6)))
        g = np.array(g_rows, dtype=EVAL_BBOX_DTYPE) if g_rows else np.zeros((0,), dtype=EVAL_BBOX_DTYPE)
        d = np.array(d_rows, dtype=EVAL_BBOX_DTYPE) if d_rows else np.zeros((0,), dtype=EVAL_BBOX_DTYPE)
        if quantize_scores and len(d):
            d['class_confidence'] = np.round(d['class_confidence'] * 8) / 8 + 0.0625
        labels.append(g[np.argsort(g['t'], kind='stable')])
        dets.append(d[np.argsort(d['t'], kind='stable')])
    return labels, dets


EVAL_CASES = [dict(seed=0), dict(seed=1, jitter_us=70000), dict(seed=2, hw=(360, 640), n_cls=3, jitter_us=30000),
              dict(seed=3, quantize_scores=True, dt_per_gt=(1, 6)), dict(seed=4, n_seq=1, n_frames=4, dt_per_gt=(0, 0))]


def synth_tta_views(case: int, hw=(240, 304)):
    """Scripted input of the TTA result fixtures (tests/golden/g16_tta_result.npz): one recording of 8 frames, labels on 5 of
    them, delivered in two chunks per view.  Returns (views, hw); a view = dict(hflip, tflip, ev_idx, gts, preds, last) with
    gts[k] = [n,8] label tensor or the float placeholder 1.0 (frame without labels) and preds[k] = [m,7] tensor
    (x1, y1, x2, y2, obj, cls_conf, cls_id) in the coordinates OF THAT VIEW.  case 0: plain + hflip; 1: all four views; 2: plain
    only (no merge)."""
    import torch
    g = torch.Generator().manual_seed(8100 + case)
    H, W = hw
    labelled = {1, 2, 4, 6, 7}
    gt_of = {}
    for f in sorted(labelled):
        n = int(torch.randint(1, 4, (1,), generator=g))
        w, h = 20 + 60 * torch.rand(n, generator=g), 15 + 50 * torch.rand(n, generator=g)
        x, y = torch.rand(n, generator=g) * (W - 1 - w), torch.rand(n, generator=g) * (H - 1 - h)
        gt_of[f] = torch.stack([torch.full((n,), 100000. * (f + 1)), x, y, w, h, torch.randint(0, 2, (n,), generator=g).float(),
                                torch.ones(n), torch.ones(n)], 1)

    def dets_for(f, hflip):
        gt = gt_of[f]
        reps = int(torch.randint(1, 4, (1,), generator=g))
        rows = gt.repeat(reps, 1)
        n = len(rows)
        x1 = rows[:, 1] + 4 * torch.randn(n, generator=g)
        y1 = rows[:, 2] + 4 * torch.randn(n, generator=g)
        wv, hv = rows[:, 3] * (1 + 0.1 * torch.randn(n, generator=g)), rows[:, 4] * (1 + 0.1 * torch.randn(n, generator=g))
        if hflip:
            x1 = W - 1 - x1 - wv
        extra = int(torch.randint(0, 3, (1,), generator=g))
        p = torch.stack([x1, y1, x1 + wv, y1 + hv, 0.2 + 0.8 * torch.rand(n, generator=g), 0.2 + 0.8 * torch.rand(n, generator=g), rows[:, 5]], 1)
        if extra:
            c = torch.rand(extra, 2, generator=g) * torch.tensor([W - 60., H - 50.])
            p = torch.cat([p, torch.cat([c, c + 40, torch.rand(extra, 2, generator=g), torch.randint(0, 2, (extra, 1), generator=g).float()], 1)])
        return p

    combos = {0: [(False, False), (True, False)], 1: [(False, False), (True, False), (False, True), (True, True)],
              2: [(False, False)]}[case]
    views = []
    for hflip, tflip in combos:
        # the reversed recording shows plain frame f at reversed index f + 1 (offset -1), in descending plain order
        order = list(range(8)) if not tflip else list(range(7, -1, -1))
        for ci, chunk in enumerate((order[:4], order[4:])):
            gts, preds, idx = [], [], []
            for f in chunk:
                idx.append(f if not tflip else f + 1)
                if f in labelled:
                    gts.append(gt_of[f])
                    preds.append(dets_for(f, hflip))
                else:
                    gts.append(1.0)
                    preds.append(1.0)
            views.append(dict(hflip=hflip, tflip=tflip, ev_idx=idx, gts=gts, preds=preds, last=ci == 1))
    return views, hw


# ---- a tiny GenX dataset tree on disk (loader goldens / tests) ------------------------------------------------------------------
BBOX_DTYPE = np.dtype({'names': ['t', 'x', 'y', 'w', 'h', 'class_id', 'class_confidence', 'objectness'],
                       'formats': ['<i8', '<f4', '<f4', '<f4', '<f4', '<u4', '<f4', '<f4'],
                       'offsets': [0, 8, 12, 16, 20, 24, 28, 32], 'itemsize': 40})


def synth_recording(split_dir: str, name: str, seed: int, n_frames: int, labelled: list, dst_name: str = 'gen1', frame_hw=(6, 8),
                    ds2: bool = False, with_h5_placeholder: bool = True, objectness: bool = True) -> str:
    """Write ``split_dir/name`` in the dataset layout of the reference (sequence_base.py:32-48): uint8 frames [N,20,h,w] as the
    raw ``.npy`` twin of the HDF5 file (plus an empty ``.h5`` placeholder so that path checks of the reference pass: its h5py
    stand-in reads the twin), ``objframe_idx_2_repr_idx.npy`` and ``labels_v2/labels.npz`` (BBOX_DTYPE).  ``labelled`` = frame
    indices that carry boxes.  Frame content encodes (recording, frame, channel) so that any mix-up shows."""
    import os
    rng = np.random.RandomState(seed)
    seq = os.path.join(split_dir, name)
    ev_dir = os.path.join(seq, 'event_representations_v2', 'stacked_histogram_dt=50_nbins=10')
    os.makedirs(ev_dir, exist_ok=True)
    os.makedirs(os.path.join(seq, 'labels_v2'), exist_ok=True)
    h, w = frame_hw
    frames = rng.randint(0, 4, size=(n_frames, 20, h, w)).astype(np.uint8)
    frames[:, :, 0, 0] = (np.arange(n_frames)[:, None] + 1) % 251           # frame index
    frames[:, :, 0, 1] = np.arange(20)[None, :] + 1                         # channel index
    frames[:, :, 0, 2] = seed % 251
    stem = 'event_representations' + ('_ds2_nearest' if ds2 else '')
    np.save(os.path.join(ev_dir, stem + '.npy'), frames)
    if with_h5_placeholder:
        open(os.path.join(ev_dir, stem + '.h5'), 'wb').close()
    np.save(os.path.join(ev_dir, 'objframe_idx_2_repr_idx.npy'), np.asarray(labelled, dtype=np.int64))
    H, W = (240, 304) if dst_name == 'gen1' else (720, 1280)
    rows, starts = [], []
    names = list(BBOX_DTYPE.names) if objectness else [n for n in BBOX_DTYPE.names if n != 'objectness']
    dt = BBOX_DTYPE if objectness else np.dtype([(n, BBOX_DTYPE.fields[n][0]) for n in names])
    for k, f in enumerate(labelled):
        starts.append(len(rows))
        for _ in range(rng.randint(1, 4)):
            bw, bh = rng.uniform(20, 90), rng.uniform(20, 70)
            x, y = rng.uniform(-10, W - bw + 10), rng.uniform(-10, H - bh + 10)     # some boxes stick out of the frame
            rec = dict(t=(f + 1) * 50000, x=x, y=y, w=bw, h=bh, class_id=rng.randint(0, 2 if dst_name == 'gen1' else 3),
                       class_confidence=1.0, objectness=1.0)
            rows.append(tuple(rec[n] for n in names))
    labels = np.array(rows, dtype=dt) if rows else np.zeros((0,), dtype=dt)
    np.savez(os.path.join(seq, 'labels_v2', 'labels.npz'), labels=labels, objframe_idx_2_label_idx=np.asarray(starts, dtype=np.int64))
    return seq


LOADER_RECORDINGS = [   # (name, seed, n_frames, labelled frames)
    ('rec_a', 11, 23, [4, 9, 10, 15, 22]),
    ('rec_b', 12, 12, [1, 2, 11]),
    ('rec_c', 13, 40, [6, 7, 30, 31, 39]),
    ('rec_d', 14, 9, [8]),
    ('rec_e', 15, 17, [3, 5, 7, 9, 11, 13, 16]),
]


def synth_dataset_tree(root: str, dst_name: str = 'gen1', ds2: bool = False, frame_hw=(6, 8)):
    """root/<dst_name>/{train,val,test}/rec_* from LOADER_RECORDINGS -> the dataset path."""
    import os
    base = os.path.join(root, dst_name)
    for split, off in (('train', 0), ('val', 100), ('test', 200)):
        for name, seed, n, lab in LOADER_RECORDINGS:
            synth_recording(os.path.join(base, split), name, seed + off, n, lab, dst_name=dst_name, ds2=ds2, frame_hw=frame_hw)
    return base


def loader_cases():
    """(kind, recording, constructor kwargs, time_flip, sample indices | None = all) -- used by tests/golden/make_golden.py (g17) and tests/test_loader_cpu.py."""
    L = 5
    cases = []
    for rec in ('rec_a', 'rec_b', 'rec_c', 'rec_d', 'rec_e'):
        for tf in (False, True):
            cases.append(('iter', rec, dict(sequence_length=L), tf, None))
            cases.append(('rnd', rec, dict(sequence_length=L), tf, None))
    cases.append(('iter', 'rec_c', dict(sequence_length=7, start_from_zero=True), False, None))
    cases.append(('iter', 'rec_c', dict(sequence_length=7, start_from_zero=True), True, None))
    cases.append(('iter', 'rec_a', dict(sequence_length=L, range_indices=(5, 16)), False, None))
    cases.append(('iter', 'rec_a', dict(sequence_length=L, range_indices=(5, 16)), True, None))
    cases.append(('iter', 'rec_e', dict(sequence_length=L, objframe_idx=[1, 4], data_ratio=0.3), False, None))     # withheld labels
    cases.append(('iter', 'rec_e', dict(sequence_length=L, objframe_idx=[], data_ratio=-1.0), True, None))         # all withheld
    cases.append(('rnd', 'rec_e', dict(sequence_length=L, data_ratio=0.5), False, None))
    cases.append(('rnd', 'rec_e', dict(sequence_length=L, data_ratio=0.5), True, None))
    cases.append(('rnd', 'rec_e', dict(sequence_length=L, objframe_idx=[2, 5], data_ratio=0.3), False, None))
    cases.append(('rnd', 'rec_c', dict(sequence_length=8, data_ratio=0.25), True, None))
    return cases
