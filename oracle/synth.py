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
