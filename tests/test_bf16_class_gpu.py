"""Precision mode 'bf16' (the benchmarked mode) against the REFERENCE'S OWN 16-bit classes.

``tests/golden/g18_autocast.npz`` (``make_golden.py g18``) records the reference run in fp32 and under ``torch.autocast`` on the CPU of the
build container, in BOTH 16-bit dtypes: bfloat16 ('ac' / 'acf') and float16 with GradScaler-style loss scaling ('h16' / 'h16f' -- the dtype
the reference itself trains with: Lightning precision=16 = fp16 autocast on CUDA, train.py:236-243) -- each once with CPU autocast's own op
placement ('ac', 'h16') and once with CUDA autocast's fp32-op list emulated ('acf', 'h16f') -- and, per tensor, how far each 16-bit run of
the reference lands from its fp32 run.  This build's 16-bit mode computes with bf16 MFMA operands (no loss scaler), so

  * it is BOUNDED by the reference's bf16 class:   dev_hip(x) = || x(HIP bf16) - x(fp32) || / || x(fp32) ||  <=  SLACK * max(dev_ac, dev_acf)
    (+ a stated floor), and
  * it is STATED against the reference's real (fp16) class: every check prints dev_hip / max(dev_h16, dev_h16f) -- the bf16 significand
    is 8 bits against fp16's 11, so multiples of ~8 are the format, not the kernels -- and asserts the measured multiples do not grow
    (``FP16_CLASS_MULT``).

The fp32 side is this build's fp32 mode, itself pinned (a) to the oracle by tests/test_engine_gpu.py and (b) here to the reference's
recorded fp32 losses / gradient norms of the same workload.  Also here: the loss trajectory bf16 vs f32 on identical batches, and bf16-mode
cases on RVT-B / Gen4 384x640 / 1 Mpx 768x1280 (BASELINE configs[3]) and on the pseudo-label pass.
``pytest -m gpu``."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import backbone as ob  # noqa: E402
from oracle import postproc as op  # noqa: E402
from oracle import train_step as ot  # noqa: E402
from oracle.synth import synth_state_dict, synth_events  # noqa: E402

import test_engine_gpu as te  # noqa: E402
import test_model_gpu as tm  # noqa: E402

DEV = 'cuda'
SLACK = 1.5
KEYS = te.KEYS


@pytest.fixture(scope='module')
def gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return True


@pytest.fixture(scope='module')
def g18(golden_dir):
    return np.load(os.path.join(golden_dir, 'g18_autocast.npz'))


class precision:
    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        from leod_amd import ops
        self.prev = ops.set_precision(self.mode)

    def __exit__(self, *exc):
        from leod_amd import ops
        ops.set_precision(self.prev)
        return False


def rel(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    n = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / n) if n > 0 else float(np.linalg.norm(a - b))


def cos(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    n = np.linalg.norm(a) * np.linalg.norm(b)
    return float(a @ b / n) if n > 0 else 1.0


def cls(g, prefix, key):
    """the reference's bf16 class for one quantity: the looser of its two bf16-autocast runs"""
    return np.maximum(np.asarray(g[f'{prefix}_ac_{key}']), np.asarray(g[f'{prefix}_acf_{key}']))


def cls16(g, prefix, key):
    """the reference's fp16 class (the dtype it trains with) for one quantity: the looser of its two fp16-autocast runs"""
    return np.maximum(np.asarray(g[f'{prefix}_h16_{key}']), np.asarray(g[f'{prefix}_h16f_{key}']))


# HIP-bf16 deviation as a multiple of the reference's fp16 class: measured ceilings (round 4, see the printed values) with ~30 % head room.
# bf16 keeps 8 significand bits against fp16's 11: a factor of 8 per rounding is the number format itself.
# Mode '16f' (round 5: fp16 operands and fp16 activation rows in the forward pass, the reference's own autocast dtype; bf16 gradients): the
# forward quantities must sit AT the fp16 class (<= 2 x), the gradient direction within 2 x of it.
FP16_CLASS_MULT = {'bf16': {'feat': 10.5, 'state_c': 10.5, 'one_minus_cos': 5.0},      # measured: 5.4-8.0, 5.6-7.9, 3.7-3.8
                   '16f': {'feat': 1.5, 'state_c': 1.5, 'one_minus_cos': 1.5}}      # measured (r05_a_16f_class_first_run.log): 0.7-0.9, 0.7-0.9, 0.04-0.7
MODES16 = ['bf16', '16f']
MEASURED = {}


def _check_against_class(tag, g, prefix, names, grads16, grads32, losses16, losses32, h16, h32, c16, c32, frac_ok=0.97, mode='bf16'):
    """Shared assertions of the micro and the full-size step; everything measured is printed (run with -s to see it)."""
    MULT = FP16_CLASS_MULT[mode]
    tag = f'{tag}/{mode}'
    # ---- losses: each component within SLACK x the reference's own 16-bit deviation (never tighter than 1e-2 of the total)
    ref32 = np.asarray(g[f'{prefix}_fp32_losses'])
    band = SLACK * np.maximum(np.abs(np.asarray(g[f'{prefix}_ac_losses']) - ref32), np.abs(np.asarray(g[f'{prefix}_acf_losses']) - ref32))
    band = np.maximum(band, 1e-2 * abs(ref32[0]) * (np.abs(ref32) > 0))
    # Every component (and every gradient) is a SUM over anchors divided by the SimOTA match count num_fg (reference yolo_head.py:383-396),
    # and the count is discrete: two runs that agree to 1e-4 absolute at the stem output (the round-3 stem kernel against the one it
    # replaced) moved it by one match of ~58, i.e. every normalised component and gradient by 1.7 % -- more than the band.  A relative
    # change `nfg_shift` of the count (bounded on its own below) is therefore allowed on top of the class band, as that rescaling.
    l16, l32 = np.asarray(losses16, np.float64), np.asarray(losses32, np.float64)
    nfg_shift = abs(l16[5] - l32[5]) / l32[5] if l32[5] > 0 else 0.0
    band[:4] = band[:4] + nfg_shift * np.abs(l32[:4])
    d = np.abs(l16 - l32)
    print(f'[{tag}] losses f32 {np.round(losses32, 5)}\n[{tag}] losses bf16 {np.round(losses16, 5)}\n[{tag}] |diff| {np.round(d, 5)} band {np.round(band, 5)}')
    for i, k in enumerate(KEYS[:4]):
        assert d[i] <= band[i], f'{k}: |bf16 - f32| = {d[i]:.5f} > {band[i]:.5f}'
    # num_fg is a ratio of counts: the reference's own 16-bit runs move it by (see the fixture); allow the same + 3 %
    assert abs(losses16[5] - losses32[5]) <= band[5] + 3e-2 * abs(losses32[5]), ('num_fg', losses16[5], losses32[5])
    # ---- stage features at the last timestep (= final LSTM h) and cell states: SLACK x class and the 2e-2 of SURVEY 8c
    hb, cb = SLACK * cls(g, prefix, 'state_h_rel'), SLACK * cls(g, prefix, 'state_c_rel')
    hd = np.array([rel(a, b) for a, b in zip(h16, h32)])
    cd = np.array([rel(a, b) for a, b in zip(c16, c32)])
    print(f'[{tag}] stage feature rel dev {np.round(hd, 5)} (class x{SLACK}: {np.round(hb, 5)})\n[{tag}] cell state rel dev {np.round(cd, 5)} (class x{SLACK}: {np.round(cb, 5)})')
    assert np.all(hd <= np.minimum(hb, 2e-2)) and np.all(cd <= np.minimum(cb, 2e-2)), (hd, hb, cd, cb)
    # the same deviations as multiples of the reference's fp16 class (the dtype it trains with)
    h16, c16cls = cls16(g, prefix, 'state_h_rel'), cls16(g, prefix, 'state_c_rel')
    mh, mc = hd / h16, cd / c16cls
    print(f'[{tag}] ... as multiples of the reference fp16-autocast class: stage features x{np.round(mh, 1)} (class {np.round(h16, 5)}), '
          f'cell states x{np.round(mc, 1)} (class {np.round(c16cls, 5)})')
    MEASURED[tag] = {'feat_x_fp16': mh.tolist(), 'state_c_x_fp16': mc.tolist()}
    assert np.all(mh <= MULT['feat']) and np.all(mc <= MULT['state_c']), (mh, mc)
    # ---- gradients: global direction, then per tensor
    flat16 = np.concatenate([grads16[n].ravel() for n in names])
    flat32 = np.concatenate([grads32[n].ravel() for n in names])
    c_hip, r_hip = cos(flat16, flat32), rel(flat16, flat32)
    c_cls = min(float(g[f'{prefix}_ac_grad_cos_global']), float(g[f'{prefix}_acf_grad_cos_global']))
    r_cls = max(float(g[f'{prefix}_ac_grad_rel_global']), float(g[f'{prefix}_acf_grad_rel_global']))
    print(f'[{tag}] gradient cosine bf16 vs f32 {c_hip:.4f} (reference autocast vs fp32: ac {float(g[f"{prefix}_ac_grad_cos_global"]):.4f}, '
          f'acf {float(g[f"{prefix}_acf_grad_cos_global"]):.4f}); rel dev {r_hip:.4f} (class {r_cls:.4f})')
    c16 = min(float(g[f'{prefix}_h16_grad_cos_global']), float(g[f'{prefix}_h16f_grad_cos_global']))
    m_cos = (1.0 - c_hip) / max(1.0 - c16, 1e-12)
    print(f'[{tag}] ... reference fp16 autocast vs fp32: cosine h16 {float(g[f"{prefix}_h16_grad_cos_global"]):.4f}, h16f '
          f'{float(g[f"{prefix}_h16f_grad_cos_global"]):.4f}; (1 - cos) of HIP bf16 = x{m_cos:.1f} the fp16 class')
    MEASURED[tag]['one_minus_cos_x_fp16'] = m_cos
    assert m_cos <= MULT['one_minus_cos'], m_cos
    assert 1.0 - c_hip <= SLACK * (1.0 - c_cls), f'gradient cosine {c_hip:.4f}: outside {SLACK} x the reference class ({c_cls:.4f})'
    assert r_hip <= SLACK * r_cls + nfg_shift
    dev = np.array([rel(grads16[n], grads32[n]) for n in names])
    bound = SLACK * cls(g, prefix, 'grad_rel') + nfg_shift
    # tensors whose gradient is rounding noise in EVERY run (key part of a qkv bias under the shift-invariant softmax, ...) have
    # class deviations ~1 and say nothing; a tensor counts when the reference class itself resolves it
    ratio = dev / np.maximum(bound, 1e-12)
    order = np.argsort(-ratio)
    print(f'[{tag}] per-tensor grad rel dev / ({SLACK} x class): median {np.median(ratio):.3f}, 90 % {np.quantile(ratio, .9):.3f}, max {ratio.max():.3f}; '
          f'{int((ratio > 1).sum())} of {len(names)} tensors above 1')
    for i in order[:8]:
        print(f'    {ratio[i]:.3f}  dev {dev[i]:.4f}  class {(bound[i] - nfg_shift) / SLACK:.4f}  {names[i]}')
    assert (ratio <= 1.0).mean() >= frac_ok, f'only {(ratio <= 1.0).mean():.3f} of the tensors are within {SLACK} x the reference class'
    assert np.median(ratio) <= 1.0 / SLACK + 0.15       # typical tensor: no further from fp32 than the reference's own 16-bit run (+15 %)
    return c_hip


def _micro_run(manifest, mode):
    from leod_amd.engine import TrainEngine
    with precision(mode):
        det, _ = te.micro_detector(manifest, 9)
        eng = TrainEngine(det, lr=2e-4, total_steps=1000, clip_value=0.0)     # raw gradients (value clipping saturates many entries)
        ev, labels, label_tb, is_first = te.g12_inputs(0)
        losses = eng.step(ev.to(DEV), labels.to(DEV), label_tb, torch.ones(2, dtype=torch.bool, device=DEV))
        grads = {n: p.grad.detach().cpu().numpy().copy() for n, p in det.named_parameters()}
        return (np.array([float(losses[k]) for k in KEYS]), grads, [h.detach().cpu().numpy() for h, _ in eng.states],
                [c.detach().cpu().numpy() for _, c in eng.states])


@pytest.mark.parametrize('mode16', MODES16)
def test_micro_step_bf16_within_reference_autocast_class(gpu, manifest, g18, mode16):
    """The g12 micro training step (T=5, B=2): fp32 mode == the reference's recorded fp32 run; bf16 mode within SLACK x the
    deviation of the reference's own autocast runs from it, per loss component, stage feature, LSTM state and parameter gradient."""
    l32, g32, h32, c32 = _micro_run(manifest, 'f32')
    names = [str(k) for k in g18['micro_grad_keys']]
    np.testing.assert_allclose(l32, g18['micro_fp32_losses'], rtol=1e-4, err_msg='fp32 mode vs the reference fp32 run')
    np.testing.assert_allclose([np.linalg.norm(g32[n].astype(np.float64)) for n in names], g18['micro_fp32_grad_norms'], rtol=3e-3, atol=1e-6)
    l16, g16, h16, c16 = _micro_run(manifest, mode16)
    _check_against_class('micro', g18, 'micro', names, g16, g32, l16, l32, h16, h32, c16, c32, frac_ok=0.9, mode=mode16)


@pytest.mark.parametrize('mode16', MODES16)
def test_tiny256_forward_bf16_within_reference_autocast_class(gpu, manifest, g18, mode16):
    """RVT-tiny at the real Gen1 geometry, two timesteps (the g04 set-up): stage features of the bf16 mode against the fp32 mode,
    bounded by the reference's autocast-vs-fp32 deviation of the same features and by 2e-2."""
    ev = synth_events(2, 1, 20, 240, 304, seed=5, as_uint8=True).to(DEV)
    out = {}
    for mode in ('f32', mode16):
        with precision(mode), torch.no_grad():
            det, _, _ = tm.build(manifest, 'tiny_gen1', 6, 'tiny')
            mode = 'bf16' if mode == mode16 else mode            # the 16-bit run is stored under 'bf16' whichever mode it is
            feats, states = det.forward_backbone(ev[0], None)
            feats, states = det.forward_backbone(ev[1], states)
            out[mode] = {k: v.float().cpu().numpy() for k, v in feats.items()}
    ks = sorted(out['f32'])
    np.testing.assert_allclose([np.linalg.norm(out['f32'][k].astype(np.float64)) for k in ks], g18['tiny_fp32_feat_norms'], rtol=2e-4)
    dev = np.array([rel(out['bf16'][k], out['f32'][k]) for k in ks])
    mx = np.array([np.abs(out['bf16'][k] - out['f32'][k]).max() / np.abs(out['f32'][k]).max() for k in ks])
    print('tiny256 feature rel dev', dev, 'class', cls(g18, 'tiny', 'feat_rel'), 'max-rel', mx, 'class', cls(g18, 'tiny', 'feat_maxrel'))
    m16 = dev / cls16(g18, 'tiny', 'feat_rel')
    print('tiny256 ... as multiples of the reference fp16-autocast class', np.round(m16, 1), '(class', cls16(g18, 'tiny', 'feat_rel'), ')')
    MEASURED[f'tiny256/{mode16}'] = {'feat_x_fp16': m16.tolist()}
    assert np.all(m16 <= FP16_CLASS_MULT[mode16]['feat']), m16
    assert np.all(dev <= np.minimum(SLACK * cls(g18, 'tiny', 'feat_rel'), 2e-2))
    assert np.all(mx <= SLACK * cls(g18, 'tiny', 'feat_maxrel'))


def _bench_workload():
    """bench.make_batch(seed 7) + synth weights seed 0: the workload g18 'small_*' was recorded on (and bench.py's cpu_baseline)."""
    import bench
    T, B, hw = 21, 8, (240, 304)
    ev, _, label_tb, labs = bench.make_batch(T, B, hw, 2, 7, 'cpu', (4, 9, 14, 19))
    nmax = max(len(l) for l in labs)
    rows = np.zeros((len(labs), nmax, 7), np.float32)
    for i, l in enumerate(labs):
        rows[i, :len(l)] = l
    return ev.to(DEV), rows, label_tb


def _small_run(manifest, mode, ev, rows, label_tb):
    from leod_amd.modules.utils.detection import Mode
    from leod_amd.optim import fit_step
    mod, opt, lrs = te._full_size_module(0)                  # Module.setup applies the configured mode: switch afterwards
    with precision(mode):
        from leod_amd import ops
        assert ops.get_precision() == mode
        mod.mdl.load_state_dict(synth_state_dict(manifest['small_gen1'], 0))
        opt.clip_value = 0.0
        first = torch.ones(8, dtype=torch.bool, device=DEV)
        out = fit_step(mod, opt, lrs, te._loader_batch(ev, rows, label_tb, first))
        losses = np.array([float(out['log_dict'][f'train/{k}'].detach()) for k in KEYS])
        grads = {n: p.grad.detach().cpu().numpy().copy() for n, p in mod.mdl.named_parameters()}
        states = mod.mode_2_rnn_states[Mode.TRAIN].get_states(0)
        res = (losses, grads, [h.detach().cpu().numpy() for h, _ in states], [c.detach().cpu().numpy() for _, c in states])
    del mod, opt, lrs, out
    torch.cuda.empty_cache()
    return res


@pytest.mark.parametrize('mode16', MODES16)
def test_full_size_step_bf16_within_reference_autocast_class(gpu, manifest, g18, mode16):
    """BASELINE configs[1] (RVT-S Gen1 240x304 T=21 bs=8, bench.py's batch, synthetic weights) through Module.training_step +
    FlatAdamW.  fp32 mode == the REFERENCE's recorded fp32 run of this very workload (six losses 2e-5, 259 gradient norms 3e-3); the
    bf16 mode -- the mode bench.py's headline number is measured in -- within SLACK x the reference's own autocast deviation.
    Measured on MI355X (round 3): gradient cosine bf16 vs f32 0.934; the reference's own autocast vs fp32: 0.916 (ac) / 0.939 (acf); per-tensor
    deviations: median 0.59 x, 90 % quantile 0.72 x of SLACK x class, one tensor of 259 above it (obj_preds.2.bias, 1.8 % vs 0.8 %)."""
    ev, rows, label_tb = _bench_workload()
    l32, g32, h32, c32 = _small_run(manifest, 'f32', ev, rows, label_tb)
    names = [str(k) for k in g18['small_grad_keys']]
    np.testing.assert_allclose(l32[:5], g18['small_fp32_losses'][:5], rtol=2e-5, atol=1e-6, err_msg='fp32 mode vs the reference fp32 run')
    assert l32[5] == pytest.approx(float(g18['small_fp32_losses'][5]), rel=1e-6)
    np.testing.assert_allclose([np.linalg.norm(g32[n].astype(np.float64)) for n in names], g18['small_fp32_grad_norms'], rtol=3e-3, atol=1e-7)
    l16, g16, h16, c16 = _small_run(manifest, mode16, ev, rows, label_tb)
    _check_against_class('small', g18, 'small', names, g16, g32, l16, l32, h16, h32, c16, c32, frac_ok=0.97, mode=mode16)


def _device_batch(T, B, seed, label_ts=(4, 9, 14, 19)):
    """A synthetic batch drawn on the device (same distribution as bench.make_batch; 30 of them in seconds)."""
    g = torch.Generator(device=DEV).manual_seed(seed)
    mask = torch.rand((T, B, 20, 240, 304), generator=g, device=DEV) < 0.08
    ev = (mask * torch.randint(1, 10, (T, B, 20, 240, 304), generator=g, device=DEV)).to(torch.uint8)
    rng = np.random.RandomState(seed)
    label_tb, rows = [], []
    for t in range(T):
        idx = list(range(B)) if t in label_ts else []
        label_tb.append(idx)
        for _ in idx:
            n = rng.randint(1, 7)
            w, h = rng.uniform(10, 90, n), rng.uniform(10, 70, n)
            x, y = rng.uniform(0, 303 - w), rng.uniform(0, 239 - h)
            rows.append(np.stack([rng.randint(0, 2, n), x + w / 2, y + h / 2, w, h, np.ones(n), np.ones(n)], 1).astype(np.float32))
    nmax = max(len(r) for r in rows)
    lab = np.zeros((len(rows), nmax, 7), np.float32)
    for i, r in enumerate(rows):
        lab[i, :len(r)] = r
    return ev, lab, label_tb


@pytest.mark.parametrize('mode16', MODES16)
def test_loss_trajectory_30_steps_bf16_vs_f32(gpu, mode16):
    """30 optimiser steps at the benchmark size from the same random initialisation on identical batches (10 distinct batches, three
    passes) at the START of the reference's 400 k-step OneCycle schedule (the warm-up from max_lr / 20 = 1e-5; ``training.max_steps`` below
    does not reach the scheduler -- the 200-step test that follows runs a whole scaled schedule), carried LSTM state on the stream half: the two
    precision modes must learn the same way -- the only proxy for "mAP within +-0.3" this image allows (no data, no checkpoints)."""
    from leod_amd.config import full_config, dynamically_modify_train_config
    from leod_amd.modules.utils.fetch import fetch_model_module
    from leod_amd.optim import fit_step
    T, B, steps = 21, 8, 30
    batches = [_device_batch(T, B, 100 + i) for i in range(10)]
    g = torch.Generator().manual_seed(5)
    firsts = [torch.ones(B, dtype=torch.bool)]
    for s in range(1, steps):
        m = torch.ones(B, dtype=torch.bool)
        m[:B // 2] = torch.rand(B // 2, generator=g) < 0.05
        firsts.append(m)
    traj = {}
    for mode in ('f32', mode16):
        cfg = dynamically_modify_train_config(full_config('gen1', 'small'))
        cfg.training.max_steps = 60
        torch.manual_seed(0)
        mod = fetch_model_module(cfg).to(DEV)
        mod.setup('fit')                                     # applies the configured mode: switch afterwards
        mod.train()
        oc = mod.configure_optimizers()
        opt, lrs = oc['optimizer'], oc['lr_scheduler']['scheduler']
        with precision(mode):
            from leod_amd import ops
            assert ops.get_precision() == mode
            out = []
            for s in range(steps):
                ev, lab, label_tb = batches[s % len(batches)]
                res = fit_step(mod, opt, lrs, te._loader_batch(ev, lab, label_tb, firsts[s].to(DEV)), s)
                out.append([float(res['log_dict'][f'train/{k}'].detach()) for k in KEYS])
            traj['f32' if mode == 'f32' else 'bf16'] = np.array(out)
        del mod, opt, lrs, oc
        torch.cuda.empty_cache()
    a, b = traj['f32'][:, 0], traj['bf16'][:, 0]
    print('loss f32 :', np.round(a, 3))
    print('loss bf16:', np.round(b, 3))
    sm = lambda x: np.convolve(x, np.ones(5) / 5, mode='valid')          # noqa: E731   5-step moving average (26 points)
    sa, sb = sm(a), sm(b)
    print('smoothed rel diff:', np.round(np.abs(sa - sb) / sa, 4))
    assert not np.array_equal(a, b), 'the two runs are bit-identical: the precision switch did not take effect'
    # both learn: the mean loss of every pass over the 10 batches lies below the previous pass's (measured: 16.32 -> 16.08 -> 15.89; a
    # random-init head is dominated by the objectness term over 1680 anchors x 32 frames, which falls slowly) ...
    pa = [a[i:i + 10].mean() for i in (0, 10, 20)]
    pb = [b[i:i + 10].mean() for i in (0, 10, 20)]
    print('pass means f32', np.round(pa, 4), 'bf16', np.round(pb, 4))
    assert pa[0] > pa[1] > pa[2] and pb[0] > pb[1] > pb[2], (pa, pb)
    # ... along the same curve: smoothed losses within 1 % of each other at every point, the drop from pass 1 to pass 3 within 25 % of
    # each other, final level within 1 %
    assert np.all(np.abs(sa - sb) <= 1e-2 * sa), np.abs(sa - sb) / sa
    assert abs((pa[0] - pa[2]) - (pb[0] - pb[2])) <= 0.25 * (pa[0] - pa[2]), (pa, pb)
    assert abs(a[-5:].mean() - b[-5:].mean()) <= 1e-2 * a[-5:].mean()
    # the component losses follow too (iou, conf, cls), at the last pass
    for i in (1, 2, 3):
        x, y = traj['f32'][20:, i].mean(), traj['bf16'][20:, i].mean()
        assert abs(x - y) <= 5e-2 * abs(x) + 1e-3, (KEYS[i], x, y)


@pytest.mark.parametrize('mode16', MODES16)
def test_loss_trajectory_200_steps_onecycle_bf16_vs_f32(gpu, mode16):
    """a WHOLE OneCycle schedule, scaled to 200 optimiser steps (20 warm-up steps from max_lr / 20 to the 2e-4 peak, 180
    steps of linear decay -- the reference's shape, modules/detection.py:498-511, with pct_start 0.1 instead of 0.005 so that the warm-up is
    more than one step), at the benchmark size on identical batches (16 distinct batches cycled, carried LSTM state on the stream half):
    the bf16 mode must train like the fp32 mode along the whole schedule.  Launch plans replay both runs from step 2 on."""
    from leod_amd.config import full_config, dynamically_modify_train_config
    from leod_amd.modules.utils.fetch import fetch_model_module
    from leod_amd.optim import fit_step
    T, B, steps = 21, 8, 200
    batches = [_device_batch(T, B, 300 + i) for i in range(16)]
    g = torch.Generator().manual_seed(6)
    firsts = [torch.ones(B, dtype=torch.bool)]
    for s in range(1, steps):
        m = torch.ones(B, dtype=torch.bool)
        m[:B // 2] = torch.rand(B // 2, generator=g) < 0.05
        firsts.append(m)
    traj, lrs_seen = {}, {}
    # controls: a second fp32 run (what the run-to-run order of the fp32 atomics alone does to a trajectory) and an fp32 run whose initial
    # weights are perturbed ONCE by relative Gaussian noise of 2^-9 (one bf16 rounding): how far a perturbation of that size is amplified
    # by 200 steps of training from random init
    for tag in ('f32', 'f32_again', 'f32_perturbed', 'bf16'):
        mode = mode16 if tag == 'bf16' else 'f32'
        cfg = dynamically_modify_train_config(full_config('gen1', 'small'))
        cfg.training.max_steps = steps
        cfg.training.lr_scheduler.total_steps = steps
        cfg.training.lr_scheduler.pct_start = 0.1
        torch.manual_seed(0)
        mod = fetch_model_module(cfg).to(DEV)
        mod.setup('fit')
        mod.train()
        oc = mod.configure_optimizers()
        opt, lrs = oc['optimizer'], oc['lr_scheduler']['scheduler']
        if tag == 'f32_perturbed':
            gp = torch.Generator(device=DEV).manual_seed(11)
            opt.flat.data.mul_(1.0 + 2.0 ** -9 * torch.randn(opt.flat.data.shape, generator=gp, device=DEV))
            opt.flat.touch()
        with precision(mode):
            out, lr = [], []
            for s in range(steps):
                ev, lab, label_tb = batches[s % len(batches)]
                lr.append(opt.param_groups[0]['lr'])
                res = fit_step(mod, opt, lrs, te._loader_batch(ev, lab, label_tb, firsts[s].to(DEV)), s)
                out.append(res['log_dict']['train/loss'].detach())
            traj[tag] = torch.stack(out).cpu().numpy().astype(np.float64)
            lrs_seen[tag] = np.array(lr)
        if mod.plan_mode:
            assert mod._plans.replays >= steps - 8, (mod._plans.replays, mod._plans.captures)
        del mod, opt, lrs, oc
        torch.cuda.empty_cache()
    assert lrs_seen['f32'][0] < 1.1e-5 and abs(lrs_seen['f32'].max() - 2e-4) < 1e-6 and lrs_seen['f32'][-1] < 5e-6     # the schedule ran
    a, b, a2, ap = traj['f32'], traj['bf16'], traj['f32_again'], traj['f32_perturbed']
    sm = lambda x: np.convolve(x, np.ones(10) / 10, mode='valid')         # noqa: E731
    sa, sb, sa2, sap = sm(a), sm(b), sm(a2), sm(ap)
    rel_d = np.abs(sa - sb) / sa
    noise = np.abs(sa - sa2) / sa
    pert = np.abs(sa - sap) / sa
    print('loss f32 perturbed    :', np.round(ap[::20], 3))
    print(f'fp32 with 2^-9 initial perturbation, smoothed relative difference: max {pert.max():.4f}, mean {pert.mean():.4f}; last 20-step mean {ap[-20:].mean():.3f}')
    print('loss f32  (every 20th):', np.round(a[::20], 3))
    print('loss f32 again        :', np.round(a2[::20], 3))
    print('loss bf16 (every 20th):', np.round(b[::20], 3))
    print(f'fp32 run-to-run smoothed relative difference: max {noise.max():.4f}, mean {noise.mean():.4f}')
    print(f'[{mode16}] smoothed relative difference: max {rel_d.max():.4f}, mean {rel_d.mean():.4f}; first / last 20-step means f32 {a[:20].mean():.3f} / '
          f'{a[-20:].mean():.3f}, bf16 {b[:20].mean():.3f} / {b[-20:].mean():.3f}')
    MEASURED['trajectory200'] = {'max_smoothed_rel_diff': float(rel_d.max()), 'final_f32': float(a[-20:].mean()), 'final_bf16': float(b[-20:].mean())}
    assert not np.array_equal(a, b)
    assert a[-20:].mean() < 0.65 * a[:20].mean() and b[-20:].mean() < 0.65 * b[:20].mean()          # both learn (16.1 -> ~9)
    # Measured on MI355X (round 4; 11 bf16 and 9 fp32 runs, eager and plan-replayed alike -- tools/traj_probe.py): the last-20-step mean of
    # the fp32 mode lands at 8.87-9.40, of the bf16 mode at 7.47-8.91 (mean 8.1): the curves agree within 2-3 % through warm-up and decay
    # up to step ~130 (fp32's own run-to-run spread there: 1-4 %) and then, while the 16 cycled batches are being fitted, the bf16 mode
    # ends ~10 % LOWER with three times the run-to-run spread.  Not a property of the launch plans (eager runs spread the same way);
    # stated in DESIGN.md as an open difference of the mode, bounded here.
    # (the bounds are those of a chaotic system observed over ~35 runs -- worst seen: 0.058 before step 100, 0.30 overall, finals 6.69 vs 9.53 --
    # with head room: this test documents the difference and catches a mode that stops learning or runs away, it cannot pin a trajectory.
    # What the difference is and is not: profiles/r04_z_trajectory_ablation.txt)
    # Mode 16f (round 5: the forward pass in the reference's own fp16 class, per-step gradient cosine 0.9993 against fp32): five runs on
    # MI355X (profiles/r05_g_trajectory_16f_runs.txt) -- the curves agree with fp32 within 1.2-1.8 % up to step ~130 (fp32 run-to-run in the
    # same runs: 0.8-5 %); the last-20-step means then land at 9.16 / 8.76 / 8.38 / 7.60 / ... against fp32's 8.70-9.30 and the perturbed-fp32
    # control's 8.67-8.88: the end of the schedule, where the 16 cycled batches are memorised, is a chaotic observable of THIS test problem
    # (a 2^-9 perturbation of the initial weights moves fp32 itself by up to 6 %), and the 16-bit modes keep landing on the low side of it.
    # The early bound is the parity statement (0.10: two fp32 runs have differed by 0.058 there; 16f measured 0.010-0.027 in eight runs); the
    # whole-schedule bound catches a mode that stops learning or runs away.
    early = slice(0, 90)
    lim_early, lim_all = 0.10, 0.45          # both modes: two fp32 runs have differed by up to 0.058 before step 90 (round 4); 16f measured 0.010-0.027 in eight runs
    assert rel_d[early].max() <= lim_early, rel_d[early].max()
    assert rel_d.max() <= lim_all, rel_d.max()
    assert abs(a[-20:].mean() - b[-20:].mean()) <= lim_all * a[-20:].mean()
    MEASURED['trajectory200'].update(fp32_run_to_run=float(noise.max()), fp32_perturbed=float(pert.max()), early_max=float(rel_d[early].max()))
    print('MEASURED', MEASURED['trajectory200'])


@pytest.mark.parametrize('mode16', MODES16)
def test_full_size_plans_equal_eager_in_16bit_modes(gpu, mode16):
    """The benchmarked executor in the benchmarked arithmetic (plan-vs-eager parity where the ConvLSTM weight packs of C = 192 /
    384 and the 16-bit weight shadows are live): four optimisation steps of BASELINE configs[1] in a 16-bit mode, eager vs launch plans
    (step 0 eager, step 1 recorded, steps 2-3 replayed, a labelled-frame count that changes at step 3), identical batches.  Plans launch
    the SAME kernels: what differs is the order of fp32 atomics -- losses agree to 2e-3 (the 16-bit step is as chaotic as its fp32 twin: a
    noise-level gradient may flip an Adam sign), final cell states to 2e-2 of their magnitude."""
    from leod_amd.optim import fit_step
    from leod_amd.modules.utils.detection import Mode
    res = {}
    firsts = [torch.ones(8, dtype=torch.bool), torch.tensor([False, True, False, False, True, False, True, True]),
              torch.tensor([False, False, True, False, False, True, False, False]), torch.tensor([False] * 8)]
    for plan in (False, True):
        mod, opt, lrs = te._full_size_module(0)
        mod.plan_mode = plan
        with precision(mode16):
            out = []
            for step in range(4):
                ev, labels, label_tb = te._full_size_batch(seed=21 + step)
                rows = labels.cpu().numpy()
                if step == 3:                                  # 24 instead of 32 labelled frames: another head plan under the same backbone plan
                    keep = [t for t in range(21) if label_tb[t]][:3]
                    n_keep = sum(len(label_tb[t]) for t in keep)
                    label_tb = [label_tb[t] if t in keep else [] for t in range(21)]
                    rows = rows[:n_keep]
                if plan and step >= 2:
                    # replayed steps read the batch's own event tensor in place (leod_plan_rebase_input re-points the stem convolution and its weight
                    # gradient): the buffer the plans were captured with is poisoned here, so any kernel still reading it would show in the losses
                    from leod_amd.modules.step_plan import BackbonePlan
                    bb = [e for e in mod._plans.entries.values() if isinstance(e, BackbonePlan)][0]
                    assert bb.rebase_ok and bb.rebase_count == 2, (bb.rebase_ok, bb.rebase_count)
                    if bb.rebase_ok:
                        bb.ev.fill_(255)
                r = fit_step(mod, opt, lrs, te._loader_batch(ev, rows, label_tb, firsts[step].to(DEV)), step)
                out.append([float(r['log_dict'][f'train/{k}'].detach()) for k in KEYS])
            states = mod.mode_2_rnn_states[Mode.TRAIN].get_states(0)
            res[plan] = (np.array(out), [c.detach().cpu().numpy() for _, c in states])
            if plan:
                pl = mod._plans
                assert (pl.captures, pl.head_captures, pl.steps, pl.replays, pl.eager_steps) == (1, 2, 3, 1, 1), pl.info()
                assert (bb.ev_now is not bb.ev) == bb.rebase_ok
        del mod, opt, lrs
        torch.cuda.empty_cache()
    a, b = res[True][0], res[False][0]
    print(f'[{mode16}] losses through plans', np.round(a[:, 0], 4), 'eager', np.round(b[:, 0], 4))
    # total loss: steps 0-1 equal to 5 digits; from step 2 on BOTH executors are bimodal from run to run in the bf16 mode (17.269 | 17.346 at step 2, whichever
    # path: one SimOTA assignment on a knife edge after two noisy updates), so a plan run may meet an eager run of the other branch (0.45 %)
    np.testing.assert_allclose(a[:2, 0], b[:2, 0], rtol=1e-4)
    np.testing.assert_allclose(a[2:, 0], b[2:, 0], rtol=1e-2)
    np.testing.assert_allclose(a[:, 1:4], b[:, 1:4], rtol=3e-2, atol=1e-3)     # components: a SimOTA assignment may flip after three noisy updates
    np.testing.assert_allclose(a[:, 5], b[:, 5], rtol=3e-2)
    for x, y in zip(res[True][1], res[False][1]):
        assert np.abs(x - y).max() <= 2e-2 * np.abs(y).max()


@pytest.mark.parametrize('mode16', MODES16)
@pytest.mark.parametrize('size,full_res,T', [('base', False, 2), ('base', True, 2), ('small', True, 1), ('tiny', False, 2)])
def test_gen4_geometries_fwd_bwd_bf16(gpu, g18, size, full_res, T, mode16):
    """The geometries of BASELINE configs[3] in the bf16 mode (tests/test_model_gpu.py::test_gen4_geometries_fwd_bwd runs them in
    fp32 mode): RVT-B / RVT-S / RVT-T on Gen4 frames at 384x640 (60-token partitions) and 768x1280 (240-token partitions), carried
    LSTM state, against the fp32 ORACLE: stage features and states 2e-2 (L2) / the reference class's worst-element deviation,
    the smooth test loss 1e-2, every backbone gradient tensor within 4 % (L2) with cosine > 0.999 -- there is no SimOTA in this
    loss, so gradient noise is the kernels' rounding only."""
    with precision(mode16):
        det, sd, cfg = tm._build_gen4(size, full_res, 21)
        in_hw = tuple(cfg.model.backbone.in_res_hw)
        part = tuple(cfg.model.backbone.stage.attention.partition_size)
        hw = (720, 1280) if full_res else (360, 640)
        E, dh = {'base': (64, 32), 'small': (48, 24), 'tiny': (32, 32)}[size]
        ocfg = ot.model_cfg(E, dh, 0.67 if size == 'base' else 0.33, part, num_classes=3, in_res_hw=in_hw)
        ev = synth_events(T, 1, 20, hw[0], hw[1], seed=31, as_uint8=True)
        states = None
        for t in range(T):
            feats, states = det.forward_backbone(ev[t].to(DEV), states)
        loss = sum((v ** 2).mean() for v in feats.values())
        loss.backward()
        torch.cuda.synchronize()
    osd = {k: v.clone() for k, v in sd.items()}
    bkeys = [k for k in osd if k.startswith('backbone.') and osd[k].is_floating_point()]
    for k in bkeys:
        osd[k].requires_grad_(True)
    evp = ob.pad_ev_repr(ev.float(), in_hw)
    ostates = None
    for t in range(T):
        ofeats, ostates = ob.backbone_forward(evp[t], ostates, osd, ocfg)
    oloss = sum((v ** 2).mean() for v in ofeats.values())
    oloss.backward()

    # worst single element: the reference's own 16-bit runs move single feature elements by 2-6 % of the map's magnitude in stages
    # 1-2 and up to 10-15 % in stages 3-4 (g18 *_feat_maxrel); the bound per stage is SLACK x the largest of those recordings
    wmax = SLACK * np.max([np.asarray(g18[f'{p}_{m}_feat_maxrel']) for p in ('micro', 'tiny', 'small') for m in ('ac', 'acf')], axis=0)

    def chk(a, b, what, stage):
        a, b = a.detach().float().cpu().numpy(), b.detach().numpy()
        r, m = rel(a, b), float(np.abs(a - b).max() / np.abs(b).max())
        assert r <= 2e-2 and m <= wmax[stage], f'{what}: rel {r:.4f} worst element {m:.4f} (bound {wmax[stage]:.4f})'
        return r, m

    devs = [chk(feats[s], ofeats[s], f'stage {s}', i) for i, s in enumerate(sorted(ofeats))]
    for i, ((h, c), (oh_, oc)) in enumerate(zip(states, ostates)):
        chk(h, oh_, 'h', i)
        chk(c, oc, 'c', i)
    assert float(loss) == pytest.approx(float(oloss), rel=1e-2)
    params = dict(det.named_parameters())
    stats = []
    for k in bkeys:
        a, b = params[k].grad.detach().cpu().numpy(), osd[k].grad.numpy()
        stats.append((rel(a, b), cos(a, b), k))
    worst = sorted(stats, reverse=True)[:5]
    print(f'{size} full_res={full_res}: feature (rel dev, worst element) {np.round(devs, 4).tolist()}; worst gradient tensors', [(round(r, 4), round(c, 5), k) for r, c, k in worst])
    for r, c, k in stats:
        # measured on MI355X: worst tensor 1.3 % (L2), cosine 0.9999 (LayerScale gammas and LayerNorm weights of stage 4)
        assert r <= 0.04 and c >= 0.999, (k, r, c)


@pytest.mark.parametrize('mode16', MODES16)
def test_pseudo_label_inference_bf16_vs_oracle(gpu, manifest, mode16):
    """The pseudo-label pass (tests/test_engine_gpu.py::test_pseudo_label_inference_vs_oracle) in the bf16 mode, the mode
    tools/bench_pseudo.py quotes its rate in: hflip-TTA inference + postprocess / NMS + pred2label against the fp32 oracle.  Scores
    near a threshold may cross it under 16-bit rounding: per frame the keep count may drift by max(2, 10 %); every HIP box must have
    an oracle partner of the same class within 3e-2 of the frame size (boxes) / 3e-2 absolute (scores), and vice versa, up to the
    drift allowance."""
    from leod_amd.engine import PseudoLabelEngine
    ev = synth_events(4, 2, 20, 60, 90, seed=3, as_uint8=True)
    with precision(mode16):
        det, sd = te.micro_detector(manifest, 5)
        pl = PseudoLabelEngine(det, 2, conf_thre=0.01, obj_thresh=[0.1, 0.05], cls_thresh=[0.1, 0.05], hflip=True, max_det=126)
        lab, lcnt, dets, cnt = pl.step(ev.to(DEV))
        dets, cnt, lab, lcnt = dets.cpu().numpy(), cnt.cpu().numpy(), lab.cpu().numpy(), lcnt.cpu().numpy()
    rdets, _, _ = ot.infer_sequence(sd, te.MICRO, ev, conf_thre=0.01, hflip=True)
    rl = op.pred2label([r.clone() for r in rdets], [0.1, 0.05], [0.1, 0.05], 'gen1', False)
    tot = dict(ref=0, got=0, unmatched=0)

    def match(got, ref, box_cols, score_cols, cls_col, scale):
        """greedy one-to-one matching; returns the number of rows on either side left without a partner"""
        if len(ref) == 0 or len(got) == 0:
            return len(ref) + len(got)
        db = np.abs(got[None, :, box_cols] - ref[:, None, box_cols]).max(-1) / scale
        ds = np.abs(got[None, :, score_cols] - ref[:, None, score_cols]).max(-1)
        ok = (db <= 3e-2) & (ds <= 3e-2) & (got[None, :, cls_col] == ref[:, None, cls_col])
        used, un = set(), 0
        for i in range(len(ref)):
            js = [j for j in np.argsort(db[i]) if ok[i, j] and j not in used]
            if js:
                used.add(js[0])
            else:
                un += 1
        return un + (len(got) - len(used))

    for i, r in enumerate(rdets):
        r = r.numpy()
        n = int(cnt[i])
        allow = max(2, int(0.1 * len(r)))
        assert abs(n - len(r)) <= allow, f'frame {i}: kept {n} boxes, oracle {len(r)}'
        un = match(dets[i, :n], r, [0, 1, 2, 3], [4, 5], 6, 96.0)
        assert un <= 2 * allow, f'frame {i}: {un} detections without a partner (kept {n}, oracle {len(r)})'
        tot['ref'] += len(r); tot['got'] += n; tot['unmatched'] += un
    for i, r in enumerate(rl):
        r = r.numpy()
        n = int(lcnt[i])
        assert abs(n - len(r)) <= max(1, int(0.1 * len(r))), f'frame {i}: {n} pseudo labels, oracle {len(r)}'
        # label rows: (t, x, y, w, h, class_id, class_confidence, objectness)
        un = match(lab[i, :n], r, [1, 2, 3, 4], [6, 7], 5, 96.0)
        assert un <= 2 * max(1, int(0.1 * len(r)))
    print('pseudo-label pass bf16 vs oracle:', tot)
    assert tot['ref'] > 20 and tot['unmatched'] <= 0.05 * (tot['ref'] + tot['got'])
