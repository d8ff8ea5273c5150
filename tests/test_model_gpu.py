"""Model-level parity on the GPU: the HIP-backed modules (reference API, reference state-dict keys)
against the golden vectors recorded from the reference and against the CPU oracle's autograd.
``pytest -m gpu``."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import backbone as ob  # noqa: E402
from oracle import head as oh  # noqa: E402
from oracle import postproc as op  # noqa: E402
from oracle import train_step as ot  # noqa: E402
from oracle.synth import synth_state_dict, synth_events, synth_labels  # noqa: E402

DEV = 'cuda'


@pytest.fixture(scope='module')
def gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return True


def build(manifest, key, seed, size, dataset='gen1', micro=False, train=False, **head_over):
    from leod_amd.config import full_config, dynamically_modify_train_config, create
    from leod_amd.models.detection.yolox_extension.models.detector import YoloXDetector
    over = {}
    if micro:
        over = dict(model=dict(backbone=dict(embed_dim=16, stage=dict(attention=dict(dim_head=8))), fpn=dict(depth=0.33)))
    cfg = dynamically_modify_train_config(full_config(dataset, size, overrides=over))
    if micro:
        cfg.model.backbone.in_res_hw = (64, 96)
        cfg.model.backbone.stage.attention.partition_size = (2, 3)
    for k, v in head_over.items():
        cfg.model.head[k] = v
    det = YoloXDetector(cfg.model)
    sd = synth_state_dict(manifest[key], seed)
    det.load_state_dict(sd, strict=True)
    det.to(DEV)
    det.train(train)
    return det, sd, cfg


def close(a, b, rtol=1e-4, atol=1e-5, what=''):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    scale = float(np.abs(b).max()) if b.size else 1.0
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol + 1e-5 * scale, err_msg=what)


MICRO = ot.model_cfg(embed_dim=16, dim_head=8, fpn_depth=0.33, partition_size=(2, 3), in_res_hw=(64, 96))


def micro_labels(n_frames, seed, hw=(60, 90)):
    labs = synth_labels(n_frames, hw, 2, seed=seed, max_boxes=4)
    for l in labs:
        l[:, 3] = l[:, 3].clamp(max=30)
        l[:, 4] = l[:, 4].clamp(max=24)
        l[:, 1] = torch.minimum(l[:, 1], hw[1] - 1 - l[:, 3])
        l[:, 2] = torch.minimum(l[:, 2], hw[0] - 1 - l[:, 4])
    return labs


def test_backbone_micro_golden(gpu, golden_dir, manifest):
    g = np.load(os.path.join(golden_dir, 'g04_backbone_micro.npz'))
    det, _, _ = build(manifest, 'micro', 5, 'small', micro=True)
    ev_u8 = synth_events(3, 2, 20, 60, 90, seed=4, as_uint8=True).to(DEV)
    ev_f32 = ob.pad_ev_repr(ev_u8.float(), (64, 96))
    for ev in (ev_u8, ev_f32):          # raw unpadded uint8 (padding folded into the stem) and reference-style padded fp32
        states = None
        with torch.no_grad():
            for t in range(3):
                feats, states = det.forward_backbone(ev[t], states)
                for k, v in feats.items():
                    assert tuple(v.shape) == g[f't{t}_s{k}'].shape
                    close(v, g[f't{t}_s{k}'], what=f't{t} stage{k}')
        for s, (h, c) in enumerate(states):
            close(c, g[f'final_c{s + 1}'])


def test_backbone_tiny256_golden(gpu, golden_dir, manifest):
    g = np.load(os.path.join(golden_dir, 'g04_backbone_tiny256.npz'))
    det, _, _ = build(manifest, 'tiny_gen1', 6, 'tiny')
    ev = synth_events(2, 1, 20, 240, 304, seed=5, as_uint8=True).to(DEV)
    with torch.no_grad():
        feats, states = det.forward_backbone(ev[0], None)
        feats, states = det.forward_backbone(ev[1], states)
    for k, v in feats.items():
        close(v.mean(), g[f's{k}_mean'], rtol=2e-4, atol=1e-6)
        close(v.abs().max(), g[f's{k}_absmax'], rtol=2e-4)
        close(v[0, :8, :4, :5], g[f's{k}_slice'], rtol=2e-4, atol=2e-5)


def test_detector_head_golden_and_grads(gpu, golden_dir, manifest):
    g = np.load(os.path.join(golden_dir, 'g05_head_micro.npz'))
    det, sd, _ = build(manifest, 'micro', 5, 'small', micro=True)

    def rnd(shape, seed):
        return torch.randn(shape, generator=torch.Generator().manual_seed(seed))

    feats_cpu = {2: rnd((3, 32, 8, 12), 51), 3: rnd((3, 64, 4, 6), 52), 4: rnd((3, 128, 2, 3), 53)}
    with torch.no_grad():
        pred, losses = det.forward_detect({k: v.to(DEV) for k, v in feats_cpu.items()})
    assert losses is None
    close(pred, g['pred_eval'], what='eval predictions')
    # training mode: losses, BN buffers and every gradient against the oracle's autograd
    labs = micro_labels(3, seed=7)
    labs[1] = labs[1][:1]
    labs[2][0, 1:5] = torch.tensor([0., 0., 12., 9.])
    targets = op.batched_yolox_labels(labs)
    det.train()
    fg = {k: v.to(DEV).requires_grad_(True) for k, v in feats_cpu.items()}
    pred, losses = det.forward_detect(fg, targets=targets.to(DEV))
    close(pred, g['pred_train'], what='train predictions')
    for k in ('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss', 'num_fg'):
        close(losses[k], g['loss_' + k], rtol=5e-5, what=k)
    dsd = det.state_dict()
    for k in ['fpn.lateral_conv0.bn.running_mean', 'fpn.lateral_conv0.bn.running_var',
              'yolox_head.stems.0.bn.running_mean', 'yolox_head.cls_convs.2.1.bn.running_var']:
        close(dsd[k], g['bn_' + k.replace('.', '_')], rtol=5e-5, atol=1e-6)
    losses['loss'].backward()
    osd = {k: v.clone() for k, v in sd.items()}
    pkeys = [k for k, v in osd.items() if v.is_floating_point() and 'running_' not in k]
    for k in pkeys:
        osd[k].requires_grad_(True)
    fo = {k: v.clone().requires_grad_(True) for k, v in feats_cpu.items()}
    _, olosses = oh.detect_forward(fo, osd, MICRO, labels=targets.clone(), training=True)
    olosses['loss'].backward()
    params = dict(det.named_parameters())
    for k in pkeys:
        if osd[k].grad is None:
            continue
        close(params[k].grad, osd[k].grad, rtol=2e-3, atol=2e-5, what='grad ' + k)
    for k in fo:
        close(fg[k].grad, fo[k].grad, rtol=2e-3, atol=2e-5, what=f'grad feature {k}')
    gk = [str(k) for k in g['grad_keys']]
    mine = np.array([float(params[k].grad.norm()) for k in gk])
    np.testing.assert_allclose(mine, g['grad_norms'], rtol=2e-3, atol=1e-6)


def test_detector_head_with_loss_options_vs_oracle(gpu, manifest):
    """``bbox_loss_weighting`` + ``ignore_bg_k`` (off in the shipped configs, yolo_head.py:335-381) through the whole detection head
    (FPN -> towers -> HeadTailFn): losses and the gradients of the backbone features / parameters against the oracle's autograd."""
    det, sd, _ = build(manifest, 'micro', 5, 'small', micro=True)

    def rnd(shape, seed):
        return torch.randn(shape, generator=torch.Generator().manual_seed(seed))

    feats_cpu = {2: rnd((3, 32, 8, 12), 51), 3: rnd((3, 64, 4, 6), 52), 4: rnd((3, 128, 2, 3), 53)}
    labs = micro_labels(3, seed=7)
    targets = op.batched_yolox_labels(labs)
    g = torch.Generator().manual_seed(9)
    nz = (targets.sum(2) > 0).float()
    targets[:, :, 5] = (0.3 + 0.7 * torch.rand(targets.shape[:2], generator=g)) * nz
    targets[:, :, 6] = (0.3 + 0.7 * torch.rand(targets.shape[:2], generator=g)) * nz
    det.yolox_head.bbox_loss_weighting, det.yolox_head.ignore_bg_k = 'cls-w**2', 0.2
    det.train()
    fg = {k: v.to(DEV).requires_grad_(True) for k, v in feats_cpu.items()}
    _, losses = det.forward_detect(fg, targets=targets.to(DEV))
    losses['loss'].backward()
    assert int(det.yolox_head.last_assignment['ignore_mask'].sum()) > 0
    osd = {k: v.clone() for k, v in sd.items()}
    pkeys = [k for k, v in osd.items() if v.is_floating_point() and 'running_' not in k]
    for k in pkeys:
        osd[k].requires_grad_(True)
    fo = {k: v.clone().requires_grad_(True) for k, v in feats_cpu.items()}
    _, olosses = oh.detect_forward(fo, osd, dict(MICRO, bbox_loss_weighting='cls-w**2', ignore_bg_k=0.2), labels=targets.clone(), training=True)
    olosses['loss'].backward()
    _, plain = oh.detect_forward({k: v.clone() for k, v in feats_cpu.items()}, {k: v.detach() for k, v in osd.items()}, MICRO,
                                 labels=targets.clone(), training=True)
    assert abs(float(plain['loss']) - float(olosses['loss'])) > 1e-3          # the options do change this batch's loss
    for k in ('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss', 'num_fg'):
        close(losses[k], float(olosses[k]), rtol=5e-5, what=k)
    params = dict(det.named_parameters())
    for k in pkeys:
        if osd[k].grad is not None:
            close(params[k].grad, osd[k].grad, rtol=2e-3, atol=2e-5, what='grad ' + k)
    for k in fo:
        close(fg[k].grad, fo[k].grad, rtol=2e-3, atol=2e-5, what=f'grad feature {k}')


def test_head_grouped_launches_equal_per_layer(gpu, manifest, monkeypatch):
    """The grouped launches of the head towers (functions.base_conv_group: six convs / BatchNorm layers of equal depth per launch) against the
    same layers evaluated one by one: predictions, losses and BatchNorm buffers bit-identical, gradients equal to the order of their fp32
    atomics; repeated to catch a race between the problems of a launch."""
    from leod_amd import functions as Fn

    def rnd(shape, seed):
        return torch.randn(shape, generator=torch.Generator().manual_seed(seed))

    feats_cpu = {2: rnd((3, 32, 8, 12), 51), 3: rnd((3, 64, 4, 6), 52), 4: rnd((3, 128, 2, 3), 53)}
    targets = op.batched_yolox_labels(micro_labels(3, seed=7))
    grouped_fn = Fn.base_conv_group

    def run(grouped):
        monkeypatch.setattr(Fn, 'base_conv_group', grouped_fn if grouped else (lambda mods, xs: [m.forward_nhwc(x) for m, x in zip(mods, xs)]))
        det, _, _ = build(manifest, 'micro', 5, 'small', micro=True)
        det.train()
        fg = {k: v.to(DEV).requires_grad_(True) for k, v in feats_cpu.items()}
        pred, losses = det.forward_detect(fg, targets=targets.to(DEV))
        losses['loss'].backward()
        torch.cuda.synchronize()
        out = {'pred': pred.detach().clone(), 'loss': losses['loss'].detach().clone()}
        out.update({'g.' + k: v.grad.clone() for k, v in det.named_parameters() if v.grad is not None})
        out.update({f'gf.{k}': v.grad.clone() for k, v in fg.items()})
        out.update({'b.' + k: v.clone() for k, v in det.state_dict().items() if 'running_' in k})
        return out

    ref = run(False)
    for _ in range(3):
        got = run(True)
        assert got.keys() == ref.keys()
        for k in ref:
            if k.startswith(('g.', 'gf.')):      # weight gradients accumulate with fp32 atomics (order varies from run to run in either mode)
                torch.testing.assert_close(got[k], ref[k], rtol=2e-5, atol=1e-6, msg=k)
            else:
                assert torch.equal(got[k], ref[k]), k


def test_backbone_backward_vs_oracle(gpu, manifest):
    """Gradients of a 3-step unrolled micro backbone (states carried, features of every stage used)."""
    det, sd, _ = build(manifest, 'micro', 5, 'small', micro=True, train=True)
    ev = synth_events(3, 2, 20, 60, 90, seed=4, as_uint8=True)
    ws = {s: torch.randn((2, 16 * 2 ** (s - 1), 64 // 2 ** (s + 1), 96 // 2 ** (s + 1)),
                         generator=torch.Generator().manual_seed(90 + s)) for s in (1, 2, 3, 4)}
    states, loss = None, 0.
    for t in range(3):
        feats, states = det.forward_backbone(ev[t].to(DEV), states)
        for s, v in feats.items():
            loss = loss + (v * ws[s].to(DEV)).sum() * (0.5 + 0.25 * t)
    loss = loss + (states[2][1] ** 2).sum()
    loss.backward()
    osd = {k: v.clone() for k, v in sd.items()}
    bkeys = [k for k in osd if k.startswith('backbone.')]
    for k in bkeys:
        osd[k].requires_grad_(True)
    evp = ob.pad_ev_repr(ev.float(), (64, 96))
    states, oloss = None, 0.
    for t in range(3):
        feats, states = ob.backbone_forward(evp[t], states, osd, MICRO)
        for s, v in feats.items():
            oloss = oloss + (v * ws[s]).sum() * (0.5 + 0.25 * t)
    oloss = oloss + (states[2][1] ** 2).sum()
    oloss.backward()
    close(loss, oloss, rtol=1e-4)
    params = dict(det.named_parameters())
    for k in bkeys:
        close(params[k].grad, osd[k].grad, rtol=2e-3, atol=1e-5, what='grad ' + k)


def test_small_real_geometry_fwd_bwd(gpu, manifest):
    """RVT-small at the real Gen1 geometry (256x320, partition 8x10, dim_head 24), 2 timesteps, bs 1:
    forward features and a sample of gradients vs the oracle."""
    det, sd, _ = build(manifest, 'small_gen1', 3, 'small', train=True)
    cfg = ot.model_cfg(48, 24, 0.33, (8, 10))
    ev = synth_events(2, 1, 20, 240, 304, seed=9, as_uint8=True)
    states, loss = None, 0.
    for t in range(2):
        feats, states = det.forward_backbone(ev[t].to(DEV), states)
    loss = sum((v ** 2).mean() for v in feats.values())
    loss.backward()
    osd = {k: v.clone() for k, v in sd.items()}
    bkeys = [k for k in osd if k.startswith('backbone.')]
    for k in bkeys:
        osd[k].requires_grad_(True)
    evp = ob.pad_ev_repr(ev.float(), (256, 320))
    states = None
    for t in range(2):
        ofeats, states = ob.backbone_forward(evp[t], states, osd, cfg)
    oloss = sum((v ** 2).mean() for v in ofeats.values())
    oloss.backward()
    for s in ofeats:
        close(feats[s], ofeats[s], rtol=2e-4, atol=2e-5, what=f'stage {s}')
    close(loss, oloss, rtol=1e-4)
    params = dict(det.named_parameters())
    for k in bkeys:
        close(params[k].grad, osd[k].grad, rtol=3e-3, atol=1e-6, what='grad ' + k)


def _build_gen4(size, full_res, seed, train=True):
    """Gen4 / 1 Mpx detector: ds2 (360x640 -> 384x640, partition 6x10) or full resolution (720x1280 -> 768x1280,
    partition 12x20 = 240-token partitions, BASELINE configs[3]); weights synthesised from the module's own key -> shape map
    (the relative-position tables depend on the partition size)."""
    from leod_amd.config import full_config, dynamically_modify_train_config
    from leod_amd.models.detection.yolox_extension.models.detector import YoloXDetector
    over = dict(dataset=dict(downsample_by_factor_2=not full_res))
    cfg = dynamically_modify_train_config(full_config('gen4', size, overrides=over))
    det = YoloXDetector(cfg.model)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in det.state_dict().items()}, seed)
    det.load_state_dict(sd, strict=True)
    det.to(DEV)
    det.train(train)
    return det, sd, cfg


@pytest.mark.parametrize('size,full_res,T', [('base', False, 2), ('base', True, 2), ('small', True, 1), ('tiny', False, 2)])
def test_gen4_geometries_fwd_bwd(gpu, size, full_res, T):
    """BASELINE configs[3] geometry as a parity case: RVT-base / RVT-small on Gen4 frames, downsampled (384x640, 60-token
    partitions) and at the full 1 Mpx resolution (768x1280, 240-token partitions, stage-1 map 192x320), bs 1, carried LSTM
    states: features of every stage and every backbone parameter gradient vs the oracle."""
    det, sd, cfg = _build_gen4(size, full_res, 21)
    in_hw = tuple(cfg.model.backbone.in_res_hw)
    part = tuple(cfg.model.backbone.stage.attention.partition_size)
    hw = (720, 1280) if full_res else (360, 640)
    assert in_hw == ((768, 1280) if full_res else (384, 640)) and part == ((12, 20) if full_res else (6, 10))
    E, dh = {'base': (64, 32), 'small': (48, 24), 'tiny': (32, 32)}[size]      # tiny: ONE head in stage 1 (padded 6 x 10 partitions)
    ocfg = ot.model_cfg(E, dh, 0.67 if size == 'base' else 0.33, part, num_classes=3, in_res_hw=in_hw)
    ev = synth_events(T, 1, 20, hw[0], hw[1], seed=31, as_uint8=True)
    states = None
    for t in range(T):
        feats, states = det.forward_backbone(ev[t].to(DEV), states)
    loss = sum((v ** 2).mean() for v in feats.values())
    loss.backward()
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
    for s in ofeats:
        assert feats[s].shape == ofeats[s].shape
        close(feats[s], ofeats[s], rtol=2e-4, atol=2e-5, what=f'stage {s}')
    for (h, c), (oh_, oc) in zip(states, ostates):
        close(h, oh_, rtol=2e-4, atol=2e-5, what='h')
        close(c, oc, rtol=2e-4, atol=2e-5, what='c')
    close(loss, oloss, rtol=1e-4)
    params = dict(det.named_parameters())
    for k in bkeys:
        close(params[k].grad, osd[k].grad, rtol=3e-3, atol=1e-6, what='grad ' + k)


def test_1mpx_training_step_and_nms_vs_oracle(gpu):
    """BASELINE configs[3] past the backbone: RVT-base on 720x1280 frames (768x1280 padded, 20160 anchors, 3 classes), T = 2, bs = 1,
    both frames labelled -- ONE full training step (backbone, PAFPN, head, SimOTA, losses, backward, clip + AdamW) against
    ``OracleTrainer.step`` (modules/detection.py:188-298, yolo_head.py:403-597): six losses to 2e-5, identical SimOTA foreground
    count; then the pseudo-label pass (head eval + postprocess / batched NMS, boxes.py:32-86) on the same frames: keep counts equal,
    kept boxes to 2e-4."""
    from oracle import postproc as op
    from oracle.synth import synth_labels
    from leod_amd.engine import TrainEngine, PseudoLabelEngine
    det, sd, cfg = _build_gen4('base', True, 23)
    in_hw = tuple(cfg.model.backbone.in_res_hw)
    part = tuple(cfg.model.backbone.stage.attention.partition_size)
    ocfg = ot.model_cfg(64, 32, 0.67, part, num_classes=3, in_res_hw=in_hw)
    T, B, hw = 2, 1, (720, 1280)
    ev = synth_events(T, B, 20, hw[0], hw[1], seed=41, as_uint8=True)
    labs = [synth_labels(B, hw, 3, seed=50 + t, max_boxes=6) for t in range(T)]
    labels = [[labs[t][b] for b in range(B)] for t in range(T)]
    first = torch.ones(B, dtype=torch.bool)
    otr = ot.OracleTrainer(sd, ocfg, total_steps=1000)
    ref, _ = otr.step(ev, labels, first)
    eng = TrainEngine(det, total_steps=1000)
    flat = [labs[t][b] for t in range(T) for b in range(B)]
    got = eng.step(ev.to(DEV), op.batched_yolox_labels(flat).to(DEV), [list(range(B)) for _ in range(T)], first.to(DEV))
    keys = ('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss', 'num_fg')
    g = {k: float(got[k]) for k in keys}
    assert g['num_fg'] == pytest.approx(ref['num_fg'], rel=1e-6), 'SimOTA foreground count differs'
    for k in keys[:5]:
        assert g[k] == pytest.approx(ref[k], rel=2e-5, abs=1e-6), (k, g[k], ref[k])
    # inference + NMS on the 20160-anchor head (weights reloaded: the step above moved them)
    det.load_state_dict(sd)
    pl = PseudoLabelEngine(det, 3, conf_thre=0.01, obj_thresh=[0.05] * 3, cls_thresh=[0.05] * 3, hflip=False, max_det=8192,
                           dataset_name='gen4', downsampled_by_2=False)
    _, _, dets, cnt = pl.step(ev.to(DEV))
    rdets, _, _ = ot.infer_sequence(sd, ocfg, ev, conf_thre=0.01, hflip=False)
    assert [int(c) for c in cnt.cpu()] == [len(r) for r in rdets], 'NMS keep counts differ from the oracle'
    for i, r in enumerate(rdets):
        # 4800 kept boxes from random weights carry many near-equal scores: kept SETS are compared (every oracle box has exactly one
        # partner within 2e-3 in all 7 columns), the score order up to swaps of neighbours whose scores agree to 1e-6
        a, b = dets[i, :len(r)].cpu().numpy().astype(np.float64), r.numpy().astype(np.float64)
        d = np.abs(a[None, :, :] - b[:, None, :]).max(-1)                  # [ref, got]
        j = d.argmin(1)
        assert d[np.arange(len(b)), j].max() < 2e-3 and len(set(j.tolist())) == len(b)
        sc_a, sc_b = a[:, 4] * a[:, 5], b[:, 4] * b[:, 5]
        assert np.all(np.diff(sc_a) <= 1e-6) and np.abs(sc_a - sc_b).max() < 1e-5


def test_yolox_blocks_reference_signatures(gpu):
    """``BaseConv`` / ``Bottleneck`` / ``CSPLayer`` called the reference's way (NCHW tensor in, NCHW tensor out, network_blocks.py:55-166) equal a
    plain torch restatement built from their own parameters (eval mode: folded BatchNorm + SiLU)."""
    import torch.nn.functional as F
    from leod_amd.models.detection.yolox.models.network_blocks import BaseConv, Bottleneck, CSPLayer
    torch.manual_seed(3)

    def ref_conv(m, x):
        y = F.conv2d(x, m.conv.weight, None, m.stride, (m.conv.kernel_size[0] - 1) // 2)
        y = F.batch_norm(y, m.bn.running_mean, m.bn.running_var, m.bn.weight, m.bn.bias, False, 0., m.bn.eps)
        return F.silu(y)

    def randomize(mod):
        for m in mod.modules():
            if isinstance(m, BaseConv):
                m.bn.running_mean.normal_(0, 0.2)
                m.bn.running_var.uniform_(0.5, 1.5)
                m.bn.weight.data.uniform_(0.5, 1.5)
                m.bn.bias.data.normal_(0, 0.2)
    x = torch.randn(2, 32, 8, 12, device=DEV)
    conv = BaseConv(32, 64, 3, 1).to(DEV).eval()
    bott = Bottleneck(32, 32, shortcut=True, expansion=1.0).to(DEV).eval()
    csp = CSPLayer(32, 64, n=1, shortcut=False).to(DEV).eval()
    for m in (conv, bott, csp):
        randomize(m)
    with torch.no_grad():
        close(conv(x), ref_conv(conv, x), rtol=2e-4, atol=2e-5)
        close(bott(x), x + ref_conv(bott.conv2, ref_conv(bott.conv1, x)), rtol=2e-4, atol=2e-5)
        b = csp.m[0]
        x1 = ref_conv(b.conv2, ref_conv(b.conv1, ref_conv(csp.conv1, x)))
        close(csp(x), ref_conv(csp.conv3, torch.cat([x1, ref_conv(csp.conv2, x)], 1)), rtol=2e-4, atol=2e-5)
    assert tuple(bott(x).shape) == (2, 32, 8, 12) and tuple(csp(x).shape) == (2, 64, 8, 12)


# ---- the depthwise variants (off in every shipped config; north_star names the depthwise stems) --------------------------------
@pytest.mark.parametrize('B,H,W,C,ks,stride', [(2, 8, 12, 32, 3, 1), (3, 16, 20, 96, 3, 2), (2, 8, 10, 16, 5, 1), (1, 9, 7, 48, 7, 2),
                                               (32, 32, 40, 96, 3, 1)])
def test_depthwise_conv_kernels_vs_torch(gpu, B, H, W, C, ks, stride):
    """leod_dwconv_nhwc_{fwd,dgrad,wgrad} (nn.Conv2d(C, C, ks, stride, groups=C): DWConv.dconv network_blocks.py:61-68, conv3x3_dws
    rnn.py:26-30) against torch's fp32 grouped convolution: plain + bias, training-BatchNorm statistics, eval BatchNorm + SiLU, input
    gradient (also accumulating), weight and bias gradient."""
    import torch.nn.functional as F
    from leod_amd import ops

    def rnd(shape, seed, s=1.0):
        return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * s

    pad = (ks - 1) // 2
    x = rnd((B, C, H, W), 1).requires_grad_(True)
    w = rnd((C, 1, ks, ks), 2, 0.3).requires_grad_(True)
    b = rnd((C,), 3, 0.2).requires_grad_(True)
    ref = F.conv2d(x, w, b, stride=stride, padding=pad, groups=C)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
    wd, bd = w.detach().to(DEV), b.detach().to(DEV)
    y = ops.conv_nhwc_fwd(xd, wd, bd, stride=stride)
    close(y, ref.permute(0, 2, 3, 1), rtol=2e-5, atol=2e-6, what='depthwise forward')
    # training BatchNorm statistics out of the same launch (4 replicas of the accumulators)
    cs = torch.zeros((4, 2, C), dtype=torch.float64, device=DEV)
    y2 = ops.conv_nhwc_fwd(xd, wd, None, stride=stride, colstats=cs)
    r0 = F.conv2d(x, w, None, stride=stride, padding=pad, groups=C).detach().permute(0, 2, 3, 1).reshape(-1, C).double()
    close(y2.reshape(-1, C), r0.float(), rtol=2e-5, atol=2e-6)
    close(cs.sum(0)[0], r0.sum(0), rtol=1e-5, atol=1e-4)
    close(cs.sum(0)[1], (r0 * r0).sum(0), rtol=1e-5, atol=1e-4)
    # eval BatchNorm folded + SiLU
    bw, bb, rm, rv = 0.5 + rnd((C,), 4).abs(), rnd((C,), 5, 0.3), rnd((C,), 6, 0.2), 0.5 + rnd((C,), 7).abs()
    y3 = ops.conv_nhwc_fwd(xd, wd, None, stride=stride, bn=tuple(t.to(DEV) for t in (bw, bb, rm, rv)))
    r3 = F.silu(F.batch_norm(F.conv2d(x, w, None, stride=stride, padding=pad, groups=C), rm, rv, bw, bb, training=False, eps=1e-5))
    close(y3, r3.detach().permute(0, 2, 3, 1), rtol=5e-5, atol=5e-6, what='depthwise + eval BatchNorm + SiLU')
    # gradients
    dy = rnd(tuple(ref.shape), 8)
    ref.backward(dy)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    dx = ops.conv_nhwc_dgrad(dyd, wd, xd.shape, stride=stride)
    close(dx, x.grad.permute(0, 2, 3, 1), rtol=5e-5, atol=5e-6, what='depthwise dgrad')
    dx2 = ops.conv_nhwc_dgrad(dyd, wd, xd.shape, stride=stride, out=dx.clone(), accumulate=True)
    close(dx2, 2 * x.grad.permute(0, 2, 3, 1), rtol=5e-5, atol=1e-5, what='depthwise dgrad accumulates')
    dw, db = torch.zeros_like(wd), torch.zeros_like(bd)
    ops.conv_nhwc_wgrad(dyd, xd, dw, db, stride=stride)
    close(dw, w.grad, rtol=2e-4, atol=2e-4 * float(w.grad.abs().max()), what='depthwise wgrad')
    close(db, b.grad, rtol=2e-4, atol=2e-4 * float(b.grad.abs().max()), what='depthwise bias gradient')


@pytest.mark.parametrize('tag,only_hidden,ks', [('h', True, 3), ('xh', False, 3), ('h5', True, 5)])
@pytest.mark.parametrize('seq', [False, True])
def test_depthwise_convlstm_golden(gpu, golden_dir, tag, only_hidden, ks, seq):
    """``DWSConvLSTM2d(dws_conv=True)`` on the HIP path against the REFERENCE's own three chained timesteps (g22): outputs and every
    gradient, through ``forward`` per timestep and through the time-batched ``forward_sequence``."""
    import json
    from leod_amd.models.layers.rnn import DWSConvLSTM2d
    g = np.load(os.path.join(golden_dir, 'g22_depthwise.npz'))

    def rnd(shape, seed, s=1.0):
        return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * s

    man = json.loads(str(g[f'lstm_{tag}_manifest']))
    for start in ('none', 'state'):
        m = DWSConvLSTM2d(dim=16, dws_conv=True, dws_conv_only_hidden=only_hidden, dws_conv_kernel_size=ks)
        assert {k: list(v.shape) for k, v in m.state_dict().items()} == man
        m.load_state_dict(synth_state_dict(man, 21))
        m.to(DEV)
        xs = [rnd((2, 16, 8, 10), 220 + t).to(DEV).requires_grad_(True) for t in range(3)]
        h0, c0 = rnd((2, 16, 8, 10), 230, 0.5).to(DEV).requires_grad_(True), rnd((2, 16, 8, 10), 231, 0.5).to(DEV).requires_grad_(True)
        hc = None if start == 'none' else (h0, c0)
        if seq:
            xcat = torch.cat(xs, 0)                       # [T*B, C, H, W]
            hall, hc = m.forward_sequence(xcat, 3, hc)
            hs = list(hall.reshape(3, 2, 16, 8, 10))
        else:
            hs = []
            for x in xs:
                hc = m(x, hc)
                hs.append(hc[0])
        loss = sum((h * rnd(tuple(h.shape), 240 + i).to(DEV)).sum() for i, h in enumerate(hs)) + (hc[1] * rnd(tuple(hc[1].shape), 250).to(DEV)).sum()
        loss.backward()
        pre = f'lstm_{tag}_{start}_'
        close(torch.stack(hs), g[pre + 'h'], rtol=2e-5, what='h')
        close(hc[1], g[pre + 'c'], rtol=2e-5, what='c')
        close(torch.stack([x.grad for x in xs]), g[pre + 'dx'], rtol=2e-4, atol=2e-6, what='dx')
        if start == 'state':
            close(h0.grad, g[pre + 'dh0'], rtol=2e-4, atol=2e-6, what='dh0')
            close(c0.grad, g[pre + 'dc0'], rtol=2e-4, atol=2e-6, what='dc0')
        for n, p in m.named_parameters():
            ref = g[pre + 'grad_' + n.replace('.', '_')]
            close(p.grad, ref, rtol=5e-4, atol=5e-4 * float(np.abs(ref).max()), what='grad ' + n)


def test_depthwise_head_golden(gpu, golden_dir, manifest):
    """PAFPN + head built with ``depthwise=True`` (DWConv in the Bottlenecks, the bottom-up convs and the head towers: yolo_pafpn.py:37,
    yolo_head.py:52, network_blocks.py:57-76) against the REFERENCE (g22): state-dict manifest, eval predictions, training losses, BatchNorm
    buffers, the gradients of the input features, of every depthwise filter / BatchNorm and the norm of every parameter gradient."""
    import json
    from leod_amd.config import full_config, dynamically_modify_train_config
    from leod_amd.models.detection.yolox_extension.models.detector import YoloXDetector
    g = np.load(os.path.join(golden_dir, 'g22_depthwise.npz'))
    over = dict(model=dict(backbone=dict(embed_dim=16, stage=dict(attention=dict(dim_head=8))), fpn=dict(depth=0.33, depthwise=True),
                           head=dict(depthwise=True)))
    cfg = dynamically_modify_train_config(full_config('gen1', 'small', overrides=over))
    cfg.model.backbone.in_res_hw = (64, 96)
    cfg.model.backbone.stage.attention.partition_size = (2, 3)
    det = YoloXDetector(cfg.model)
    man = json.loads(str(g['det_manifest']))
    assert {k: list(v.shape) for k, v in det.state_dict().items()} == man
    det.load_state_dict(synth_state_dict(man, 22), strict=True)
    det.to(DEV).eval()

    def rnd(shape, seed):
        return torch.randn(shape, generator=torch.Generator().manual_seed(seed))

    feats_cpu = {2: rnd((3, 32, 8, 12), 51), 3: rnd((3, 64, 4, 6), 52), 4: rnd((3, 128, 2, 3), 53)}
    with torch.no_grad():
        pred, losses = det.forward_detect({k: v.to(DEV) for k, v in feats_cpu.items()})
    assert losses is None
    close(pred, g['pred_eval'], what='eval predictions')
    targets = op.batched_yolox_labels(micro_labels(3, seed=7))
    det.train()
    fg = {k: v.to(DEV).requires_grad_(True) for k, v in feats_cpu.items()}
    pred, losses = det.forward_detect(fg, targets=targets.to(DEV))
    close(pred, g['pred_train'], what='train predictions')
    for k in ('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss', 'num_fg'):
        close(losses[k], g['loss_' + k], rtol=5e-5, what=k)
    dsd = det.state_dict()
    for k in ['fpn.bu_conv2.dconv.bn.running_mean', 'fpn.bu_conv2.pconv.bn.running_var', 'yolox_head.cls_convs.2.1.dconv.bn.running_var']:
        close(dsd[k], g['bn_' + k.replace('.', '_')], rtol=5e-5, atol=1e-6)
    losses['loss'].backward()
    for k in fg:
        close(fg[k].grad, g[f'dfeat{k}'], rtol=2e-3, atol=2e-5, what=f'grad feature {k}')
    params = dict(det.named_parameters())
    gk = [str(k) for k in g['grad_keys']]
    for k in gk:
        if '.dconv.' in k:
            close(params[k].grad, g['grad_' + k.replace('.', '_')], rtol=2e-3, atol=2e-5, what='grad ' + k)
    mine = np.array([float(params[k].grad.norm()) for k in gk])
    np.testing.assert_allclose(mine, g['grad_norms'], rtol=2e-3, atol=1e-6)


@pytest.mark.parametrize('tag,factor,cin,cout,overlap,affine', [('f4_patch', 4, 20, 16, False, True), ('f4_noaffine', 4, 20, 16, True, False),
                                                                ('f2_patch_noaffine', 2, 16, 32, False, False), ('f2_patch', 2, 16, 32, False, True),
                                                                ('f4_patch_noaffine', 4, 20, 16, False, False)])
@pytest.mark.parametrize('mode', ['f32', '16f'])
def test_downsample_options_golden(gpu, golden_dir, tag, factor, cin, cout, overlap, affine, mode):
    """``ConvDownsampling_Cf2Cl`` with non-overlapping patches (``overlap=False``) and / or a LayerNorm without affine parameters
    (``norm_affine=False``; maxvit.py:160-172, defaults in every shipped config) against the REFERENCE (g23): state-dict keys, output, input
    gradient (stride-2 layers; the stem has no differentiable input) and parameter gradients.  Mode 16f at the 16-bit operand tolerance."""
    import json
    from leod_amd import ops
    from leod_amd.config import create
    from leod_amd.models.layers.maxvit.maxvit import ConvDownsampling_Cf2Cl
    g = np.load(os.path.join(golden_dir, 'g23_downsample_options.npz'))
    man = json.loads(str(g[tag + '_manifest']))
    prev = ops.set_precision(mode)
    try:
        m = ConvDownsampling_Cf2Cl(cin, cout, factor, create(dict(type='patch', overlap=overlap, norm_affine=affine)))
        assert {k: list(v.shape) for k, v in m.state_dict().items()} == man
        m.load_state_dict(synth_state_dict(man, 23))
        m.to(DEV)
        x = torch.randn((2, cin, 16, 24), generator=torch.Generator().manual_seed(230 + factor)).to(DEV).requires_grad_(factor != 4)
        y = m(x)
        r = torch.randn(tuple(y.shape), generator=torch.Generator().manual_seed(239)).to(DEV)
        (y * r).sum().backward()
        tol = dict(rtol=2e-5, atol=2e-6) if mode == 'f32' else dict(rtol=3e-2, atol=3e-2)
        gt = dict(rtol=2e-4, atol=2e-5) if mode == 'f32' else dict(rtol=5e-2, atol=5e-2 * float(np.abs(g[tag + '_dx']).max()))
        close(y, g[tag + '_y'], what='y', **tol)
        if factor != 4:
            close(x.grad, g[tag + '_dx'], what='dx', **gt)
        for n, p in m.named_parameters():
            ref = g[tag + '_grad_' + n.replace('.', '_')]
            close(p.grad, ref, rtol=gt['rtol'], atol=(2e-4 if mode == 'f32' else 5e-2) * float(np.abs(ref).max()), what='grad ' + n)
    finally:
        ops.set_precision(prev)


BLOCK_OPTION_CASES = [
    ('geglu', True, False, dict(mlp_gated=True, mlp_activation='gelu')),
    ('swiglu_grid', False, False, dict(mlp_gated=True, mlp_activation='swish')),
    ('reglu_skipnorm', True, True, dict(mlp_gated=True, mlp_activation='relu')),
    ('glu_sigmoid_mha', True, False, dict(mlp_gated=True, mlp_activation='sigmoid', use_torch_mha=True)),
    ('mha_grid', False, False, dict(use_torch_mha=True)),
    ('relu_plain', True, False, dict(mlp_activation='relu')),
    ('mish_nols_nobias', False, False, dict(mlp_activation='mish', ls_init_value=0.0, attention_bias=False, mlp_bias=False)),
    ('hswish_tanhglu', True, False, dict(mlp_activation='hard_swish')),
    ('tanh_glu_nols', True, False, dict(mlp_gated=True, mlp_activation='tanh', ls_init_value=0.0)),
    ('elu', False, False, dict(mlp_activation='elu')), ('selu', True, False, dict(mlp_activation='selu')),
    ('hsig_glu', True, False, dict(mlp_gated=True, mlp_activation='hard_sigmoid')), ('relu6', True, False, dict(mlp_activation='relu6')),
    ('leaky', True, False, dict(mlp_activation='leaky_relu')), ('celu', True, False, dict(mlp_activation='celu')),
    ('hmish', True, False, dict(mlp_activation='hard_mish')), ('silu', True, False, dict(mlp_activation='silu')),
    ('mha_nobias', True, False, dict(use_torch_mha=True, attention_bias=False)),
]


def _block_cfg(**over):
    from leod_amd.config import create
    base = dict(use_torch_mha=False, partition_size=(8, 10), dim_head=16, attention_bias=True, mlp_activation='gelu', mlp_gated=False,
                mlp_bias=True, mlp_ratio=4, drop_mlp=0, drop_path=0, ls_init_value=1e-5)
    base.update(over)
    return create(base)


@pytest.mark.parametrize('tag,window,skip,over', BLOCK_OPTION_CASES)
def test_attention_block_options_golden(gpu, golden_dir, tag, window, skip, over):
    """``PartitionAttentionCl`` with the options no shipped config enables -- gated MLP (GLU), every `mlp_activation` name but prelu,
    ``use_torch_mha`` (nn.MultiheadAttention's parameter layout), ``ls_init_value = 0``, ``attention_bias`` / ``mlp_bias`` off
    (maxvit.py:56-118,185-270,307-325) -- against the REFERENCE (g24): state-dict manifest, output, input gradient, every parameter gradient."""
    import json
    from leod_amd.models.layers.maxvit.maxvit import PartitionAttentionCl, PartitionType
    g = np.load(os.path.join(golden_dir, 'g24_block_options.npz'))
    man = json.loads(str(g[tag + '_manifest']))
    m = PartitionAttentionCl(32, PartitionType.WINDOW if window else PartitionType.GRID, _block_cfg(**over), skip_first_norm=skip)
    assert m.generic and {k: list(v.shape) for k, v in m.state_dict().items()} == man
    m.load_state_dict(synth_state_dict(man, 24))
    m.to(DEV)
    x = torch.randn((1, 16, 20, 32), generator=torch.Generator().manual_seed(241)).to(DEV).requires_grad_(True)
    y = m(x)
    r = torch.randn(tuple(y.shape), generator=torch.Generator().manual_seed(242)).to(DEV)
    (y * r).sum().backward()
    close(y, g[tag + '_y'], rtol=2e-5, atol=2e-6, what='y')
    close(x.grad, g[tag + '_dx'], rtol=5e-4, atol=5e-5 * float(np.abs(g[tag + '_dx']).max()), what='dx')      # (fp32 summation order: a few 1e-5 of the largest element)
    for n, p in m.named_parameters():
        ref = g[tag + '_grad_' + n.replace('.', '_')]
        close(p.grad, ref, rtol=5e-4, atol=5e-5 * max(1.0, float(np.abs(ref).max())), what='grad ' + n)


def test_token_masking_golden(gpu, golden_dir):
    """``RNNDetectorStage(enable_token_masking=True)`` (maxvit_rnn.py:174-192: ``x[token_mask] = mask_token`` behind the downsample layer)
    against the REFERENCE (g24): two timesteps with carried state through ``forward`` and through ``forward_sequence``, every gradient."""
    import json
    from leod_amd.config import full_config, dynamically_modify_train_config
    from leod_amd.models.detection.recurrent_backbone.maxvit_rnn import RNNDetectorStage
    g = np.load(os.path.join(golden_dir, 'g24_block_options.npz'))
    man = json.loads(str(g['mask_manifest']))
    over = dict(model=dict(backbone=dict(embed_dim=16, stage=dict(attention=dict(dim_head=8)))))
    cfg = dynamically_modify_train_config(full_config('gen1', 'small', overrides=over))
    cfg.model.backbone.stage.attention.partition_size = (2, 3)
    masks = torch.from_numpy(g['mask_masks']).to(DEV)

    def rnd(shape, seed):
        return torch.randn(shape, generator=torch.Generator().manual_seed(seed))

    for seq in (False, True):
        st = RNNDetectorStage(dim_in=20, stage_dim=16, spatial_downsample_factor=4, num_blocks=1, enable_token_masking=True,
                              T_max_chrono_init=4, stage_cfg=cfg.model.backbone.stage)
        assert {k: list(v.shape) for k, v in st.state_dict().items()} == man
        st.load_state_dict(synth_state_dict(man, 25))
        st.to(DEV)
        xs = [rnd((2, 20, 64, 96), 251 + t).to(DEV) for t in range(2)]
        if seq:
            hall, hc = st.forward_sequence(torch.cat(xs, 0), 2, None, token_mask=masks.reshape(4, 16, 24))
            hs = list(hall.reshape(2, 2, 16, 16, 24))
        else:
            hc, hs = None, []
            for x, mk in zip(xs, masks):
                h, hc = st(x, hc, mk)
                hs.append(h)
        (sum((h * rnd(tuple(h.shape), 258 + i).to(DEV)).sum() for i, h in enumerate(hs)) + (hc[1] * rnd(tuple(hc[1].shape), 260).to(DEV)).sum()).backward()
        close(torch.stack(hs), g['mask_h'], rtol=2e-5, atol=2e-6, what='h')
        close(hc[1], g['mask_c'], rtol=2e-5, atol=2e-6, what='c')
        for n, p in st.named_parameters():
            ref = g['mask_grad_' + n.replace('.', '_')]
            close(p.grad, ref, rtol=1e-3, atol=1e-4 * max(1.0, float(np.abs(ref).max())), what='grad ' + n)
