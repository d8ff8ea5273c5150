#!/usr/bin/env python
"""Record golden vectors by importing and running the REFERENCE (/root/reference) on CPU.

Runs only in the build container (the reference never travels to the GPU box); its outputs -- small
.npz/.json fixtures next to this script -- are committed and are what pins ``oracle/`` to the
reference (tests/test_oracle_golden.py).  Third-party packages the reference imports but that are
absent here are served by the stand-ins in ``ref_stubs/`` (not reference source; see its README).

Weights are ``oracle.synth.synth_state_dict(manifest, seed)`` -- a pure function of key/shape/seed --
loaded into the reference model with ``load_state_dict``; inputs are seeded torch/numpy draws that
the tests regenerate.  So fixtures store only the *expected outputs* (plus tiny inputs where that is
simpler).

Usage:  python tests/golden/make_golden.py            # rewrites tests/golden/*.npz, *.json
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(HERE, 'ref_stubs'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.cuda.get_device_name = lambda *a, **k: 'none'  # called at import time by the reference's coco_eval
torch.set_num_threads(8)

from omegaconf import DictConfig  # noqa: E402  (stand-in)
from oracle.synth import synth_state_dict, synth_tensor, synth_events, synth_labels  # noqa: E402

# ---- reference imports ---------------------------------------------------------------------------
from models.layers.rnn import DWSConvLSTM2d  # noqa: E402
from models.layers.maxvit import maxvit as ref_maxvit  # noqa: E402
from models.detection.yolox_extension.models.detector import YoloXDetector  # noqa: E402
from models.detection.yolox.utils.boxes import postprocess as ref_postprocess, bboxes_iou  # noqa: E402
from modules.utils import ssod as ref_ssod  # noqa: E402
from modules.utils.tta import tta_postprocess as ref_tta_postprocess  # noqa: E402
from modules.utils.detection import RNNStates, BackboneFeatureSelector  # noqa: E402
from data.genx_utils.labels import ObjectLabels  # noqa: E402
from data.utils.representations import StackedHistogram, MixedDensityEventStack  # noqa: E402
from utils.padding import InputPadderFromShape  # noqa: E402


def make_cfg(embed_dim, dim_head, fpn_depth, in_hw, part, num_classes=2, **head_kw):
    head = dict(name='YoloX', compile=dict(enable=False, args=dict(mode='reduce-overhead')), depthwise=False,
                act='silu', obj_focal_loss=False, bbox_loss_weighting='', ignore_bbox_thresh=None,
                ignore_label=1024, ignore_bg_k=0, num_classes=num_classes)
    head.update(head_kw)
    return DictConfig(dict(
        backbone=dict(name='MaxViTRNN', compile=dict(enable=False, args=dict(mode='reduce-overhead')),
                      input_channels=20, enable_masking=False, partition_split_32=1, embed_dim=embed_dim,
                      dim_multiplier=[1, 2, 4, 8], num_blocks=[1, 1, 1, 1], T_max_chrono_init=[4, 8, 16, 32],
                      stem=dict(patch_size=4),
                      stage=dict(downsample=dict(type='patch', overlap=True, norm_affine=True),
                                 attention=dict(use_torch_mha=False, partition_size=part, dim_head=dim_head,
                                                attention_bias=True, mlp_activation='gelu', mlp_gated=False,
                                                mlp_bias=True, mlp_ratio=4, drop_mlp=0, drop_path=0,
                                                ls_init_value=1e-5),
                                 lstm=dict(dws_conv=False, dws_conv_only_hidden=True, dws_conv_kernel_size=3,
                                           drop_cell_update=0)),
                      in_res_hw=in_hw),
        fpn=dict(name='PAFPN', compile=dict(enable=False, args=dict(mode='reduce-overhead')), depth=fpn_depth,
                 in_stages=[2, 3, 4], depthwise=False, act='silu'),
        head=head,
        postprocess=dict(confidence_threshold=0.1, nms_threshold=0.45)))


def manifest_of(module):
    return {k: list(v.shape) for k, v in module.state_dict().items()}


def load_synth(module, seed=0):
    man = manifest_of(module)
    sd = synth_state_dict(man, seed)
    module.load_state_dict(sd, strict=True)
    return man


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print('wrote', name, {k: v.shape for k, v in out.items()})


# ====================================================================================================
def g01_convlstm():
    m = DWSConvLSTM2d(dim=32, dws_conv=False, dws_conv_only_hidden=True, dws_conv_kernel_size=3)
    load_synth(m, 1)
    x = rnd((2, 32, 8, 10), 11)
    h0, c0 = rnd((2, 32, 8, 10), 12, 0.5), rnd((2, 32, 8, 10), 13, 0.5)
    with torch.no_grad():
        h_a, c_a = m(x, None)
        h_b, c_b = m(x, (h0, c0))
    save('g01_convlstm.npz', h_nostate=h_a, c_nostate=c_a, h_state=h_b, c_state=c_b)


def g02_partition():
    x = torch.arange(2 * 16 * 20, dtype=torch.float32).view(2, 16, 20, 1)
    ws = (8, 10)
    wp = ref_maxvit.window_partition(x, ws)
    gp = ref_maxvit.grid_partition(x, ws)
    wr = ref_maxvit.window_reverse(wp, ws, (16, 20))
    gr = ref_maxvit.grid_reverse(gp, ws, (16, 20))
    x2 = torch.arange(1 * 12 * 20, dtype=torch.float32).view(1, 12, 20, 1)   # gen4-like (6,10)
    save('g02_partition.npz', window=wp.long(), grid=gp.long(), window_rev=wr.long(), grid_rev=gr.long(),
         window_6x10=ref_maxvit.window_partition(x2, (6, 10)).long(),
         grid_6x10=ref_maxvit.grid_partition(x2, (6, 10)).long())


def g03_attention():
    cfg = make_cfg(48, 24, 0.33, (256, 320), (8, 10)).backbone.stage.attention
    x = rnd((2, 16, 20, 48), 31)
    out = {}
    for window, skip in [(True, False), (True, True), (False, False)]:
        pt = ref_maxvit.PartitionType.WINDOW if window else ref_maxvit.PartitionType.GRID
        m = ref_maxvit.PartitionAttentionCl(dim=48, partition_type=pt, attention_cfg=cfg, skip_first_norm=skip)
        load_synth(m, 3)
        with torch.no_grad():
            out[f"blk_{'window' if window else 'grid'}_{'skip' if skip else 'norm'}"] = m(x)
    sa = ref_maxvit.SelfAttentionCl(dim=48, dim_head=24, bias=True)
    load_synth(sa, 4)
    with torch.no_grad():
        out['self_attn'] = sa(ref_maxvit.window_partition(x, (8, 10)))
    save('g03_attention.npz', **out)


MICRO = dict(embed_dim=16, dim_head=8, fpn_depth=0.33, in_hw=(64, 96), part=(2, 3))


def g04_backbone():
    det = YoloXDetector(make_cfg(**MICRO))
    load_synth(det, 5)
    det.eval()
    ev = synth_events(3, 2, 20, 60, 90, seed=4, as_uint8=False)
    padder = InputPadderFromShape(desired_hw=(64, 96))
    ev = padder.pad_tensor_ev_repr(ev)
    out, states = {}, None
    with torch.no_grad():
        for t in range(3):
            feats, states = det.forward_backbone(ev[t], states)
            for k, v in feats.items():
                out[f't{t}_s{k}'] = v
        for s, (h, c) in enumerate(states):
            out[f'final_c{s + 1}'] = c
    save('g04_backbone_micro.npz', **out)
    # RVT-tiny at the real Gen1 geometry: one timestep, checksums + slices
    det = YoloXDetector(make_cfg(32, 32, 0.33, (256, 320), (8, 10)))
    load_synth(det, 6)
    det.eval()
    ev = InputPadderFromShape(desired_hw=(256, 320)).pad_tensor_ev_repr(
        synth_events(2, 1, 20, 240, 304, seed=5, as_uint8=False))
    out = {}
    with torch.no_grad():
        feats, states = det.forward_backbone(ev[0], None)
        feats, states = det.forward_backbone(ev[1], states)
    for k, v in feats.items():
        out[f's{k}_mean'] = v.mean()
        out[f's{k}_absmax'] = v.abs().max()
        out[f's{k}_slice'] = v[0, :8, :4, :5]
    save('g04_backbone_tiny256.npz', **out)


def micro_labels(n_frames, seed, hw=(60, 90)):
    labs = synth_labels(n_frames, hw, 2, seed=seed, max_boxes=4)
    for l in labs:   # micro geometry: smaller boxes
        l[:, 3] = l[:, 3].clamp(max=30)
        l[:, 4] = l[:, 4].clamp(max=24)
        l[:, 1] = torch.minimum(l[:, 1], hw[1] - 1 - l[:, 3])
        l[:, 2] = torch.minimum(l[:, 2], hw[0] - 1 - l[:, 4])
    return labs


def g05_head():
    det = YoloXDetector(make_cfg(**MICRO))
    load_synth(det, 5)
    feats = {2: rnd((3, 32, 8, 12), 51), 3: rnd((3, 64, 4, 6), 52), 4: rnd((3, 128, 2, 3), 53)}
    det.eval()
    with torch.no_grad():
        pred_eval, _ = det.forward_detect(feats)
    labs = micro_labels(3, seed=7)
    labs[1] = labs[1][:1]                        # padded rows
    labs[2][0, 1:5] = torch.tensor([0., 0., 12., 9.])   # border GT
    targets = ObjectLabels.get_labels_as_batched_tensor([ObjectLabels(l, (60, 90)) for l in labs])
    det.train()
    for p in det.parameters():
        p.grad = None
    pred_tr, losses = det.forward_detect({k: v.clone().requires_grad_(True) for k, v in feats.items()},
                                         targets=targets.clone())
    losses['loss'].backward()
    out = dict(pred_eval=pred_eval, pred_train=pred_tr, targets=targets,
               **{f'loss_{k}': (v.detach() if torch.is_tensor(v) else v) for k, v in losses.items()})
    sd = det.state_dict()
    for k in ['fpn.lateral_conv0.bn.running_mean', 'fpn.lateral_conv0.bn.running_var',
              'yolox_head.stems.0.bn.running_mean', 'yolox_head.cls_convs.2.1.bn.running_var']:
        out['bn_' + k.replace('.', '_')] = sd[k]
    gn = {n: p.grad.norm() for n, p in det.named_parameters() if p.grad is not None}
    out['grad_keys'] = np.array(sorted(gn.keys()))
    out['grad_norms'] = np.array([float(gn[k]) for k in sorted(gn.keys())], dtype=np.float64)
    save('g05_head_micro.npz', **out)


def g06_simota():
    det = YoloXDetector(make_cfg(32, 32, 0.33, (256, 320), (8, 10)))
    head = det.yolox_head
    # anchors of the Gen1 geometry: 32x40, 16x20, 8x10
    from oracle.head import make_grids
    gx, gy, gs = make_grids([(32, 40), (16, 20), (8, 10)], (8, 16, 32))
    A = gx.numel()
    g = torch.Generator().manual_seed(61)
    out = dict()
    for case in range(3):
        n = [5, 9, 3][case]
        labs = synth_labels(1, (240, 304), 2, seed=60 + case, max_boxes=n, min_boxes=n)[0]
        gt = torch.stack([labs[:, 1] + labs[:, 3] / 2, labs[:, 2] + labs[:, 4] / 2, labs[:, 3], labs[:, 4]], 1)
        cls = labs[:, 5]
        # predictions: anchors' own centres with noisy sizes, some pulled towards the GTs
        bp = torch.stack([(gx + 0.5) * gs + torch.randn(A, generator=g) * 3,
                          (gy + 0.5) * gs + torch.randn(A, generator=g) * 3,
                          gs * (2 + 4 * torch.rand(A, generator=g)), gs * (2 + 3 * torch.rand(A, generator=g))], 1)
        cls_l = torch.randn(A, 2, generator=g) * 2
        obj_l = torch.randn(A, 1, generator=g) * 2
        if case == 1:   # duplicate GT (ties) and identical predictions
            gt[1] = gt[0]
            cls[1] = cls[0]
            bp[100:110] = bp[100]
        fg, geom = head.get_geometry_constraint(gt, gs[None], gx[None], gy[None])
        (mcls, fg_mask, pious, minds, nfg) = head.get_assignments(
            n, gt, cls, bp, gs[None], gx[None], gy[None], cls_l, obj_l)
        out.update({f'c{case}_gt': gt, f'c{case}_cls': cls, f'c{case}_bp': bp, f'c{case}_cls_l': cls_l,
                    f'c{case}_obj_l': obj_l, f'c{case}_geom_fg': fg, f'c{case}_fg_mask': fg_mask,
                    f'c{case}_matched': minds, f'c{case}_mcls': mcls, f'c{case}_pious': pious,
                    f'c{case}_nfg': nfg})
        # ignore variant: mark every third box as ignore
        valid = torch.ones(n, dtype=torch.bool)
        valid[::3] = False
        (mcls, fg_mask, pious, minds, nfg, ign) = head.get_assignments_w_ignore(
            int(valid.sum()), gt, cls, bp, gs[None], gx[None], gy[None], cls_l, obj_l, valid)
        out.update({f'c{case}_valid': valid, f'c{case}_ig_fg_mask': fg_mask, f'c{case}_ig_matched': minds,
                    f'c{case}_ig_mcls': mcls, f'c{case}_ig_pious': pious, f'c{case}_ig_nfg': nfg,
                    f'c{case}_ig_ignore_mask': ign})
    # full loss with ignore labels through get_losses (ignore_label 1024 rows + an all-ignore image)
    labs = synth_labels(4, (240, 304), 2, seed=66, max_boxes=5, min_boxes=2)
    tg = ObjectLabels.get_labels_as_batched_tensor([ObjectLabels(l, (240, 304)) for l in labs])
    tg[0, 1, 0] = 1024
    tg[2, :, 0] = torch.where(tg[2].sum(1) > 0, torch.full_like(tg[2, :, 0], 1024.), tg[2, :, 0])
    tg[3] = 0
    B = 4
    outputs = torch.cat([
        torch.stack([(gx + 0.5) * gs, (gy + 0.5) * gs, gs * 3, gs * 2.5], 1)[None].repeat(B, 1, 1)
        + torch.randn(B, A, 4, generator=g), torch.randn(B, A, 3, generator=g)], -1)
    xs = [gx[None, :1280], gx[None, 1280:1600], gx[None, 1600:]]
    ys = [gy[None, :1280], gy[None, 1280:1600], gy[None, 1600:]]
    ss = [gs[None, :1280], gs[None, 1280:1600], gs[None, 1600:]]
    res = head.get_losses(xs, ys, ss, tg.clone(), outputs.clone(), [], torch.float32)
    out.update(ign_targets=tg, ign_outputs=outputs,
               ign_losses=np.array([float(r) for r in res], dtype=np.float64))
    tg2 = tg.clone()
    tg2[:, :, 0] = torch.where(tg2[:, :, 0] == 1024, torch.zeros_like(tg2[:, :, 0]), tg2[:, :, 0])
    res2 = head.get_losses(xs, ys, ss, tg2.clone(), outputs.clone(), [], torch.float32)
    out.update(noign_losses=np.array([float(r) for r in res2], dtype=np.float64))
    # focal objectness
    head2 = YoloXDetector(make_cfg(32, 32, 0.33, (256, 320), (8, 10), obj_focal_loss=True)).yolox_head
    res3 = head2.get_losses(xs, ys, ss, tg2.clone(), outputs.clone(), [], torch.float32)
    out.update(focal_losses=np.array([float(r) for r in res3], dtype=np.float64))
    # ignore_bbox_thresh (soft-anchor config rnndet-soft.yaml:16)
    head3 = YoloXDetector(make_cfg(32, 32, 0.33, (256, 320), (8, 10), ignore_bbox_thresh=[0.7, 0.35])).yolox_head
    tg3 = tg2.clone()
    tg3[:, :, 5] = torch.rand(tg3.shape[:2], generator=g) * (tg3.sum(2) > 0)
    tg3[:, :, 6] = torch.rand(tg3.shape[:2], generator=g) * (tg3.sum(2) > 0)
    res4 = head3.get_losses(xs, ys, ss, tg3.clone(), outputs.clone(), [], torch.float32)
    out.update(thr_targets=tg3, thr_losses=np.array([float(r) for r in res4], dtype=np.float64))
    save('g06_simota.npz', **out)


def g07_postprocess():
    g = torch.Generator().manual_seed(71)
    out = {}

    def run(name, pred, nc, conf, agnostic=False):
        res = ref_postprocess(pred.clone(), nc, conf, 0.45, class_agnostic=agnostic, pad=torch.zeros((0, 7)))
        out[name + '_pred'] = pred
        out[name + '_n'] = np.array([len(r) for r in res])
        out[name + '_det'] = torch.cat(res, 0)

    A = 400
    base = torch.rand(3, A, 2, generator=g) * torch.tensor([300., 230.])
    pred = torch.cat([base, 10 + 60 * torch.rand(3, A, 2, generator=g), torch.rand(3, A, 1, generator=g),
                      torch.rand(3, A, 2, generator=g)], -1)
    for conf in (0.1, 0.01, 0.001):
        run(f'rand_c{conf}', pred, 2, conf)
    run('rand_agnostic', pred, 2, 0.1, agnostic=True)
    # adversarial: clusters of identical / heavily overlapping boxes, tied scores, 3 classes
    A = 240
    centers = torch.tensor([[50., 50.], [52., 51.], [200., 120.]])
    cid = torch.randint(0, 3, (2, A), generator=g)
    cxy = centers[cid] + torch.round(torch.randn(2, A, 2, generator=g) * 2)
    wh = torch.tensor([40., 30.]).expand(2, A, 2) + torch.round(torch.randn(2, A, 2, generator=g))
    obj = torch.round(torch.rand(2, A, 1, generator=g) * 4) / 4       # many ties
    cls = torch.round(torch.rand(2, A, 3, generator=g) * 4) / 4
    pred = torch.cat([cxy, wh, obj, cls], -1)
    pred[0, 10:20] = pred[0, 10]                                       # identical rows
    run('adv_c0.1', pred, 3, 0.1)
    run('adv_c0.001', pred, 3, 0.001)
    # > 1000 surviving boxes -> torchvision's per-class branch on CPU
    A = 1680
    pred = torch.cat([torch.rand(1, A, 2, generator=g) * torch.tensor([300., 230.]),
                      8 + 20 * torch.rand(1, A, 2, generator=g), 0.5 + 0.5 * torch.rand(1, A, 1, generator=g),
                      0.5 + 0.5 * torch.rand(1, A, 2, generator=g)], -1)
    run('many_c0.001', pred, 2, 0.001)
    # nothing survives
    run('none', pred * torch.tensor([1, 1, 1, 1, 0.01, 0.01, 0.01]), 2, 0.5)
    save('g07_postprocess.npz', **out)


def g08_pseudo():
    g = torch.Generator().manual_seed(81)
    out = {}
    preds = []
    for n in (12, 0, 30):
        x1 = torch.rand(n, generator=g) * 330 - 20
        y1 = torch.rand(n, generator=g) * 260 - 15
        w = torch.rand(n, generator=g) * 120
        h = torch.rand(n, generator=g) * 60
        if n == 30:
            w[:3] = torch.tensor([3., 280., 299.])
        preds.append(torch.stack([x1, y1, x1 + w, y1 + h, torch.rand(n, generator=g), torch.rand(n, generator=g),
                                  torch.randint(0, 2, (n,), generator=g).float()], 1))
    fb = lambda b: ref_ssod.filter_pred_boxes(b, dataset_name='gen1', downsampled_by_2=False)  # noqa
    labs = ref_ssod.pred2label([p.clone() for p in preds], obj_thresh=[0.6, 0.3], cls_thresh=[0.6, 0.3],
                               filter_bbox_fn=fb, hw=(240, 304))
    out['p2l_in'] = torch.cat(preds, 0)
    out['p2l_lens_in'] = np.array([len(p) for p in preds])
    out['p2l_lens'] = np.array([len(l) for l in labs])
    out['p2l_out'] = torch.cat([l.object_labels for l in labs], 0)
    fb4 = lambda b: ref_ssod.filter_pred_boxes(b, dataset_name='gen4', downsampled_by_2=True)  # noqa
    preds4 = [p.clone() for p in preds]
    for p in preds4:
        p[:, :4] *= 2
        p[:, 6] = torch.randint(0, 3, (len(p),), generator=g).float()
    labs4 = ref_ssod.pred2label([p.clone() for p in preds4], obj_thresh=[0.3, 0.3, 0.6], cls_thresh=[0.3, 0.3, 0.6],
                                filter_bbox_fn=fb4, hw=(360, 640))
    out['p2l4_in'] = torch.cat(preds4, 0)
    out['p2l4_lens'] = np.array([len(l) for l in labs4])
    out['p2l4_out'] = torch.cat([l.object_labels for l in labs4], 0)
    labs_f = ref_ssod.pred2label([p.clone() for p in preds], obj_thresh=0.5, cls_thresh=0.4, hw=(240, 304))
    out['p2lf_lens'] = np.array([len(l) for l in labs_f])
    out['p2lf_out'] = torch.cat([l.object_labels for l in labs_f], 0)
    # TTA merge, tensor flavour: two views of a frame concatenated
    views = [torch.cat([preds[0], preds[0] + torch.tensor([1., 1., 1., 1., 0, 0, 0])], 0), preds[2], preds[1]]
    res = ref_tta_postprocess([v.clone() for v in views], conf_thre=0.01, nms_thre=0.45, pad=torch.zeros((0, 7)))
    out['tta_in0'], out['tta_in1'] = views[0], views[1]
    out['tta_n'] = np.array([len(r) for r in res])
    out['tta_out'] = torch.cat(res, 0)
    # label helpers
    ol = ObjectLabels(out_lab := synth_labels(1, (240, 304), 2, seed=8)[0].clone(), (240, 304))
    out['lab_in'] = out_lab.clone()
    out['lab_yolox'] = ol.get_labels_as_tensors('yolox')
    ol.flip_lr_()
    out['lab_flip'] = ol.object_labels
    out['subsample_21_1'] = np.array(ref_ssod.get_subsample_label_idx(21, use_every=1))
    out['subsample_21_5'] = np.array(ref_ssod.get_subsample_label_idx(21, use_every=5))
    out['subsample_10_r3'] = np.array(sorted(ref_ssod.get_subsample_label_idx(10, remove_every=3)))
    save('g08_pseudo.npz', **out)


def g10_voxel():
    rng = np.random.RandomState(101)
    out = {}
    for name, n, fast, cutoff in [('a', 20000, True, None), ('b', 50000, False, 10), ('c', 3000, True, 3)]:
        H, W, bins = 24, 30, 10
        x = rng.randint(0, W, n)
        y = rng.randint(0, H, n)
        p = rng.randint(0, 2, n)
        t = np.sort(rng.randint(1000, 51000, n))
        if name == 'a':   # hot pixel overflowing uint8
            x[:600], y[:600], p[:600] = 3, 4, 1
            t[:600] = t[0]
        rep = StackedHistogram(bins=bins, height=H, width=W, count_cutoff=cutoff, fastmode=fast).construct(
            torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(p), torch.from_numpy(t))
        out.update({f'{name}_x': x, f'{name}_y': y, f'{name}_p': p, f'{name}_t': t, f'{name}_rep': rep})
    # t0 == t1
    x, y, p, t = np.array([1, 2, 2]), np.array([0, 1, 1]), np.array([0, 1, 1]), np.array([5, 5, 5])
    rep = StackedHistogram(bins=4, height=3, width=4).construct(*(torch.from_numpy(a) for a in (x, y, p, t)))
    out.update(d_rep=rep)
    save('g10_voxel.npz', **out)


def tracker_sequences():
    """Deterministic synthetic label sequences for the tracking post-filter: a few persistent objects that drift
    (some of them out of the frame, which exercises the clamp-aware velocity), short-lived false positives, missed
    detections, unlabelled gaps, occasional GT frames and a degenerate zero-area box.  Returned as plain arrays so the
    tests can rebuild the very same input: per sequence (frame_idx [F], list of [n,8] float32 label rows)."""
    seqs = []
    for seed in range(6):
        rng = np.random.RandomState(7000 + seed)
        H, W = (240, 304) if seed % 2 == 0 else (360, 640)
        n_frames = 70 + 10 * seed
        objs = []
        for _ in range(4 + seed % 3):
            t0, life = rng.randint(0, n_frames // 2), rng.randint(8, n_frames)
            w, h = rng.uniform(12, 80), rng.uniform(12, 60)
            objs.append(dict(t0=t0, t1=t0 + life, x=rng.uniform(-10, W - 20), y=rng.uniform(-10, H - 20), w=w, h=h,
                             vx=rng.uniform(-6, 6), vy=rng.uniform(-3, 3), cls=float(rng.randint(0, 2))))
        frame_idx, rows = [], []
        for f in range(n_frames):
            if rng.rand() < 0.25:                      # frame without any label
                continue
            is_gt_frame = rng.rand() < 0.08
            cur = []
            for o in objs:
                if not (o['t0'] <= f < o['t1']) or rng.rand() < 0.15:     # not alive / missed detection
                    continue
                x = o['x'] + o['vx'] * (f - o['t0']) + rng.normal(0, 1.0)
                y = o['y'] + o['vy'] * (f - o['t0']) + rng.normal(0, 1.0)
                x0, y0 = np.clip(x, 0, W - 1), np.clip(y, 0, H - 1)
                x1, y1 = np.clip(x + o['w'], 0, W - 1), np.clip(y + o['h'], 0, H - 1)
                if x1 - x0 < 2 or y1 - y0 < 2:
                    continue
                cur.append([1000. * (f + 1) if is_gt_frame else 0., x0, y0, x1 - x0, y1 - y0, o['cls'], 0.9, 0.8])
            if not is_gt_frame:
                for _ in range(rng.poisson(0.6)):      # false positives
                    cur.append([0., rng.uniform(0, W - 40), rng.uniform(0, H - 40), rng.uniform(8, 40), rng.uniform(8, 40),
                                float(rng.randint(0, 2)), 0.5, 0.5])
                if seed == 3 and f == 20:
                    cur.append([0., 50., 60., 0., 25., 0., 0.5, 0.5])     # zero-area box
            if cur:
                frame_idx.append(f)
                rows.append(np.array(cur, dtype=np.float32))
        seqs.append((np.array(frame_idx, dtype=np.int64), rows, (H, W)))
    return seqs


def g13_tracker():
    """EventSeqData._track_filter of the reference (modules/pseudo_labeler.py:201-333) on the sequences above, for both
    track methods, with and without in-painting.  Stored: which boxes end up with the ignore label, and the in-painted
    boxes per frame."""
    from modules.pseudo_labeler import EventSeqData
    out = {}
    for si, (frame_idx, rows, hw) in enumerate(tracker_sequences()):
        out[f's{si}_frame_idx'] = frame_idx
        out[f's{si}_counts'] = np.array([len(r) for r in rows], dtype=np.int64)
        out[f's{si}_rows'] = np.concatenate(rows, 0)
        out[f's{si}_hw'] = np.array(hw, dtype=np.int64)
        for method in ('forward', 'forward or backward'):
            for inpaint in (False, True):
                cfg = DictConfig(dict(min_track_len=6, track_method=method, inpaint=inpaint, ignore_label=1024))
                seq = EventSeqData(path='none', scale_ratio=1, filter_config=cfg, postproc_cfg=DictConfig({}))
                seq.labels = [ObjectLabels(torch.from_numpy(r.copy()), hw) for r in rows]
                seq.frame_idx = [int(f) for f in frame_idx]
                seq._track_filter()
                tag = f's{si}_{"fb" if "backward" in method else "f"}_{"inp" if inpaint else "noinp"}'
                out[tag + '_frame_idx'] = np.array(seq.frame_idx, dtype=np.int64)
                out[tag + '_counts'] = np.array([len(l) for l in seq.labels], dtype=np.int64)
                lab = [l.object_labels if isinstance(l.object_labels, np.ndarray) else l.object_labels.numpy() for l in seq.labels]
                out[tag + '_rows'] = np.concatenate(lab, 0).astype(np.float32)
    save('g13_tracker.npz', **out)


def g11_manifest():
    man = {}
    for name, ed, dh, fd in [('tiny', 32, 32, 0.33), ('small', 48, 24, 0.33), ('base', 64, 32, 0.67)]:
        for ds, hw, part, nc in [('gen1', (256, 320), (8, 10), 2), ('gen4', (384, 640), (6, 10), 3)]:
            det = YoloXDetector(make_cfg(ed, dh, fd, hw, part, nc))
            man[f'{name}_{ds}'] = manifest_of(det)
    man['micro'] = manifest_of(YoloXDetector(make_cfg(**MICRO)))
    json.dump(man, open(os.path.join(HERE, 'g11_manifest.json'), 'w'))
    print('wrote g11_manifest.json', {k: len(v) for k, v in man.items()})
    # OneCycleLR values as configure_optimizers builds it (modules/detection.py:485-518)
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=2e-4, weight_decay=0)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=2e-4, div_factor=20, final_div_factor=10000 / 20,
                                              total_steps=400000, pct_start=0.005, cycle_momentum=False,
                                              anneal_strategy='linear')
    want = {0, 1, 1000, 1999, 2000, 200000, 399998}
    lrs = {}
    for step in range(399999):
        if step in want:
            lrs[step] = opt.param_groups[0]['lr']
        opt.step()
        sch.step()
    json.dump({str(k): v for k, v in lrs.items()}, open(os.path.join(HERE, 'g11_onecycle.json'), 'w'))
    print('wrote g11_onecycle.json', lrs)


def g12_trainstep():
    """Two consecutive training steps of the micro detector driven exactly like
    Module.training_step (modules/detection.py:150-298) with the reference's own RNNStates /
    BackboneFeatureSelector / ObjectLabels, then clip-by-value + AdamW + OneCycleLR."""
    torch.manual_seed(0)
    det = YoloXDetector(make_cfg(**MICRO))
    load_synth(det, 9)
    det.train()
    opt = torch.optim.AdamW(det.parameters(), lr=2e-4, weight_decay=0)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=2e-4, div_factor=20, final_div_factor=500,
                                              total_steps=1000, pct_start=0.005, cycle_momentum=False,
                                              anneal_strategy='linear')
    padder = InputPadderFromShape(desired_hw=(64, 96))
    rnn = RNNStates()
    T, B = 5, 2
    out = {}
    for step in range(2):
        ev = padder.pad_tensor_ev_repr(synth_events(T, B, 20, 60, 90, seed=20 + step, as_uint8=False))
        lab_list = micro_labels(T * B, seed=30 + step)
        labels = [[lab_list[t * B + b] if (t in (2, 4) or (t == 1 and b == 0)) else None for b in range(B)]
                  for t in range(T)]
        is_first = torch.tensor([True, True]) if step == 0 else torch.tensor([False, True])
        rnn.reset(worker_id=0, indices_or_bool_tensor=is_first)
        prev = rnn.get_states(worker_id=0)
        sel = BackboneFeatureSelector()
        obj_labels = []
        for t in range(T):
            feats, states = det.forward_backbone(x=ev[t], previous_states=prev)
            prev = states
            idx = [b for b in range(B) if labels[t][b] is not None]
            if idx:
                sel.add_backbone_features(backbone_features=feats, selected_indices=idx)
                obj_labels.extend(ObjectLabels(labels[t][b], (60, 90)) for b in idx)
        rnn.save_states_and_detach(worker_id=0, states=prev)
        targets = ObjectLabels.get_labels_as_batched_tensor(obj_label_list=obj_labels, format_='yolox')
        preds, losses = det.forward_detect(backbone_features=sel.get_batched_backbone_features(), targets=targets)
        opt.zero_grad(set_to_none=True)
        losses['loss'].backward()
        torch.nn.utils.clip_grad_value_(det.parameters(), 1.0)
        names = sorted(n for n, p in det.named_parameters() if p.grad is not None)
        out[f's{step}_grad_keys'] = np.array(names)
        gd = dict(det.named_parameters())
        out[f's{step}_grad_norms'] = np.array([float(gd[n].grad.norm()) for n in names], dtype=np.float64)
        out[f's{step}_grad_absmax'] = np.array([float(gd[n].grad.abs().max()) for n in names], dtype=np.float64)
        out[f's{step}_losses'] = np.array([float(losses[k]) for k in
                                          ('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss', 'num_fg')],
                                         dtype=np.float64)
        out[f's{step}_pred_slice'] = preds[:, :40].detach()
        opt.step()
        sch.step()
        out[f's{step}_lr_next'] = np.float64(opt.param_groups[0]['lr'])
        out[f's{step}_param_norms'] = np.array([float(gd[n].detach().norm()) for n in names], dtype=np.float64)
        out[f's{step}_state_c4'] = prev[3][1].detach().clone()  # clone: RNNStates.reset zeroes saved rows in place
    save('g12_trainstep_micro.npz', **out)


def g15_evaluator():
    """Prophesee evaluation up to the COCO records: the reference's box filters, +-50 ms time matching and
    `_to_coco_format` on synthetic label / detection records (pycocotools itself is absent: the AP numbers are not in
    this fixture, see oracle/coco_eval.py)."""
    from oracle.synth import synth_eval_sequences, EVAL_CASES
    from utils.evaluation.prophesee.io.box_filtering import filter_boxes
    from utils.evaluation.prophesee.metrics import coco_eval as ref_ce
    from utils.evaluation.prophesee.io.box_loading import to_prophesee, BBOX_DTYPE as REF_EVAL_DTYPE
    out = {}
    for ci, case in enumerate(EVAL_CASES):
        labels, dets = synth_eval_sequences(**case)
        gen4 = case.get('n_cls', 2) == 3
        diag, side = (30, 10) if gen4 else (30, 10)          # gen4 case is the downsampled one: 60/2, 20/2
        gts, dts = [], []
        for k, (g, d) in enumerate(zip(labels, dets)):
            gf, df = filter_boxes(g, int(5e5), diag, side), filter_boxes(d, int(5e5), diag, side)
            # which rows survived (the records are distinct, so membership identifies them)
            out[f'c{ci}_s{k}_gt_kept'] = np.array([r.item() in {x.item() for x in gf} for r in g], dtype=bool)
            out[f'c{ci}_s{k}_dt_kept'] = np.array([r.item() in {x.item() for x in df} for r in d], dtype=bool)
            out[f'c{ci}_s{k}_n_gt'], out[f'c{ci}_s{k}_n_dt'] = np.int64(len(gf)), np.int64(len(df))
            gw, dw = ref_ce._match_times(np.unique(gf['t']), gf, df, 50000)
            gts += gw
            dts += dw
        out[f'c{ci}_gt_cnt'] = np.array([len(x) for x in gts], dtype=np.int64)
        out[f'c{ci}_dt_cnt'] = np.array([len(x) for x in dts], dtype=np.int64)
        cats = [dict(id=i + 1, name=str(i), supercategory='none') for i in range(case.get('n_cls', 2))]
        dataset, results = ref_ce._to_coco_format(gts, dts, cats, height=case.get('hw', (240, 304))[0],
                                                 width=case.get('hw', (240, 304))[1])
        ann = dataset['annotations']
        out[f'c{ci}_ann_area'] = np.array([a['area'] for a in ann], dtype=np.float64)
        out[f'c{ci}_ann_bbox'] = np.array([[float(v) for v in a['bbox']] for a in ann], dtype=np.float64).reshape(-1, 4)
        out[f'c{ci}_ann_cat'] = np.array([a['category_id'] for a in ann], dtype=np.int64)
        out[f'c{ci}_ann_img'] = np.array([a['image_id'] for a in ann], dtype=np.int64)
        out[f'c{ci}_res_score'] = np.array([r['score'] for r in results], dtype=np.float64)
        out[f'c{ci}_res_bbox'] = np.array([[float(v) for v in r['bbox']] for r in results], dtype=np.float64).reshape(-1, 4)
        out[f'c{ci}_res_cat'] = np.array([r['category_id'] for r in results], dtype=np.int64)
        out[f'c{ci}_res_img'] = np.array([r['image_id'] for r in results], dtype=np.int64)
        out[f'c{ci}_n_img'] = np.int64(len(dataset['images']))
    # to_prophesee: labels + post-processed detections of single frames -> records
    g = torch.Generator().manual_seed(15)
    labs, preds = [], []
    for f in range(4):
        n = 1 + f
        l = torch.rand((n, 8), generator=g) * 50
        l[:, 0] = 1000000 + 50000 * f
        l[:, 5] = torch.randint(0, 2, (n,), generator=g).float()
        labs.append(ObjectLabels(l, (240, 304)))
        m = [3, 0, 2, 1][f]
        p = torch.rand((m, 7), generator=g) * 100
        p[:, 2:4] += p[:, 0:2]
        p[:, 6] = torch.randint(0, 2, (m,), generator=g).float()
        preds.append(p if m else None)
    assert REF_EVAL_DTYPE.itemsize == 40
    lp, pp = to_prophesee(labs, preds)
    for f in range(4):
        for name in REF_EVAL_DTYPE.names:                       # field by field (the 4 padding bytes of a record are not data)
            out[f'proph_lab{f}_{name}'] = lp[f][name]
            out[f'proph_pred{f}_{name}'] = pp[f][name]
    save('g15_evaluator.npz', **out)


def g16_tta_result():
    """EventSeqResult (modules/utils/tta.py:64-195) driven with scripted TTA views of one recording: plain, h-flipped,
    time-reversed (offset -1) and both; 8 frames of which 5 carry labels; the merged detections and label records."""
    from oracle.synth import synth_tta_views
    from modules.utils.tta import EventSeqResult
    out = {}
    for case in range(3):
        views, hw = synth_tta_views(case)
        res = EventSeqResult(path='seq', img_hw=hw, postproc_cfg=DictConfig(dict(confidence_threshold=0.1, nms_threshold=0.45)))
        for v in views:
            gts = [ObjectLabels(g.clone(), hw) if torch.is_tensor(g) else g for g in v['gts']]
            preds = [p.clone() if torch.is_tensor(p) else p for p in v['preds']]
            res.update(is_hflip=v['hflip'], is_tflip=v['tflip'], preds=preds, gts=gts, ev_idx=list(v['ev_idx']),
                       is_last_sample=v['last'], tflip_offset=-1)
        labels, preds = res.aggregate_results()
        out[f'c{case}_n'] = np.int64(len(labels))
        for k, (l, p) in enumerate(zip(labels, preds)):
            for name in l.dtype.names:
                out[f'c{case}_lab{k}_{name}'] = l[name]
                out[f'c{case}_pred{k}_{name}'] = p[name]
    save('g16_tta_result.npz', **out)


def g14_augment():
    """RandomSpatialAugmentorGenX.__call__ of the reference (data/utils/augmentor.py:455-476) on seeded loader samples with
    the shipped augmentation config (hflip 0.5, zoom 0.8: in 8 / out 2): resulting augmentation state, augmented uint8
    tensors and transformed labels."""
    from data.utils.augmentor import RandomSpatialAugmentorGenX
    from data.genx_utils.labels import SparselyBatchedObjectLabels
    from data.utils.types import DataType
    from oracle.synth import synth_augment_sample, AUGMENT_CASES, AUGMENT_CFG
    cfg = DictConfig(AUGMENT_CFG)
    out = {}
    for seed, H, W in AUGMENT_CASES:
        ev, labels = synth_augment_sample(seed, H, W)
        aug = RandomSpatialAugmentorGenX(dataset_hw=(H, W), automatic_randomization=True, augm_config=cfg)
        objs = [None if l is None else ObjectLabels(l.clone(), (H, W)) for l in labels]
        torch.manual_seed(900 + seed)
        res = aug({DataType.EV_REPR: [e.clone() for e in ev], DataType.OBJLABELS_SEQ: SparselyBatchedObjectLabels(objs)})
        st = res[DataType.AUGM_STATE]
        out[f's{seed}_state'] = np.array([float(st.apply_h_flip), float(st.zoom_in.active), st.zoom_in.x0, st.zoom_in.y0,
                                          st.zoom_in.zoom_in_factor, float(st.zoom_out.active), st.zoom_out.x0,
                                          st.zoom_out.y0, st.zoom_out.zoom_out_factor], dtype=np.float64)
        out[f's{seed}_ev'] = torch.stack(res[DataType.EV_REPR]).numpy()
        for t, l in enumerate(res[DataType.OBJLABELS_SEQ]):
            if l is not None:
                out[f's{seed}_lab{t}'] = l.object_labels.numpy().astype(np.float32)
                out[f's{seed}_hw{t}'] = np.array(l.input_size_hw, dtype=np.float64)
    save('g14_augment.npz', **out)


def record_sample(out, key, sample, DataType):
    """Flatten one loader sample into arrays under ``key``."""
    if DataType.EV_REPR in sample:
        out[f'{key}_ev'] = torch.stack(list(sample[DataType.EV_REPR])).numpy()
        out[f'{key}_idx'] = np.asarray(sample[DataType.EV_IDX], dtype=np.int64)
        out[f'{key}_pad'] = np.asarray(sample[DataType.IS_PADDED_MASK], dtype=bool)
        out[f'{key}_flags'] = np.asarray([sample[DataType.IS_FIRST_SAMPLE], sample[DataType.IS_LAST_SAMPLE], sample[DataType.IS_REVERSED]], dtype=bool)
        out[f'{key}_path'] = np.asarray(os.path.basename(sample[DataType.PATH]))
    for name, k in (('lab', DataType.OBJLABELS_SEQ), ('skip', DataType.SKIPPED_OBJLABELS_SEQ)):
        rows, where = [], []
        for t, l in enumerate(sample[k]):
            if l is not None:
                rows.append(l.object_labels.numpy().astype(np.float32))
                where += [t] * len(l)
                out[f'{key}_{name}hw{t}'] = np.asarray(l.input_size_hw, dtype=np.float64)
        out[f'{key}_{name}'] = np.concatenate(rows) if rows else np.zeros((0, 8), np.float32)
        out[f'{key}_{name}_t'] = np.asarray(where, dtype=np.int64)


def g17_loader():
    """The reference's own sequence classes (data/genx_utils/sequence_streaming.py:54-277, sequence_rnd.py:11-148,
    sequence_base.py:52-205) over a synthetic dataset tree (oracle.synth.synth_dataset_tree; frames come from the .npy twin
    through the h5py stand-in): every sample of every case -- frames, frame indices, padding masks, first / last / reversed
    flags, visible and withheld labels -- plus the sub-sequence ranges of get_sequences_with_guaranteed_labels and the
    recording -> worker dealing of ShardedStreamingDataPipe.assign_datapipes_to_worker."""
    import tempfile
    from pathlib import Path
    from oracle.synth import synth_dataset_tree, loader_cases
    from data.genx_utils.sequence_streaming import SequenceForIter as RefIter
    from data.genx_utils.sequence_rnd import SequenceForRandomAccess as RefRnd
    from data.utils.stream_sharded_datapipe import ShardedStreamingDataPipe as RefSharded
    from data.utils.types import DataType, DatasetType
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for dst, ds2, dtype in (('gen1', False, DatasetType.GEN1), ('gen4', True, DatasetType.GEN4)):
            base = synth_dataset_tree(tmp, dst, ds2)
            common = dict(ev_representation_name='stacked_histogram_dt=50_nbins=10', dataset_type=dtype, downsample_by_factor_2=ds2,
                          tflip_offset=-1 if dst == 'gen1' else -2)
            for ci, (kind, rec, kw, tf, _) in enumerate(loader_cases()):
                if dst == 'gen4' and ci % 3:                  # a third of the cases at half resolution is plenty
                    continue
                path = Path(base) / 'train' / rec
                seq = (RefIter(path=path, **common, **kw) if kind == 'iter'
                       else RefRnd(path=path, only_load_end_labels=False, **common, **kw))
                seq.time_flip = tf
                out[f'{dst}_c{ci}_len'] = np.asarray(len(seq))
                for i in range(len(seq)):
                    np.random.seed(1000 * ci + i)             # _rand_another draws from numpy's global RNG
                    try:
                        sample = seq[i]
                    except ValueError:                        # replacement draw from an empty pool (one sample, reversed view)
                        out[f'{dst}_c{ci}_s{i}_err'] = np.asarray(1)
                        continue
                    record_sample(out, f'{dst}_c{ci}_s{i}', sample, DataType)
            for rec in ('rec_a', 'rec_c', 'rec_e'):
                for L in (3, 5, 9):
                    subs = RefIter.get_sequences_with_guaranteed_labels(path=Path(base) / 'train' / rec, sequence_length=L, **common)
                    out[f'{dst}_{rec}_L{L}_ranges'] = np.asarray([[s.start_indices[0], s.stop_indices[-1], len(s)] for s in subs], dtype=np.int64)
    lens = [7, 3, 9, 9, 1, 4, 12, 2, 5, 5, 6]

    class _DP:
        def __init__(self, n, tag):
            self.n, self.tag = n, tag

        def __len__(self):
            return self.n
    pipes = sorted([_DP(n, i) for i, n in enumerate(lens)], key=lambda x: len(x), reverse=True)
    out['shard_sorted'] = np.asarray([p.tag for p in pipes], dtype=np.int64)
    for total in (1, 2, 3, 4, 8):
        for w in range(total):
            got = RefSharded.assign_datapipes_to_worker(pipes, total_num_workers=total, global_worker_id=w)
            out[f'shard_w{total}_{w}'] = np.asarray([p.tag for p in got], dtype=np.int64)
    save('g17_loader.npz', **out)


# ====================================================================================================
class _CudaLikeFp32Ops(torch.overrides.TorchFunctionMode):
    """CUDA autocast keeps layer_norm / softmax / BCE-with-logits in its fp32 list (they upcast their inputs and return fp32);
    CPU autocast lets them run in the input dtype (bf16).  Inside this mode the CPU autocast run gets the CUDA placement the
    reference trains with (Lightning precision=16, train.py:236-243)."""
    FP32 = None

    def __torch_function__(self, func, types, args=(), kwargs=None):
        import torch.nn.functional as F
        kwargs = kwargs or {}
        if _CudaLikeFp32Ops.FP32 is None:
            _CudaLikeFp32Ops.FP32 = {F.layer_norm, torch.layer_norm, torch.Tensor.softmax, torch.softmax, F.softmax,
                                     F.binary_cross_entropy_with_logits, torch.Tensor.exp, torch.exp, torch.Tensor.log, torch.log}
        if func in _CudaLikeFp32Ops.FP32:
            up = lambda a: a.float() if torch.is_tensor(a) and a.is_floating_point() and a.dtype != torch.float32 else a
            args = tuple(up(a) for a in args)
            kwargs = {k: up(v) for k, v in kwargs.items()}
        return func(*args, **kwargs)


def _ref_train_step(det, ev_padded, labels, hw, T, B, mode):
    """One forward + backward of the reference detector driven like Module.training_step (as g12), in ``mode``:
    'fp32' | 'ac' (torch.autocast('cpu', bfloat16) as it is) | 'acf' (the same with CUDA autocast's fp32-op placement) |
    'h16' / 'h16f' (the same two placements under torch.autocast('cpu', float16) -- the dtype the reference trains with,
    Lightning precision=16, train.py:236-243 -- with the loss scaled as GradScaler does: start at 2**16, halve until every
    gradient is finite, un-scale afterwards).
    -> dict(losses[6], grads {name: tensor}, feats of the last timestep {stage: tensor}, final LSTM (h, c) per stage)"""
    import contextlib
    det.train()
    half = mode in ('h16', 'h16f')
    scale = 65536.0 if half else 1.0
    while True:
        for p_ in det.parameters():
            p_.grad = None
        ctx = contextlib.ExitStack()
        if mode in ('ac', 'acf', 'h16', 'h16f'):
            ctx.enter_context(torch.autocast('cpu', dtype=torch.float16 if half else torch.bfloat16))
        if mode in ('acf', 'h16f'):
            ctx.enter_context(_CudaLikeFp32Ops())
        with ctx:
            rnn = RNNStates()
            rnn.reset(worker_id=0, indices_or_bool_tensor=torch.ones(B, dtype=torch.bool))
            prev = rnn.get_states(worker_id=0)
            sel = BackboneFeatureSelector()
            obj_labels = []
            feats = None
            for t in range(T):
                feats, prev = det.forward_backbone(x=ev_padded[t], previous_states=prev)
                idx = [b for b in range(B) if labels[t][b] is not None]
                if idx:
                    sel.add_backbone_features(backbone_features=feats, selected_indices=idx)
                    obj_labels.extend(ObjectLabels(labels[t][b], hw) for b in idx)
            targets = ObjectLabels.get_labels_as_batched_tensor(obj_label_list=obj_labels, format_='yolox')
            preds, losses = det.forward_detect(backbone_features=sel.get_batched_backbone_features(), targets=targets)
            (losses['loss'].float() * scale).backward()
        if not half or all(bool(torch.isfinite(p_.grad).all()) for p_ in det.parameters() if p_.grad is not None) or scale <= 1.0:
            break
        scale /= 2.0                                   # GradScaler: skip the step, back off, try again
    return dict(losses=np.array([float(losses[k]) for k in ('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss', 'num_fg')], dtype=np.float64),
                grads={n: (p_.grad.detach().float() / scale).clone() for n, p_ in det.named_parameters() if p_.grad is not None},
                feats={k: v.detach().float().clone() for k, v in feats.items()},
                states=[(h.detach().float().clone(), c.detach().float().clone()) for h, c in prev], loss_scale=scale)


def _rel(a, b):
    """||a - b|| / ||b|| (0 when both vanish)"""
    d, n = float((a.double() - b.double()).norm()), float(b.double().norm())
    return d / n if n > 0 else (0.0 if d == 0 else float('inf'))


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    n = float(a.norm() * b.norm())
    return float(a @ b) / n if n > 0 else 1.0


def _class_record(out, tag, ref, run):
    """Deviation of one 16-bit run of the reference from its fp32 run: what the HIP bf16 mode is allowed to differ by."""
    names = sorted(ref['grads'])
    out[f'{tag}_losses'] = run['losses']
    out[f'{tag}_loss_scale'] = np.float64(run.get('loss_scale', 1.0))
    out[f'{tag}_grad_rel'] = np.array([_rel(run['grads'][n], ref['grads'][n]) for n in names], dtype=np.float64)
    out[f'{tag}_grad_cos'] = np.array([_cos(run['grads'][n], ref['grads'][n]) for n in names], dtype=np.float64)
    flat = lambda g: torch.cat([g[n].flatten() for n in names])
    out[f'{tag}_grad_cos_global'] = np.float64(_cos(flat(run['grads']), flat(ref['grads'])))
    out[f'{tag}_grad_rel_global'] = np.float64(_rel(flat(run['grads']), flat(ref['grads'])))
    out[f'{tag}_feat_rel'] = np.array([_rel(run['feats'][k], ref['feats'][k]) for k in sorted(ref['feats'])], dtype=np.float64)
    out[f'{tag}_feat_maxrel'] = np.array([float((run['feats'][k] - ref['feats'][k]).abs().max() / ref['feats'][k].abs().max())
                                          for k in sorted(ref['feats'])], dtype=np.float64)
    out[f'{tag}_state_h_rel'] = np.array([_rel(h, rh) for (h, _), (rh, _) in zip(run['states'], ref['states'])], dtype=np.float64)
    out[f'{tag}_state_c_rel'] = np.array([_rel(c, rc) for (_, c), (_, rc) in zip(run['states'], ref['states'])], dtype=np.float64)
    out[f'{tag}_state_c_maxrel'] = np.array([float((c - rc).abs().max() / rc.abs().max())
                                             for (_, c), (_, rc) in zip(run['states'], ref['states'])], dtype=np.float64)


def g18_autocast(full_size=True):
    """The reference's own 16-bit classes: the reference run in fp32 and under torch.autocast on the CPU,
    in bfloat16 ('ac' / 'acf') and in float16 with GradScaler-style loss scaling ('h16' / 'h16f': the dtype the reference itself
    trains with on CUDA, Lightning precision=16, train.py:236-243), each once as CPU autocast places ops ('ac', 'h16') and once
    with CUDA autocast's fp32-op list emulated ('acf', 'h16f').  Stored: the fp32 run's losses /
    gradient norms (so that the tests can pin their own fp32 side to it) and, per tensor, how far each 16-bit run is from fp32.
    The -m gpu tests bound the HIP bf16 mode's deviation from fp32 by 1.5 x the bf16 class and print / record it as a multiple
    of the fp16 class.
      micro_*   micro detector, T=5 B=2 (the g12 set-up, step 0)
      tiny_*    RVT-tiny at 256x320, T=2 B=1 forward features (the g04 set-up)
      small_*   RVT-small Gen1 240x304 T=21 bs=8, the benchmark workload (bench.make_batch seed 7): scalars only"""
    out = {}
    # ---- micro train step -------------------------------------------------------------------------------------------------
    torch.manual_seed(0)
    det = YoloXDetector(make_cfg(**MICRO))
    load_synth(det, 9)
    T, B = 5, 2
    ev = InputPadderFromShape(desired_hw=(64, 96)).pad_tensor_ev_repr(synth_events(T, B, 20, 60, 90, seed=20, as_uint8=False))
    lab_list = micro_labels(T * B, seed=30)
    labels = [[lab_list[t * B + b] if (t in (2, 4) or (t == 1 and b == 0)) else None for b in range(B)] for t in range(T)]
    runs = {m: _ref_train_step(det, ev, labels, (60, 90), T, B, m) for m in ('fp32', 'ac', 'acf', 'h16', 'h16f')}
    names = sorted(runs['fp32']['grads'])
    out['micro_grad_keys'] = np.array(names)
    out['micro_fp32_losses'] = runs['fp32']['losses']
    out['micro_fp32_grad_norms'] = np.array([float(runs['fp32']['grads'][n].double().norm()) for n in names], dtype=np.float64)
    out['micro_fp32_feat_norms'] = np.array([float(runs['fp32']['feats'][k].double().norm()) for k in sorted(runs['fp32']['feats'])])
    for m in ('ac', 'acf', 'h16', 'h16f'):
        _class_record(out, f'micro_{m}', runs['fp32'], runs[m])
    # ---- RVT-tiny forward at the real geometry ----------------------------------------------------------------------------
    det = YoloXDetector(make_cfg(32, 32, 0.33, (256, 320), (8, 10)))
    load_synth(det, 6)
    det.eval()
    ev = InputPadderFromShape(desired_hw=(256, 320)).pad_tensor_ev_repr(synth_events(2, 1, 20, 240, 304, seed=5, as_uint8=False))
    import contextlib
    fe = {}
    for m in ('fp32', 'ac', 'acf', 'h16', 'h16f'):
        with contextlib.ExitStack() as ctx, torch.no_grad():
            if m != 'fp32':
                ctx.enter_context(torch.autocast('cpu', dtype=torch.float16 if m.startswith('h16') else torch.bfloat16))
            if m in ('acf', 'h16f'):
                ctx.enter_context(_CudaLikeFp32Ops())
            f_, st = det.forward_backbone(ev[0], None)
            f_, st = det.forward_backbone(ev[1], st)
            fe[m] = {k: v.float() for k, v in f_.items()}
    out['tiny_fp32_feat_norms'] = np.array([float(fe['fp32'][k].double().norm()) for k in sorted(fe['fp32'])])
    for m in ('ac', 'acf', 'h16', 'h16f'):
        out[f'tiny_{m}_feat_rel'] = np.array([_rel(fe[m][k], fe['fp32'][k]) for k in sorted(fe['fp32'])], dtype=np.float64)
        out[f'tiny_{m}_feat_maxrel'] = np.array([float((fe[m][k] - fe['fp32'][k]).abs().max() / fe['fp32'][k].abs().max())
                                                 for k in sorted(fe['fp32'])], dtype=np.float64)
    # ---- the benchmark workload -------------------------------------------------------------------------------------------
    if full_size:
        import bench
        import time
        T, B, hw = 21, 8, (240, 304)
        evu, _, label_tb, labs = bench.make_batch(T, B, hw, 2, 7, 'cpu', (4, 9, 14, 19))
        it = iter(labs)
        labels = []
        for t in range(T):
            row = [None] * B
            for b in label_tb[t]:
                l = next(it)
                row[b] = torch.from_numpy(np.concatenate([np.ones((len(l), 1), np.float32), l[:, 1:2] - l[:, 3:4] / 2,
                                                          l[:, 2:3] - l[:, 4:5] / 2, l[:, 3:5], l[:, 0:1], l[:, 6:7], l[:, 5:6]], 1))
            labels.append(row)
        man = json.load(open(os.path.join(HERE, 'g11_manifest.json')))['small_gen1']
        det = YoloXDetector(make_cfg(embed_dim=48, dim_head=24, fpn_depth=0.33, in_hw=(256, 320), part=(8, 10)))
        det.load_state_dict(synth_state_dict(man, 0), strict=True)
        x = InputPadderFromShape(desired_hw=(256, 320)).pad_tensor_ev_repr(evu.to(torch.float32))
        runs = {}
        for m in ('fp32', 'ac', 'acf', 'h16', 'h16f'):
            t0 = time.time()
            runs[m] = _ref_train_step(det, x, labels, hw, T, B, m)
            print(f'small {m}: {time.time() - t0:.1f} s, losses {runs[m]["losses"]}')
        names = sorted(runs['fp32']['grads'])
        out['small_grad_keys'] = np.array(names)
        out['small_fp32_losses'] = runs['fp32']['losses']
        out['small_fp32_grad_norms'] = np.array([float(runs['fp32']['grads'][n].double().norm()) for n in names], dtype=np.float64)
        for m in ('ac', 'acf', 'h16', 'h16f'):
            _class_record(out, f'small_{m}', runs['fp32'], runs[m])
    save('g18_autocast.npz', **out)
    for k in sorted(out):
        if 'global' in k or k.endswith('losses') or 'feat_rel' in k or 'state_c_rel' in k:
            print(k, out[k])


def g19_head_options():
    """The two head-loss options every shipped config leaves off (yolo_head.py:335-381): ``bbox_loss_weighting`` and ``ignore_bg_k``.
    Losses AND the gradient wrt the decoded predictions of the reference, on the Gen1 anchors; the batch holds an image without labels."""
    from oracle.head import make_grids
    gx, gy, gs = make_grids([(32, 40), (16, 20), (8, 10)], (8, 16, 32))
    A = gx.numel()
    g = torch.Generator().manual_seed(191)
    B = 4
    labs = synth_labels(B, (240, 304), 2, seed=190, max_boxes=6, min_boxes=2)
    tg = ObjectLabels.get_labels_as_batched_tensor([ObjectLabels(l, (240, 304)) for l in labs])
    tg[3] = 0
    nz = (tg.sum(2) > 0).float()
    tg[:, :, 5] = (0.3 + 0.7 * torch.rand(tg.shape[:2], generator=g)) * nz        # objectness / class confidences of pseudo boxes
    tg[:, :, 6] = (0.3 + 0.7 * torch.rand(tg.shape[:2], generator=g)) * nz
    outputs = torch.cat([
        torch.stack([(gx + 0.5) * gs, (gy + 0.5) * gs, gs * 3, gs * 2.5], 1)[None].repeat(B, 1, 1)
        + torch.randn(B, A, 4, generator=g), torch.randn(B, A, 3, generator=g)], -1)
    xs = [gx[None, :1280], gx[None, 1280:1600], gx[None, 1600:]]
    ys = [gy[None, :1280], gy[None, 1280:1600], gy[None, 1600:]]
    ss = [gs[None, :1280], gs[None, 1280:1600], gs[None, 1600:]]
    tg_ign = tg.clone()
    tg_ign[0, 1, 0] = 1024
    out = dict(targets=tg, targets_ign=tg_ign, outputs=outputs)
    cases = dict(w_obj=dict(bbox_loss_weighting='obj'), w_cls=dict(bbox_loss_weighting='cls'), w_objxcls=dict(bbox_loss_weighting='objxcls'),
                 w_cls_sq=dict(bbox_loss_weighting='cls-w**2'), bg05=dict(ignore_bg_k=0.05), bg30=dict(ignore_bg_k=0.3),
                 w_obj_bg10=dict(bbox_loss_weighting='obj', ignore_bg_k=0.1), w_obj_focal=dict(bbox_loss_weighting='obj', obj_focal_loss=True))
    for name, kw in cases.items():
        head = YoloXDetector(make_cfg(32, 32, 0.33, (256, 320), (8, 10), **kw)).yolox_head
        for suffix, targets in (('', tg), ('_ign', tg_ign)):              # '_ign': a batch WITH an ignore box (get_losses_w_ignore, no top-k step)
            o = outputs.clone().requires_grad_(True)
            res = head.get_losses(xs, ys, ss, targets.clone(), o, [], torch.float32)
            res[0].backward()
            out[f'{name}{suffix}_losses'] = np.array([float(r) for r in res], dtype=np.float64)
            out[f'{name}{suffix}_grad'] = o.grad.clone()
    save('g19_head_options.npz', **out)


def _ref_trajectory(mode, steps, perturb=0.0, seed=0, T=5, B=2, nb=16, hw=(60, 90), pad=(64, 96)):
    """``steps`` AdamW steps of the micro detector through ONE OneCycle schedule (peak 2e-4, pct_start 0.1, the reference's shape,
    modules/detection.py:498-511) on ``nb`` cycled batches, driven like Module.training_step (sample 0 streams its LSTM state, sample 1
    restarts every step), in ``mode``: 'fp32' | 'h16f' (torch.autocast(float16) with CUDA autocast's fp32-op placement and GradScaler-style
    loss scaling: the class the reference trains in, train.py:236-243) | 'acf' (the same placement in bfloat16).  ``perturb``: relative
    Gaussian noise applied ONCE to the initial weights (the chaos control).  -> loss per step [steps]"""
    import contextlib
    padder = InputPadderFromShape(desired_hw=pad)
    data = []
    for i in range(nb):
        ev = padder.pad_tensor_ev_repr(synth_events(T, B, 20, hw[0], hw[1], seed=700 + i, as_uint8=False))
        labs = micro_labels(T * B, seed=800 + i)
        data.append((ev, [[labs[t * B + b] if (t in (2, 4) or (t == 1 and b == 0)) else None for b in range(B)] for t in range(T)]))
    torch.manual_seed(0)
    det = YoloXDetector(make_cfg(**MICRO))
    load_synth(det, 9)
    if perturb:
        g = torch.Generator().manual_seed(100 + seed)
        with torch.no_grad():
            for p in det.parameters():
                p.mul_(1.0 + perturb * torch.randn(p.shape, generator=g))
    det.train()
    opt = torch.optim.AdamW(det.parameters(), lr=2e-4, weight_decay=0)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=2e-4, div_factor=20, final_div_factor=500, total_steps=steps, pct_start=0.1,
                                              cycle_momentum=False, anneal_strategy='linear')
    half = mode == 'h16f'
    scale, good = 65536.0, 0
    rnn = RNNStates()
    seen = []
    for step in range(steps):
        ev, labels = data[step % nb]
        while True:
            opt.zero_grad(set_to_none=True)
            ctx = contextlib.ExitStack()
            if mode in ('h16f', 'acf'):
                ctx.enter_context(torch.autocast('cpu', dtype=torch.float16 if half else torch.bfloat16))
                ctx.enter_context(_CudaLikeFp32Ops())
            with ctx:
                rnn.reset(worker_id=0, indices_or_bool_tensor=torch.tensor([step == 0, True]))
                prev = rnn.get_states(worker_id=0)
                sel = BackboneFeatureSelector()
                obj_labels = []
                for t in range(T):
                    feats, prev = det.forward_backbone(x=ev[t], previous_states=prev)
                    idx = [b for b in range(B) if labels[t][b] is not None]
                    if idx:
                        sel.add_backbone_features(backbone_features=feats, selected_indices=idx)
                        obj_labels.extend(ObjectLabels(labels[t][b], hw) for b in idx)
                targets = ObjectLabels.get_labels_as_batched_tensor(obj_label_list=obj_labels, format_='yolox')
                _, losses = det.forward_detect(backbone_features=sel.get_batched_backbone_features(), targets=targets)
                (losses['loss'].float() * (scale if half else 1.0)).backward()
            if not half:
                break
            if all(bool(torch.isfinite(p.grad).all()) for p in det.parameters() if p.grad is not None):
                with torch.no_grad():
                    for p in det.parameters():
                        if p.grad is not None:
                            p.grad.div_(scale)
                good += 1
                if good % 2000 == 0:
                    scale *= 2.0
                break
            scale, good = scale / 2.0, 0                       # GradScaler: skip the step, back off, try again
        rnn.save_states_and_detach(worker_id=0, states=[(h.float(), c.float()) for h, c in prev])
        torch.nn.utils.clip_grad_value_(det.parameters(), 1.0)
        opt.step()
        sch.step()
        seen.append(float(losses['loss']))
    return np.array(seen, dtype=np.float64)


def g20_trajectory(steps=200):
    """Whole training trajectories of the reference (micro geometry, ~5 min of CPU): its fp32 run, two fp32 runs from initial weights perturbed
    by 2^-12 / 2^-9 (how chaotic the observable is in the reference itself) and its fp16-autocast run -- what a 200-step run of the HIP
    precision modes is compared with (tests/test_module_gpu.py::test_module_200_step_trajectory_vs_reference)."""
    out = {}
    for name, mode, eps, seed in (('fp32', 'fp32', 0.0, 0), ('fp32_p12', 'fp32', 2.0 ** -12, 1), ('fp32_p9', 'fp32', 2.0 ** -9, 3), ('h16f', 'h16f', 0.0, 0)):
        out[name] = _ref_trajectory(mode, steps, eps, seed)
        print(name, np.round(out[name][::20], 3), 'last-20 mean', round(float(out[name][-20:].mean()), 3))
    save('g20_trajectory_micro.npz', **out)


def g21_label_quality():
    """The pseudo-label quality statistics of the reference (modules/utils/ssod.py:209-350: ``evaluate_label``, ``get_scores_ious``) on seeded
    synthetic GT / pseudo-label lists of a two-class and a three-class head: perturbed copies of the GT boxes (hits at various IoUs), dropped
    boxes (misses), extra boxes (false positives), an empty pseudo frame, a frame without GT and a frame masked out by ``pred_mask``."""
    from modules.utils.ssod import evaluate_label, get_scores_ious
    out = {}
    for tag, nc, hw in (('gen1', 2, (240, 304)), ('gen4', 3, (360, 640))):
        g = torch.Generator().manual_seed(211 + nc)
        gts = synth_labels(8, hw, nc, seed=210 + nc, max_boxes=6, min_boxes=1)
        gt_l, pse_l, rows = [], [], []
        for i, lab in enumerate(gts):
            p = lab.clone()
            p[:, 1:5] += torch.randn(p[:, 1:5].shape, generator=g) * torch.tensor([4., 4., 6., 6.]) * (i % 3)      # frames 0, 3, 6: exact copies
            p[:, 3:5].clamp_(min=2.)
            p[:, 6] = torch.rand(len(p), generator=g)
            p[:, 7] = torch.rand(len(p), generator=g)
            if i % 4 == 1:
                p = p[:-1]                                             # a missed box
            if i % 4 == 2:
                extra = p[:1].clone()
                extra[:, 1:3] = torch.tensor([5., 7.])
                extra[:, 5] = float((i // 4) % nc)
                p = torch.cat([p, extra])                              # a false positive
            if i == 5:
                p = p[:0]                                              # nothing predicted on this frame
            p[:, 0] = 0
            gt_l.append(None if i == 7 else ObjectLabels(lab, hw))
            pse_l.append(ObjectLabels(p, hw))
            rows.append((lab, p))
        mask = np.ones(8, dtype=bool)
        mask[4] = False
        ev = evaluate_label(gt_l, pse_l, pred_mask=mask, num_cls=nc, prefix='ssod/')
        sc = get_scores_ious(gt_l, pse_l, pred_mask=mask, num_cls=nc, prefix='ssod/')
        for i, (lab, p) in enumerate(rows):
            out[f'{tag}_gt{i}'], out[f'{tag}_pse{i}'] = lab, p
        out[f'{tag}_eval_keys'] = np.array(sorted(ev))
        out[f'{tag}_eval_vals'] = np.array([float(ev[k]) for k in sorted(ev)], dtype=np.float64)
        for k, v in sc.items():
            out[f'{tag}_scores_' + k.replace('/', '_')] = np.array(v, dtype=np.float64)
    save('g21_label_quality.npz', **out)


ALL = dict(g01=g01_convlstm, g02=g02_partition, g03=g03_attention, g04=g04_backbone, g05=g05_head,
           g06=g06_simota, g07=g07_postprocess, g08=g08_pseudo, g10=g10_voxel, g11=g11_manifest,
           g12=g12_trainstep, g13=g13_tracker, g14=g14_augment, g15=g15_evaluator, g16=g16_tta_result, g17=g17_loader, g18=g18_autocast,
           g19=g19_head_options, g20=g20_trajectory, g21=g21_label_quality)

def g22_depthwise():
    """The depthwise options the shipped configs leave off: ``DWSConvLSTM2d(dws_conv=True)`` in both placements
    (rnn.py:20-30,50-55) over three chained timesteps with gradients, and PAFPN + head built with ``depthwise=True``
    (network_blocks.py:57-76, yolo_pafpn.py:37, yolo_head.py:52): eval predictions, training losses, every parameter gradient."""
    out = {}
    for tag, only_hidden, ks in (('h', True, 3), ('xh', False, 3), ('h5', True, 5)):
        m = DWSConvLSTM2d(dim=16, dws_conv=True, dws_conv_only_hidden=only_hidden, dws_conv_kernel_size=ks)
        load_synth(m, 21)
        out[f'lstm_{tag}_manifest'] = json.dumps(manifest_of(m))
        xs = [rnd((2, 16, 8, 10), 220 + t).requires_grad_(True) for t in range(3)]
        h0, c0 = rnd((2, 16, 8, 10), 230, 0.5).requires_grad_(True), rnd((2, 16, 8, 10), 231, 0.5).requires_grad_(True)
        for start in ('none', 'state'):
            for p in m.parameters():
                p.grad = None
            for t in xs + [h0, c0]:
                t.grad = None
            hc, hs = (None if start == 'none' else (h0, c0)), []
            for x in xs:
                hc = m(x, hc)
                hs.append(hc[0])
            loss = sum((h * rnd(h.shape, 240 + i)).sum() for i, h in enumerate(hs)) + (hc[1] * rnd(hc[1].shape, 250)).sum()
            loss.backward()
            pre = f'lstm_{tag}_{start}_'
            out[pre + 'h'] = torch.stack(hs)
            out[pre + 'c'] = hc[1]
            out[pre + 'dx'] = torch.stack([x.grad for x in xs])
            if start == 'state':
                out[pre + 'dh0'], out[pre + 'dc0'] = h0.grad, c0.grad
            for n, p in m.named_parameters():
                out[pre + 'grad_' + n.replace('.', '_')] = p.grad
    cfg = make_cfg(**MICRO, depthwise=True)
    cfg.fpn.depthwise = True
    det = YoloXDetector(cfg)
    load_synth(det, 22)
    out['det_manifest'] = json.dumps(manifest_of(det))
    feats = {2: rnd((3, 32, 8, 12), 51), 3: rnd((3, 64, 4, 6), 52), 4: rnd((3, 128, 2, 3), 53)}
    det.eval()
    with torch.no_grad():
        out['pred_eval'], _ = det.forward_detect(feats)
    labs = micro_labels(3, seed=7)
    targets = ObjectLabels.get_labels_as_batched_tensor([ObjectLabels(l, (60, 90)) for l in labs])
    det.train()
    for p in det.parameters():
        p.grad = None
    fin = {k: v.clone().requires_grad_(True) for k, v in feats.items()}
    pred_tr, losses = det.forward_detect(fin, targets=targets.clone())
    losses['loss'].backward()
    out.update(pred_train=pred_tr, targets=targets, **{f'loss_{k}': (v.detach() if torch.is_tensor(v) else v) for k, v in losses.items()})
    for k, v in fin.items():
        out[f'dfeat{k}'] = v.grad
    sd = det.state_dict()
    for k in ['fpn.bu_conv2.dconv.bn.running_mean', 'fpn.bu_conv2.pconv.bn.running_var', 'yolox_head.cls_convs.2.1.dconv.bn.running_var']:
        out['bn_' + k.replace('.', '_')] = sd[k]
    gk = sorted(n for n, p in det.named_parameters() if p.grad is not None)
    out['grad_keys'] = np.array(gk)
    params = dict(det.named_parameters())
    for n in gk:
        if '.dconv.' in n:                        # the depthwise filters and their BatchNorm: full gradients
            out['grad_' + n.replace('.', '_')] = params[n].grad
    out['grad_norms'] = np.array([float(params[k].grad.norm()) for k in gk], dtype=np.float64)
    save('g22_depthwise.npz', **out)


ALL['g22'] = g22_depthwise


DOWNSAMPLE_CASES = [('f4_patch', 4, 20, 16, False, True), ('f4_noaffine', 4, 20, 16, True, False), ('f2_patch_noaffine', 2, 16, 32, False, False),
                    ('f2_patch', 2, 16, 32, False, True), ('f4_patch_noaffine', 4, 20, 16, False, False)]


def g23_downsample_options():
    """``ConvDownsampling_Cf2Cl`` with the options the shipped configs leave at their defaults (maxvit.py:160-172): non-overlapping patches
    (``overlap=False``: kernel = stride, no padding) and a LayerNorm without affine parameters (``norm_affine=False``) -- output and gradients."""
    out = {}
    for tag, factor, cin, cout, overlap, affine in DOWNSAMPLE_CASES:
        m = ref_maxvit.ConvDownsampling_Cf2Cl(cin, cout, factor, DictConfig(dict(type='patch', overlap=overlap, norm_affine=affine)))
        load_synth(m, 23)
        out[tag + '_manifest'] = json.dumps(manifest_of(m))
        x = rnd((2, cin, 16, 24), 230 + factor).requires_grad_(True)
        y = m(x)
        (y * rnd(tuple(y.shape), 239)).sum().backward()
        out[tag + '_y'], out[tag + '_dx'] = y, x.grad
        for n, p in m.named_parameters():
            out[tag + '_grad_' + n.replace('.', '_')] = p.grad
    save('g23_downsample_options.npz', **out)


ALL['g23'] = g23_downsample_options


# (tag, window, skip_first_norm, attention-config overrides)
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


def g24_block_options():
    """``PartitionAttentionCl`` with the options no shipped config enables (maxvit.py:56-118,185-270,307-325): gated MLP, `mlp_activation`
    names, torch-MHA parameter layout, no LayerScale, no biases -- output, input gradient and every parameter gradient; and a
    ``RNNDetectorStage`` with token masking (maxvit_rnn.py:174-192)."""
    out = {}
    base = dict(make_cfg(32, 16, 0.33, (256, 320), (8, 10)).backbone.stage.attention)
    x0 = rnd((1, 16, 20, 32), 241)
    for tag, window, skip, over in BLOCK_OPTION_CASES:
        cfg = DictConfig({**base, **over})
        pt = ref_maxvit.PartitionType.WINDOW if window else ref_maxvit.PartitionType.GRID
        m = ref_maxvit.PartitionAttentionCl(dim=32, partition_type=pt, attention_cfg=cfg, skip_first_norm=skip)
        load_synth(m, 24)
        out[tag + '_manifest'] = json.dumps(manifest_of(m))
        x = x0.clone().requires_grad_(True)
        y = m(x)
        (y * rnd(tuple(y.shape), 242)).sum().backward()
        out[tag + '_y'], out[tag + '_dx'] = y, x.grad
        for n, p in m.named_parameters():
            out[tag + '_grad_' + n.replace('.', '_')] = p.grad
    # token masking: first stage of a backbone with enable_masking, two timesteps with carried state
    from models.detection.recurrent_backbone.maxvit_rnn import RNNDetectorStage
    scfg = make_cfg(16, 8, 0.33, (64, 96), (2, 3)).backbone.stage
    st = RNNDetectorStage(dim_in=20, stage_dim=16, spatial_downsample_factor=4, num_blocks=1, enable_token_masking=True,
                          T_max_chrono_init=4, stage_cfg=scfg)
    load_synth(st, 25)
    out['mask_manifest'] = json.dumps(manifest_of(st))
    xs = [rnd((2, 20, 64, 96), 251 + t) for t in range(2)]
    masks = [torch.rand((2, 16, 24), generator=torch.Generator().manual_seed(255 + t)) < 0.3 for t in range(2)]
    hc, hs = None, []
    for x, mk in zip(xs, masks):
        h, hc = st(x, hc, mk)
        hs.append(h)
    (sum((h * rnd(tuple(h.shape), 258 + i)).sum() for i, h in enumerate(hs)) + (hc[1] * rnd(tuple(hc[1].shape), 260)).sum()).backward()
    out['mask_h'], out['mask_c'] = torch.stack(hs), hc[1]
    out['mask_masks'] = torch.stack(masks)
    for n, p in st.named_parameters():
        out['mask_grad_' + n.replace('.', '_')] = p.grad
    save('g24_block_options.npz', **out)


ALL['g24'] = g24_block_options


def g25_mixed_density():
    """MixedDensityEventStack.construct: logarithmic time bins, int8 accumulation and channel running sum, +-cutoff"""
    rng = np.random.RandomState(125)
    out = {}
    for name, n, bins, cutoff in [('a', 20000, 10, None), ('b', 50000, 6, 5), ('c', 3000, 12, 0), ('e', 4096, 8, 127)]:
        H, W = 24, 30
        x = rng.randint(0, W, n)
        y = rng.randint(0, H, n)
        p = rng.randint(0, 2, n)
        t = np.sort(rng.randint(1000, 51000, n))
        if name == 'a':   # a hot pixel overflowing int8 in the accumulation and again in the running sum
            x[-700:], y[-700:], p[-700:] = 3, 4, 1
            x[-1500:-700], y[-1500:-700], p[-1500:-700] = 5, 6, 0
        if name == 'e':   # normalised times at and next to the exact bin edges 2^-k
            t = np.sort(np.concatenate([[0, 1 << 20]] + [[(1 << k) - 1, 1 << k, (1 << k) + 1] for k in range(1, 20)]
                                       + [rng.randint(0, 1 << 20, n - 2 - 57)]))
        rep = MixedDensityEventStack(bins=bins, height=H, width=W, count_cutoff=cutoff).construct(
            torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(p), torch.from_numpy(t))
        out.update({f'{name}_x': x, f'{name}_y': y, f'{name}_p': p, f'{name}_t': t, f'{name}_rep': rep.numpy()})
    # t0 == t1
    x, y, p, t = np.array([1, 2, 2]), np.array([0, 1, 1]), np.array([0, 1, 1]), np.array([5, 5, 5])
    rep = MixedDensityEventStack(bins=4, height=3, width=4).construct(*(torch.from_numpy(a) for a in (x, y, p, t)))
    out.update(d_rep=rep.numpy())
    save('g25_mixed_density.npz', **out)


ALL['g25'] = g25_mixed_density


if __name__ == '__main__':
    which = sys.argv[1:] or list(ALL)
    for w in which:
        ALL[w]()
