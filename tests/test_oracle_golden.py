"""Pin the oracle (oracle/) to the reference: every check here compares the CPU restatement with
vectors recorded by running the reference itself (tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import backbone as ob
from oracle import head as oh
from oracle import postproc as op
from oracle import train_step as ot
from oracle.synth import synth_state_dict, synth_events, synth_labels

RTOL, ATOL = 1e-5, 1e-6     # fp32 restatement vs reference


def G(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def close(a, b, rtol=RTOL, atol=ATOL):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rtol, atol=atol)


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def sub_sd(man, seed):
    return synth_state_dict(man, seed)


def test_g01_convlstm(golden_dir):
    g = G(golden_dir, 'g01_convlstm.npz')
    sd = synth_state_dict({'conv1x1.weight': (128, 64, 1, 1), 'conv1x1.bias': (128,)}, 1)
    sd = {'l.' + k: v for k, v in sd.items()}
    x = rnd((2, 32, 8, 10), 11)
    h0, c0 = rnd((2, 32, 8, 10), 12, 0.5), rnd((2, 32, 8, 10), 13, 0.5)
    # synth keys are relative to the module in make_golden ('conv1x1.weight'): re-key
    sd = {k.replace('l.', 'l.'): v for k, v in sd.items()}
    sd2 = {'l.conv1x1.weight': synth_state_dict({'conv1x1.weight': (128, 64, 1, 1)}, 1)['conv1x1.weight'],
           'l.conv1x1.bias': synth_state_dict({'conv1x1.bias': (128,)}, 1)['conv1x1.bias']}
    h, c = ob.conv_lstm(x, None, sd2, 'l')
    close(h, g['h_nostate']); close(c, g['c_nostate'])
    h, c = ob.conv_lstm(x, (h0, c0), sd2, 'l')
    close(h, g['h_state']); close(c, g['c_state'])


def test_g02_partition(golden_dir):
    g = G(golden_dir, 'g02_partition.npz')
    x = torch.arange(2 * 16 * 20, dtype=torch.float32).view(2, 16, 20, 1)
    wp, gp = ob.window_partition(x, (8, 10)), ob.grid_partition(x, (8, 10))
    assert np.array_equal(wp.long().numpy(), g['window'])
    assert np.array_equal(gp.long().numpy(), g['grid'])
    assert np.array_equal(ob.window_reverse(wp, (8, 10), (16, 20)).long().numpy(), g['window_rev'])
    assert np.array_equal(ob.grid_reverse(gp, (8, 10), (16, 20)).long().numpy(), g['grid_rev'])
    x2 = torch.arange(12 * 20, dtype=torch.float32).view(1, 12, 20, 1)
    assert np.array_equal(ob.window_partition(x2, (6, 10)).long().numpy(), g['window_6x10'])
    assert np.array_equal(ob.grid_partition(x2, (6, 10)).long().numpy(), g['grid_6x10'])


def _attn_manifest(C, with_norm1):
    m = {}
    if with_norm1:
        m.update({'norm1.weight': (C,), 'norm1.bias': (C,)})
    m.update({'self_attn.qkv.weight': (3 * C, C), 'self_attn.qkv.bias': (3 * C,),
              'self_attn.proj.weight': (C, C), 'self_attn.proj.bias': (C,), 'ls1.gamma': (C,),
              'norm2.weight': (C,), 'norm2.bias': (C,), 'mlp.net.0.0.weight': (4 * C, C),
              'mlp.net.0.0.bias': (4 * C,), 'mlp.net.2.weight': (C, 4 * C), 'mlp.net.2.bias': (C,),
              'ls2.gamma': (C,)})
    return m


def test_g03_attention(golden_dir):
    g = G(golden_dir, 'g03_attention.npz')
    x = rnd((2, 16, 20, 48), 31)
    for window, skip, key in [(True, False, 'blk_window_norm'), (True, True, 'blk_window_skip'),
                              (False, False, 'blk_grid_norm')]:
        sd = {'b.' + k: v for k, v in synth_state_dict(_attn_manifest(48, not skip), 3).items()}
        y = ob.partition_attention(x, sd, 'b', (8, 10), window, 24, skip_first_norm=skip)
        close(y, g[key], rtol=2e-5, atol=2e-6)
    sd = {'a.' + k: v for k, v in synth_state_dict(
        {'qkv.weight': (144, 48), 'qkv.bias': (144,), 'proj.weight': (48, 48), 'proj.bias': (48,)}, 4).items()}
    close(ob.self_attention(ob.window_partition(x, (8, 10)), sd, 'a', 24), g['self_attn'], rtol=2e-5, atol=2e-6)


MICRO = ot.model_cfg(embed_dim=16, dim_head=8, fpn_depth=0.33, partition_size=(2, 3), in_res_hw=(64, 96))


def test_g04_backbone(golden_dir, manifest):
    g = G(golden_dir, 'g04_backbone_micro.npz')
    sd = synth_state_dict(manifest['micro'], 5)
    ev = ob.pad_ev_repr(synth_events(3, 2, 20, 60, 90, seed=4, as_uint8=False), (64, 96))
    states = None
    for t in range(3):
        feats, states = ob.backbone_forward(ev[t], states, sd, MICRO)
        for k, v in feats.items():
            close(v, g[f't{t}_s{k}'], rtol=5e-5, atol=5e-6)
    for s, (h, c) in enumerate(states):
        close(c, g[f'final_c{s + 1}'], rtol=5e-5, atol=5e-6)
    g = G(golden_dir, 'g04_backbone_tiny256.npz')
    cfg = ot.model_cfg(32, 32, 0.33, (8, 10))
    sd = synth_state_dict(manifest['tiny_gen1'], 6)
    ev = ob.pad_ev_repr(synth_events(2, 1, 20, 240, 304, seed=5, as_uint8=False), (256, 320))
    feats, states = ob.backbone_forward(ev[0], None, sd, cfg)
    feats, states = ob.backbone_forward(ev[1], states, sd, cfg)
    for k, v in feats.items():
        close(v.mean(), g[f's{k}_mean'], rtol=1e-4, atol=1e-6)
        close(v.abs().max(), g[f's{k}_absmax'], rtol=1e-4)
        close(v[0, :8, :4, :5], g[f's{k}_slice'], rtol=1e-4, atol=1e-5)


def micro_labels(n_frames, seed, hw=(60, 90)):
    labs = synth_labels(n_frames, hw, 2, seed=seed, max_boxes=4)
    for l in labs:
        l[:, 3] = l[:, 3].clamp(max=30)
        l[:, 4] = l[:, 4].clamp(max=24)
        l[:, 1] = torch.minimum(l[:, 1], hw[1] - 1 - l[:, 3])
        l[:, 2] = torch.minimum(l[:, 2], hw[0] - 1 - l[:, 4])
    return labs


def test_g05_head(golden_dir, manifest):
    g = G(golden_dir, 'g05_head_micro.npz')
    sd = synth_state_dict(manifest['micro'], 5)
    feats = {2: rnd((3, 32, 8, 12), 51), 3: rnd((3, 64, 4, 6), 52), 4: rnd((3, 128, 2, 3), 53)}
    pred, losses = oh.detect_forward(feats, sd, MICRO, training=False)
    assert losses is None
    close(pred, g['pred_eval'], rtol=5e-5, atol=5e-6)
    labs = micro_labels(3, seed=7)
    labs[1] = labs[1][:1]
    labs[2][0, 1:5] = torch.tensor([0., 0., 12., 9.])
    targets = op.batched_yolox_labels(labs)
    close(targets, g['targets'])
    sd = {k: v.clone() for k, v in sd.items()}
    pkeys = [k for k, v in sd.items() if v.is_floating_point() and 'running_' not in k]
    for k in pkeys:
        sd[k].requires_grad_(True)
    pred, losses = oh.detect_forward(feats, sd, MICRO, labels=targets.clone(), training=True)
    close(pred, g['pred_train'], rtol=5e-5, atol=5e-6)
    for k in ('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss', 'num_fg'):
        close(torch.as_tensor(losses[k]).float(), g['loss_' + k], rtol=2e-5)
    for k in ['fpn.lateral_conv0.bn.running_mean', 'fpn.lateral_conv0.bn.running_var',
              'yolox_head.stems.0.bn.running_mean', 'yolox_head.cls_convs.2.1.bn.running_var']:
        close(sd[k], g['bn_' + k.replace('.', '_')], rtol=2e-5, atol=1e-6)
    losses['loss'].backward()
    gk = [str(k) for k in g['grad_keys']]
    mine = np.array([float(sd[k].grad.norm()) for k in gk])
    np.testing.assert_allclose(mine, g['grad_norms'], rtol=2e-4, atol=1e-7)


def test_g06_simota(golden_dir):
    g = G(golden_dir, 'g06_simota.npz')
    gx, gy, gs = oh.make_grids([(32, 40), (16, 20), (8, 10)], (8, 16, 32))
    for c in range(3):
        T = lambda k: torch.from_numpy(g[f'c{c}_{k}'])  # noqa
        gt, cls, bp, cl, ol = T('gt'), T('cls'), T('bp'), T('cls_l'), T('obj_l')
        n = gt.shape[0]
        inc = oh.is_in_centers(gt, gx, gy, gs)
        assert np.array_equal((inc.sum(0) > 0).numpy(), g[f'c{c}_geom_fg'])
        r = oh.get_assignments(gt, cls, torch.ones(n, dtype=torch.bool), bp, gx, gy, gs, cl, ol, 2)
        assert np.array_equal(r['fg_mask'].numpy(), g[f'c{c}_fg_mask'])
        assert np.array_equal(r['matched_gt_inds'].numpy(), g[f'c{c}_matched'])
        assert np.array_equal(r['gt_matched_classes'].numpy(), g[f'c{c}_mcls'])
        assert r['num_fg'] == int(g[f'c{c}_nfg'])
        close(r['pred_ious'], g[f'c{c}_pious'], rtol=1e-6)
        valid = T('valid')
        r = oh.get_assignments(gt, cls, valid, bp, gx, gy, gs, cl, ol, 2)
        assert np.array_equal(r['fg_mask'].numpy(), g[f'c{c}_ig_fg_mask'])
        assert np.array_equal(r['ignore_mask'].numpy(), g[f'c{c}_ig_ignore_mask'])
        assert np.array_equal(r['matched_gt_inds'].numpy(), g[f'c{c}_ig_matched'])
        assert np.array_equal(r['gt_matched_classes'].numpy(), g[f'c{c}_ig_mcls'])
        assert r['num_fg'] == int(g[f'c{c}_ig_nfg'])
        # two-pass equivalence (reference's commented oracle, yolo_head.py:1112-1116)
        af = oh.is_in_centers(gt, gx, gy, gs).sum(0) > 0
        afv = oh.is_in_centers(gt[valid], gx, gy, gs).sum(0) > 0
        assert torch.equal(r['ignore_mask'], af & ~afv)

    def run(targets, outputs, **kw):
        r = oh.get_losses(gx, gy, gs, targets.clone(), outputs.clone(), num_classes=2, **kw)
        return np.array([float(r[k]) for k in ('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss', 'num_fg')])

    tg, outp = torch.from_numpy(g['ign_targets']), torch.from_numpy(g['ign_outputs'])
    np.testing.assert_allclose(run(tg, outp), g['ign_losses'], rtol=2e-6)
    tg2 = tg.clone()
    tg2[:, :, 0] = torch.where(tg2[:, :, 0] == 1024, torch.zeros_like(tg2[:, :, 0]), tg2[:, :, 0])
    np.testing.assert_allclose(run(tg2, outp), g['noign_losses'], rtol=2e-6)
    np.testing.assert_allclose(run(tg2, outp, obj_focal_loss=True), g['focal_losses'], rtol=2e-6)
    np.testing.assert_allclose(run(torch.from_numpy(g['thr_targets']), outp, ignore_bbox_thresh=[0.7, 0.35]),
                               g['thr_losses'], rtol=2e-6)


HEAD_OPTION_CASES = dict(w_obj=dict(bbox_loss_weighting='obj'), w_cls=dict(bbox_loss_weighting='cls'), w_objxcls=dict(bbox_loss_weighting='objxcls'),
                         w_cls_sq=dict(bbox_loss_weighting='cls-w**2'), bg05=dict(ignore_bg_k=0.05), bg30=dict(ignore_bg_k=0.3),
                         w_obj_bg10=dict(bbox_loss_weighting='obj', ignore_bg_k=0.1), w_obj_focal=dict(bbox_loss_weighting='obj', obj_focal_loss=True))


@pytest.mark.parametrize('name', sorted(HEAD_OPTION_CASES))
def test_g19_head_loss_options(golden_dir, name):
    """bbox_loss_weighting / ignore_bg_k (yolo_head.py:335-381) against the reference's losses and gradients; the '_ign' batch holds an
    ignore box, which sends the reference through get_losses_w_ignore (weights apply, the top-k background step does not)."""
    g = G(golden_dir, 'g19_head_options.npz')
    gx, gy, gs = oh.make_grids([(32, 40), (16, 20), (8, 10)], (8, 16, 32))
    outp = torch.from_numpy(g['outputs'])
    for suffix, key in (('', 'targets'), ('_ign', 'targets_ign')):
        o = outp.clone().requires_grad_(True)
        r = oh.get_losses(gx, gy, gs, torch.from_numpy(g[key]).clone(), o, num_classes=2, **HEAD_OPTION_CASES[name])
        r['loss'].backward()
        got = np.array([float(r[k]) for k in ('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss', 'num_fg')])
        np.testing.assert_allclose(got, g[f'{name}{suffix}_losses'], rtol=2e-6)
        np.testing.assert_allclose(o.grad.numpy(), g[f'{name}{suffix}_grad'], rtol=1e-5, atol=1e-9)
    if 'bg' in name:                                        # the option does change the objectness term, and only without ignore boxes
        base = oh.get_losses(gx, gy, gs, torch.from_numpy(g['targets']).clone(), outp.clone(), num_classes=2)
        assert float(base['conf_loss']) > g[f'{name}_losses'][2] * 1.01


@pytest.mark.parametrize('name,nc,conf,agn', [
    ('rand_c0.1', 2, 0.1, False), ('rand_c0.01', 2, 0.01, False), ('rand_c0.001', 2, 0.001, False),
    ('rand_agnostic', 2, 0.1, True), ('adv_c0.1', 3, 0.1, False), ('adv_c0.001', 3, 0.001, False),
    ('many_c0.001', 2, 0.001, False), ('none', 2, 0.5, False)])
def test_g07_postprocess(golden_dir, name, nc, conf, agn):
    g = G(golden_dir, 'g07_postprocess.npz')
    pred = torch.from_numpy(g[name + '_pred']).clone()
    res = op.postprocess(pred, nc, conf, 0.45, class_agnostic=agn, pad=torch.zeros((0, 7)),
                         device_semantics='cpu')
    assert [len(r) for r in res] == list(g[name + '_n'])
    det = torch.cat(res, 0).numpy()
    assert np.array_equal(det, g[name + '_det'])          # bit-exact incl. order
    # in-place xyxy side effect (boxes.py:41-46)
    orig = torch.from_numpy(g[name + '_pred'])
    assert torch.equal(pred[..., 0], orig[..., 0] - orig[..., 2] / 2)


def test_g08_pseudo(golden_dir):
    g = G(golden_dir, 'g08_pseudo.npz')
    allp = torch.from_numpy(g['p2l_in'])
    preds = list(torch.split(allp, list(g['p2l_lens_in'])))
    labs = op.pred2label([p.clone() for p in preds], [0.6, 0.3], [0.6, 0.3], 'gen1', False)
    assert [len(l) for l in labs] == list(g['p2l_lens'])
    assert np.array_equal(torch.cat(labs).numpy(), g['p2l_out'])
    preds4 = list(torch.split(torch.from_numpy(g['p2l4_in']), list(g['p2l_lens_in'])))
    labs = op.pred2label(preds4, [0.3, 0.3, 0.6], [0.3, 0.3, 0.6], 'gen4', True)
    assert [len(l) for l in labs] == list(g['p2l4_lens'])
    assert np.array_equal(torch.cat(labs).numpy(), g['p2l4_out'])
    labs = op.pred2label([p.clone() for p in preds], 0.5, 0.4, filter_boxes=False)
    assert [len(l) for l in labs] == list(g['p2lf_lens'])
    assert np.array_equal(torch.cat(labs).numpy(), g['p2lf_out'])
    views = [torch.from_numpy(g['tta_in0']), torch.from_numpy(g['tta_in1']), torch.zeros((0, 7))]
    res = op.tta_postprocess(views, 0.01, 0.45, pad=torch.zeros((0, 7)), device_semantics='cpu')
    assert [len(r) for r in res] == list(g['tta_n'])
    assert np.array_equal(torch.cat(res).numpy(), g['tta_out'])
    lab = torch.from_numpy(g['lab_in'])
    assert np.array_equal(op.labels_to_yolox(lab).numpy(), g['lab_yolox'])
    assert np.array_equal(op.flip_lr_labels(lab, 304).numpy(), g['lab_flip'])
    assert list(op.get_subsample_label_idx(21, use_every=1)) == list(g['subsample_21_1'])
    assert list(op.get_subsample_label_idx(21, use_every=5)) == list(g['subsample_21_5'])
    assert sorted(op.get_subsample_label_idx(10, remove_every=3)) == list(g['subsample_10_r3'])


def test_g10_voxel(golden_dir):
    g = G(golden_dir, 'g10_voxel.npz')
    for name, fast, cutoff in [('a', True, None), ('b', False, 10), ('c', True, 3)]:
        rep = op.stacked_histogram(g[f'{name}_x'], g[f'{name}_y'], g[f'{name}_p'], g[f'{name}_t'], 10, 24, 30,
                                   count_cutoff=cutoff, fastmode=fast)
        assert rep.dtype == np.uint8 and np.array_equal(rep, g[f'{name}_rep'])
    rep = op.stacked_histogram(np.array([1, 2, 2]), np.array([0, 1, 1]), np.array([0, 1, 1]), np.array([5, 5, 5]), 4, 3, 4)
    assert np.array_equal(rep, g['d_rep'])


def test_g25_mixed_density(golden_dir):
    g = G(golden_dir, 'g25_mixed_density.npz')
    for name, bins, cutoff in [('a', 10, None), ('b', 6, 5), ('c', 12, 0), ('e', 8, 127)]:
        rep = op.mixed_density_stack(g[f'{name}_x'], g[f'{name}_y'], g[f'{name}_p'], g[f'{name}_t'], bins, 24, 30, count_cutoff=cutoff)
        assert rep.dtype == np.int8 and np.array_equal(rep, g[f'{name}_rep']), name
    rep = op.mixed_density_stack(np.array([1, 2, 2]), np.array([0, 1, 1]), np.array([0, 1, 1]), np.array([5, 5, 5]), 4, 3, 4)
    assert np.array_equal(rep, g['d_rep'])
    assert not op.mixed_density_stack(np.zeros(0, int), np.zeros(0, int), np.zeros(0, int), np.zeros(0, int), 3, 4, 5).any()


def test_g11_onecycle(golden_dir):
    want = json.load(open(os.path.join(golden_dir, 'g11_onecycle.json')))
    from oracle.schedule import one_cycle_lr
    for k, v in want.items():
        assert abs(one_cycle_lr(int(k), 2e-4, 400000, 0.005, 20, 10000) - v) <= 1e-12 + 1e-9 * v


def test_g12_trainstep(golden_dir, manifest):
    g = G(golden_dir, 'g12_trainstep_micro.npz')
    sd = synth_state_dict(manifest['micro'], 9)
    tr = ot.OracleTrainer(sd, MICRO, lr=2e-4, total_steps=1000, div_factor=20, final_div_factor=10000)
    T, B = 5, 2
    for step in range(2):
        ev = synth_events(T, B, 20, 60, 90, seed=20 + step, as_uint8=True)
        lab_list = micro_labels(T * B, seed=30 + step)
        labels = [[lab_list[t * B + b] if (t in (2, 4) or (t == 1 and b == 0)) else None for b in range(B)]
                  for t in range(T)]
        is_first = torch.tensor([True, True]) if step == 0 else torch.tensor([False, True])
        losses, grads = tr.step(ev, labels, is_first)
        want = g[f's{step}_losses']
        got = np.array([losses[k] for k in ('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss', 'num_fg')])
        np.testing.assert_allclose(got, want, rtol=5e-5)
        keys = [str(k) for k in g[f's{step}_grad_keys']]
        np.testing.assert_allclose(np.array([float(grads[k].norm()) for k in keys]),
                                   g[f's{step}_grad_norms'], rtol=2e-3, atol=1e-7)
        np.testing.assert_allclose(np.array([float(tr.sd[k].detach().norm()) for k in keys]),
                                   g[f's{step}_param_norms'], rtol=2e-4)  # Adam's first steps ~ lr*sign(g): noise-level grads flip
        assert abs(tr.opt.param_groups[0]['lr'] - float(g[f's{step}_lr_next'])) < 1e-12
        close(tr.states[3][1], g[f's{step}_state_c4'], rtol=1e-4, atol=1e-5)


def test_g20_trajectory_first_steps(golden_dir, manifest):
    """The first 16 optimisation steps of the reference's recorded 200-step fp32 trajectory (g20: OneCycle over 200 steps with pct_start 0.1,
    16 cycled batches, one streaming and one restarting sample) through OracleTrainer: AdamW, the schedule, gradient clipping and the carried
    LSTM state restated over many steps, not two.  Training from random init amplifies the different rounding order of the two
    implementations: measured 1e-7 for five steps, 1e-6 to step 12, 2e-5 at step 15, 3e-3 at step 17 and 4.6 % at step 21 -- which is
    why the later part of a trajectory can only be compared as a distribution (the -m gpu test of the same fixture)."""
    g = G(golden_dir, 'g20_trajectory_micro.npz')
    sd = synth_state_dict(manifest['micro'], 9)
    tr = ot.OracleTrainer(sd, MICRO, lr=2e-4, total_steps=200, pct_start=0.1, div_factor=20, final_div_factor=10000)
    T, B = 5, 2
    got = []
    for step in range(16):
        i = step % 16
        ev = synth_events(T, B, 20, 60, 90, seed=700 + i, as_uint8=True)
        lab_list = micro_labels(T * B, seed=800 + i)
        labels = [[lab_list[t * B + b] if (t in (2, 4) or (t == 1 and b == 0)) else None for b in range(B)] for t in range(T)]
        losses, _ = tr.step(ev, labels, torch.tensor([step == 0, True]))
        got.append(losses['loss'])
    got, want = np.array(got), g['fp32'][:16]
    rel = np.abs(got - want) / want
    print('relative difference per step:', np.round(rel, 6))
    assert rel[:5].max() < 2e-6 and rel[:12].max() < 5e-5 and rel.max() < 2e-3, rel


def test_g18_fp32_side_of_the_autocast_fixture(golden_dir, manifest):
    """g18_autocast.npz records how far the REFERENCE's autocast runs land from its fp32 run; the -m gpu tests measure this
    build's bf16 mode against this build's fp32 mode on the same workloads.  The two fp32 sides must be the same function:
    the oracle reproduces the fixture's fp32 losses and per-parameter gradient norms -- micro step and the benchmark workload
    (RVT-S Gen1 240x304 T=21 bs=8, bench.make_batch seed 7; one full-size oracle step, ~10 s)."""
    import bench
    g = G(golden_dir, 'g18_autocast.npz')
    keys = ('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss', 'num_fg')
    # micro (the g12 set-up, step 0, un-clipped gradients)
    tr = ot.OracleTrainer(synth_state_dict(manifest['micro'], 9), MICRO, lr=2e-4, total_steps=1000, clip_value=0.0)
    T, B = 5, 2
    ev = synth_events(T, B, 20, 60, 90, seed=20, as_uint8=True)
    lab_list = micro_labels(T * B, seed=30)
    labels = [[lab_list[t * B + b] if (t in (2, 4) or (t == 1 and b == 0)) else None for b in range(B)] for t in range(T)]
    losses, grads = tr.step(ev, labels, torch.ones(B, dtype=torch.bool))
    np.testing.assert_allclose([losses[k] for k in keys], g['micro_fp32_losses'], rtol=5e-5)
    names = [str(k) for k in g['micro_grad_keys']]
    np.testing.assert_allclose([float(grads[k].double().norm()) for k in names], g['micro_fp32_grad_norms'], rtol=2e-3, atol=1e-7)
    # the benchmark workload
    T, B, hw = 21, 8, (240, 304)
    ev, _, label_tb, labs = bench.make_batch(T, B, hw, 2, 7, 'cpu', (4, 9, 14, 19))
    it = iter(labs)
    labels = []
    for t in range(T):
        row = [None] * B
        for b in label_tb[t]:
            l = next(it)
            row[b] = torch.from_numpy(np.concatenate([np.ones((len(l), 1), np.float32), l[:, 1:2] - l[:, 3:4] / 2, l[:, 2:3] - l[:, 4:5] / 2,
                                                      l[:, 3:5], l[:, 0:1], l[:, 6:7], l[:, 5:6]], 1))
        labels.append(row)
    tr = ot.OracleTrainer(synth_state_dict(manifest['small_gen1'], 0), ot.model_cfg(48, 24, 0.33, (8, 10)), clip_value=0.0)
    losses, grads = tr.step(ev, labels, torch.ones(B, dtype=torch.bool))
    np.testing.assert_allclose([losses[k] for k in keys], g['small_fp32_losses'], rtol=2e-5, atol=1e-6)
    names = [str(k) for k in g['small_grad_keys']]
    np.testing.assert_allclose([float(grads[k].double().norm()) for k in names], g['small_fp32_grad_norms'], rtol=2e-3, atol=1e-7)
    # sanity of the class itself: the looser autocast run of the reference sits at a gradient cosine of ~0.92 from its fp32 run
    assert 0.85 < float(g['small_ac_grad_cos_global']) < float(g['small_acf_grad_cos_global']) < 0.99


def _tracker_cases(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g13_tracker.npz'))
    for si in range(6):
        fi, cnt, rows_all = g[f's{si}_frame_idx'], g[f's{si}_counts'], g[f's{si}_rows']
        hw = tuple(int(v) for v in g[f's{si}_hw'])
        off = np.concatenate([[0], np.cumsum(cnt)])
        rows = [rows_all[off[k]:off[k + 1]] for k in range(len(cnt))]
        for method, mt in (('forward', 'f'), ('forward or backward', 'fb')):
            for inpaint, it in ((False, 'noinp'), (True, 'inp')):
                tag = f's{si}_{mt}_{it}'
                yield rows, fi, hw, method, inpaint, g[tag + '_frame_idx'], g[tag + '_counts'], g[tag + '_rows']


def test_tracker_oracle_matches_reference(golden_dir):
    """oracle.tracker == the reference's EventSeqData._track_filter (modules/pseudo_labeler.py:201-333), bit for bit:
    ignore labels, in-painted boxes, inserted frames."""
    from oracle import tracker as ot
    n = 0
    for rows, fi, hw, method, inpaint, exp_f, exp_c, exp_rows in _tracker_cases(golden_dir):
        f2, r2 = ot.apply_track_filter(rows, fi, hw, 6, method, inpaint, ignore_label=1024)
        assert list(f2) == list(exp_f)
        assert [len(r) for r in r2] == list(exp_c)
        np.testing.assert_array_equal(np.concatenate(r2, 0), exp_rows)
        n += 1
    assert n == 24


def test_augment_oracle_matches_reference(golden_dir):
    """oracle.augment (flip / zoom-in / zoom-out with ATen's nearest-exact index rule) == the tensors the reference's
    RandomSpatialAugmentorGenX produced for the recorded augmentation states (data/utils/augmentor.py:229-331,396-401)."""
    from oracle import augment as oa
    from oracle.synth import synth_augment_sample, AUGMENT_CASES
    g = np.load(os.path.join(golden_dir, 'g14_augment.npz'))
    modes = set()
    for seed, H, W in AUGMENT_CASES:
        ev, _ = synth_augment_sample(seed, H, W)
        st = g[f's{seed}_state']
        mode = 1 if st[1] else (2 if st[5] else 0)
        x0, y0, f = (int(st[2]), int(st[3]), float(st[4])) if mode == 1 else (int(st[6]), int(st[7]), float(st[8]))
        out = oa.apply(torch.stack(ev).numpy(), bool(st[0]), mode, x0, y0, f)
        np.testing.assert_array_equal(out, g[f's{seed}_ev'])
        modes.add((bool(st[0]), mode))
    assert {m for _, m in modes} == {0, 1, 2} and {f for f, _ in modes} == {False, True}


def test_tta_result_oracle_matches_reference(golden_dir):
    """oracle.tta.aggregate_views == the reference's EventSeqResult (modules/utils/tta.py:64-195) on scripted views."""
    from oracle import tta as otta
    from oracle.synth import synth_tta_views
    g = np.load(os.path.join(golden_dir, 'g16_tta_result.npz'))
    for case in range(3):
        views, hw = synth_tta_views(case)
        got = otta.aggregate_views(views, hw, 0.1, 0.45)
        assert len(got) == int(g[f'c{case}_n']) == 5
        for k, (lab, prd) in enumerate(got):
            for name in lab.dtype.names:
                assert np.array_equal(lab[name], g[f'c{case}_lab{k}_{name}']), (case, k, name)
                assert np.array_equal(prd[name], g[f'c{case}_pred{k}_{name}']), (case, k, name)


def _g22_lstm_inputs():
    xs = [rnd((2, 16, 8, 10), 220 + t) for t in range(3)]
    return xs, rnd((2, 16, 8, 10), 230, 0.5), rnd((2, 16, 8, 10), 231, 0.5)


@pytest.mark.parametrize('tag', ['h', 'xh', 'h5'])
def test_g22_depthwise_convlstm(golden_dir, tag):
    """``DWSConvLSTM2d(dws_conv=True)`` (rnn.py:20-30,50-55; off in the shipped configs): the oracle's depthwise placement against three
    chained timesteps of the reference, outputs and every gradient."""
    g = G(golden_dir, 'g22_depthwise.npz')
    man = json.loads(str(g[f'lstm_{tag}_manifest']))
    for start in ('none', 'state'):
        sd = {'lstm.' + k: v.clone().requires_grad_(True) for k, v in synth_state_dict(man, 21).items()}
        xs, h0, c0 = _g22_lstm_inputs()
        for t in xs + [h0, c0]:
            t.requires_grad_(True)
        hc, hs = (None if start == 'none' else (h0, c0)), []
        for x in xs:
            hc = ob.conv_lstm(x, hc, sd, 'lstm')
            hs.append(hc[0])
        loss = sum((h * rnd(h.shape, 240 + i)).sum() for i, h in enumerate(hs)) + (hc[1] * rnd(hc[1].shape, 250)).sum()
        loss.backward()
        pre = f'lstm_{tag}_{start}_'
        close(torch.stack(hs), g[pre + 'h'])
        close(hc[1], g[pre + 'c'])
        close(torch.stack([x.grad for x in xs]), g[pre + 'dx'], rtol=5e-5, atol=1e-6)
        if start == 'state':
            close(h0.grad, g[pre + 'dh0'], rtol=5e-5, atol=1e-6)
            close(c0.grad, g[pre + 'dc0'], rtol=5e-5, atol=1e-6)
        for k in man:
            close(sd['lstm.' + k].grad, g[pre + 'grad_' + k.replace('.', '_')], rtol=1e-4, atol=2e-5)


def test_g22_depthwise_head(golden_dir):
    """PAFPN + head with ``depthwise=True`` (DWConv, network_blocks.py:57-76; yolo_pafpn.py:37, yolo_head.py:52): the oracle reads the
    variant off the state-dict keys (``*.dconv.* / *.pconv.*``)."""
    g = G(golden_dir, 'g22_depthwise.npz')
    man = json.loads(str(g['det_manifest']))
    sd = synth_state_dict(man, 22)
    feats = {2: rnd((3, 32, 8, 12), 51), 3: rnd((3, 64, 4, 6), 52), 4: rnd((3, 128, 2, 3), 53)}
    pred, losses = oh.detect_forward(feats, sd, MICRO, training=False)
    assert losses is None
    close(pred, g['pred_eval'], rtol=5e-5, atol=5e-6)
    targets = op.batched_yolox_labels(micro_labels(3, seed=7))
    close(targets, g['targets'])
    sd = {k: v.clone() for k, v in sd.items()}
    for k, v in sd.items():
        if v.is_floating_point() and 'running_' not in k:
            v.requires_grad_(True)
    fin = {k: v.clone().requires_grad_(True) for k, v in feats.items()}
    pred, losses = oh.detect_forward(fin, sd, MICRO, labels=targets.clone(), training=True)
    close(pred, g['pred_train'], rtol=5e-5, atol=5e-6)
    for k in ('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss', 'num_fg'):
        close(torch.as_tensor(losses[k]).float(), g['loss_' + k], rtol=2e-5)
    for k in ['fpn.bu_conv2.dconv.bn.running_mean', 'fpn.bu_conv2.pconv.bn.running_var', 'yolox_head.cls_convs.2.1.dconv.bn.running_var']:
        close(sd[k], g['bn_' + k.replace('.', '_')], rtol=2e-5, atol=1e-6)
    losses['loss'].backward()
    for k in fin:
        close(fin[k].grad, g[f'dfeat{k}'], rtol=2e-4, atol=1e-6)
    gk = [str(k) for k in g['grad_keys']]
    np.testing.assert_allclose(np.array([float(sd[k].grad.norm()) for k in gk]), g['grad_norms'], rtol=2e-4, atol=1e-7)
    for k in gk:
        if '.dconv.' in k:
            close(sd[k].grad, g['grad_' + k.replace('.', '_')], rtol=2e-4, atol=2e-6)


DOWNSAMPLE_CASES = [('f4_patch', 4, 20, 16), ('f4_noaffine', 4, 20, 16), ('f2_patch_noaffine', 2, 16, 32), ('f2_patch', 2, 16, 32),
                    ('f4_patch_noaffine', 4, 20, 16)]


@pytest.mark.parametrize('tag,factor,cin,cout', DOWNSAMPLE_CASES)
def test_g23_downsample_options(golden_dir, tag, factor, cin, cout):
    """``ConvDownsampling_Cf2Cl`` with ``overlap=False`` / ``norm_affine=False`` (maxvit.py:160-172): the oracle reads both off the state dict."""
    g = G(golden_dir, 'g23_downsample_options.npz')
    man = json.loads(str(g[tag + '_manifest']))
    sd = {'d.' + k: v.clone().requires_grad_(True) for k, v in synth_state_dict(man, 23).items()}
    x = rnd((2, cin, 16, 24), 230 + factor).requires_grad_(True)
    y = ob.conv_downsample(x, sd, 'd', factor)
    (y * rnd(tuple(y.shape), 239)).sum().backward()
    close(y, g[tag + '_y'], rtol=2e-5, atol=2e-6)
    close(x.grad, g[tag + '_dx'], rtol=1e-4, atol=1e-5)
    for k in man:
        close(sd['d.' + k].grad, g[tag + '_grad_' + k.replace('.', '_')], rtol=1e-4, atol=2e-5)


BLOCK_OPTION_CASES = [('geglu', True, False, 'gelu'), ('swiglu_grid', False, False, 'swish'), ('reglu_skipnorm', True, True, 'relu'),
                      ('glu_sigmoid_mha', True, False, 'sigmoid'), ('mha_grid', False, False, 'gelu'), ('relu_plain', True, False, 'relu'),
                      ('mish_nols_nobias', False, False, 'mish'), ('hswish_tanhglu', True, False, 'hard_swish'), ('tanh_glu_nols', True, False, 'tanh'),
                      ('elu', False, False, 'elu'), ('selu', True, False, 'selu'), ('hsig_glu', True, False, 'hard_sigmoid'),
                      ('relu6', True, False, 'relu6'), ('leaky', True, False, 'leaky_relu'), ('celu', True, False, 'celu'),
                      ('hmish', True, False, 'hard_mish'), ('silu', True, False, 'silu'), ('mha_nobias', True, False, 'gelu')]


@pytest.mark.parametrize('tag,window,skip,act', BLOCK_OPTION_CASES)
def test_g24_block_options(golden_dir, tag, window, skip, act):
    """``PartitionAttentionCl`` with the options no shipped config enables (gated MLP, `mlp_activation`, torch-MHA layout, no LayerScale,
    no biases: maxvit.py:56-118,185-270,307-325): the oracle reads the structure off the state dict, the activation by name."""
    g = G(golden_dir, 'g24_block_options.npz')
    man = json.loads(str(g[tag + '_manifest']))
    sd = {'b.' + k: v.clone().requires_grad_(True) for k, v in synth_state_dict(man, 24).items()}
    x = rnd((1, 16, 20, 32), 241).requires_grad_(True)
    y = ob.partition_attention(x, sd, 'b', (8, 10), window, 16, skip_first_norm=skip, act=act)
    (y * rnd(tuple(y.shape), 242)).sum().backward()
    close(y, g[tag + '_y'], rtol=2e-5, atol=2e-6)
    close(x.grad, g[tag + '_dx'], rtol=1e-4, atol=1e-5)
    for k in man:
        ref = g[tag + '_grad_' + k.replace('.', '_')]
        close(sd['b.' + k].grad, ref, rtol=2e-4, atol=2e-5 * max(1.0, float(np.abs(ref).max())))


def test_g24_token_masking(golden_dir):
    """``RNNDetectorStage`` with ``enable_token_masking`` (maxvit_rnn.py:174-192): two timesteps with carried state, every gradient."""
    g = G(golden_dir, 'g24_block_options.npz')
    man = json.loads(str(g['mask_manifest']))
    sd = {'s.' + k: v.clone().requires_grad_(True) for k, v in synth_state_dict(man, 25).items()}
    masks = torch.from_numpy(g['mask_masks'])
    hc, hs = None, []
    for t in range(2):
        h, hc = ob.stage_forward(rnd((2, 20, 64, 96), 251 + t), hc, sd, 's', 4, (2, 3), 8, token_mask=masks[t])
        hs.append(h)
    (sum((h * rnd(tuple(h.shape), 258 + i)).sum() for i, h in enumerate(hs)) + (hc[1] * rnd(tuple(hc[1].shape), 260)).sum()).backward()
    close(torch.stack(hs), g['mask_h'], rtol=2e-5, atol=2e-6)
    close(hc[1], g['mask_c'], rtol=2e-5, atol=2e-6)
    for k in man:
        ref = g['mask_grad_' + k.replace('.', '_')]
        close(sd['s.' + k].grad, ref, rtol=3e-4, atol=3e-5 * max(1.0, float(np.abs(ref).max())))
