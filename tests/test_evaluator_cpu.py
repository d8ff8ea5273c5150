"""Prophesee / COCO detection evaluation: oracle vs the reference-recorded fixture, the native
evaluator (leod_coco_eval, host C++ -- no GPU involved) vs the oracle, and hand-checkable known answers.

Pinned against the reference: box filters, +-50 ms time matching, COCO record conversion, to_prophesee
(tests/golden/g15_evaluator.npz).  Unpinned: the AP computation itself (pycocotools is absent, oracle/coco_eval.py)."""
import os
import warnings

import numpy as np
import pytest
import torch

from oracle import coco_eval as oc
from oracle.synth import synth_eval_sequences, EVAL_CASES, EVAL_BBOX_DTYPE


def _filtered(case):
    labels, dets = synth_eval_sequences(**case)
    diag, side = 30, 10                      # Gen1, and Gen4 downsampled by 2 (60/2, 20/2)
    return labels, dets, [oc.filter_boxes(g, int(5e5), diag, side) for g in labels], \
        [oc.filter_boxes(d, int(5e5), diag, side) for d in dets]


@pytest.mark.parametrize('ci', range(len(EVAL_CASES)))
def test_oracle_pipeline_matches_reference(golden_dir, ci):
    """filter_boxes -> _match_times -> _to_coco_format of the reference (box_filtering.py:18-36, coco_eval.py:49-97,143-194)."""
    g = np.load(os.path.join(golden_dir, 'g15_evaluator.npz'))
    labels, dets, gf, df = _filtered(EVAL_CASES[ci])
    gts, dts = [], []
    for k in range(len(labels)):
        assert len(gf[k]) == int(g[f'c{ci}_s{k}_n_gt']) and len(df[k]) == int(g[f'c{ci}_s{k}_n_dt'])
        assert np.array_equal(labels[k][g[f'c{ci}_s{k}_gt_kept']], gf[k])
        assert np.array_equal(dets[k][g[f'c{ci}_s{k}_dt_kept']], df[k])
        gw, dw = oc.match_times(gf[k], df[k], 50000)
        gts += gw
        dts += dw
    assert [len(x) for x in gts] == list(g[f'c{ci}_gt_cnt']) and [len(x) for x in dts] == list(g[f'c{ci}_dt_cnt'])
    assert len(gts) == int(g[f'c{ci}_n_img'])
    anns, res = oc.to_coco_records(gts, dts)
    assert np.array_equal(np.array([a['area'] for a in anns]), g[f'c{ci}_ann_area'])
    assert np.array_equal(np.array([a['bbox'] for a in anns]).reshape(-1, 4), g[f'c{ci}_ann_bbox'])
    assert [a['category_id'] for a in anns] == list(g[f'c{ci}_ann_cat']) and [a['image_id'] for a in anns] == list(g[f'c{ci}_ann_img'])
    assert np.array_equal(np.array([r['score'] for r in res]), g[f'c{ci}_res_score'])
    assert np.array_equal(np.array([r['bbox'] for r in res]).reshape(-1, 4), g[f'c{ci}_res_bbox'])
    assert [r['category_id'] for r in res] == list(g[f'c{ci}_res_cat']) and [r['image_id'] for r in res] == list(g[f'c{ci}_res_img'])


@pytest.mark.parametrize('ci', range(len(EVAL_CASES)))
def test_product_filter_and_windows(golden_dir, ci):
    """The product's vectorised filter / searchsorted windows == the reference's loops."""
    from leod_amd.utils.evaluation.prophesee.io.box_filtering import filter_boxes
    from leod_amd.utils.evaluation.prophesee.metrics.coco_eval import match_times
    g = np.load(os.path.join(golden_dir, 'g15_evaluator.npz'))
    labels, dets, _, _ = _filtered(EVAL_CASES[ci])
    g_cnt, d_cnt, ann_bbox, res_bbox = [], [], [], []
    for k in range(len(labels)):
        gf, df = filter_boxes(labels[k], int(5e5), 30, 10), filter_boxes(dets[k], int(5e5), 30, 10)
        assert np.array_equal(labels[k][g[f'c{ci}_s{k}_gt_kept']], gf) and np.array_equal(dets[k][g[f'c{ci}_s{k}_dt_kept']], df)
        g_idx, g_off, d_idx, d_off = match_times(gf, df, 50000)
        g_cnt += list(np.diff(g_off))
        d_cnt += list(np.diff(d_off))
        ann_bbox.append(np.stack([gf[n][g_idx] for n in 'xywh'], 1).astype(np.float64).reshape(-1, 4))
        res_bbox.append(np.stack([df[n][d_idx] for n in 'xywh'], 1).astype(np.float64).reshape(-1, 4))
    assert g_cnt == list(g[f'c{ci}_gt_cnt']) and d_cnt == list(g[f'c{ci}_dt_cnt'])
    assert np.array_equal(np.concatenate(ann_bbox), g[f'c{ci}_ann_bbox'])
    assert np.array_equal(np.concatenate(res_bbox), g[f'c{ci}_res_bbox'])


def test_to_prophesee_matches_reference(golden_dir):
    from leod_amd.data.genx_utils.labels import ObjectLabels
    from leod_amd.utils.evaluation.prophesee.io.box_loading import to_prophesee, BBOX_DTYPE, reformat_boxes
    assert BBOX_DTYPE == EVAL_BBOX_DTYPE
    g = np.load(os.path.join(golden_dir, 'g15_evaluator.npz'))
    gen = torch.Generator().manual_seed(15)
    labs, preds = [], []
    for f in range(4):
        n = 1 + f
        l = torch.rand((n, 8), generator=gen) * 50
        l[:, 0] = 1000000 + 50000 * f
        l[:, 5] = torch.randint(0, 2, (n,), generator=gen).float()
        labs.append(ObjectLabels(l, (240, 304)))
        m = [3, 0, 2, 1][f]
        p = torch.rand((m, 7), generator=gen) * 100
        p[:, 2:4] += p[:, 0:2]
        p[:, 6] = torch.randint(0, 2, (m,), generator=gen).float()
        preds.append(p if m else None)
    lp, pp = to_prophesee(labs, preds)
    for f in range(4):
        for name in BBOX_DTYPE.names:
            for got, key in ((lp[f], f'proph_lab{f}_{name}'), (pp[f], f'proph_pred{f}_{name}')):
                assert got[name].dtype == g[key].dtype and np.array_equal(got[name], g[key]), key
    old = np.zeros((2,), dtype=[('ts', '<i8'), ('x', '<f4'), ('y', '<f4'), ('w', '<f4'), ('h', '<f4'), ('class_id', '<u4'),
                                ('confidence', '<f4')])
    old['ts'], old['confidence'] = [5, 6], [0.5, 0.25]
    new = reformat_boxes(old)
    assert new.dtype == BBOX_DTYPE and list(new['t']) == [5, 6] and list(new['class_confidence']) == [0.5, 0.25]


def _native_tables(gts, dts, n_cat):
    from leod_amd.utils.evaluation.prophesee.metrics.coco_eval import coco_tables
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros((0,), EVAL_BBOX_DTYPE)  # noqa: E731
    off = lambda xs: np.concatenate([[0], np.cumsum([len(x) for x in xs])]).astype(np.int64)  # noqa: E731
    return coco_tables(cat(gts), off(gts), cat(dts), off(dts), n_cat)


@pytest.mark.parametrize('ci', range(len(EVAL_CASES)))
def test_native_tables_equal_oracle(ci):
    """leod_coco_eval == the loop restatement of COCOeval, every entry of precision[T,R,K,A,M] / recall[T,K,A,M] bit for bit."""
    case = EVAL_CASES[ci]
    _, _, gf, df = _filtered(case)
    gts, dts = [], []
    for g, d in zip(gf, df):
        gw, dw = oc.match_times(g, d, 50000)
        gts += gw
        dts += dw
    n_cat = case.get('n_cls', 2)
    anns, res = oc.to_coco_records(gts, dts)
    p_ref, r_ref = oc.coco_tables(anns, res, len(gts), n_cat)
    p, r = _native_tables(gts, dts, n_cat)
    assert np.array_equal(p, p_ref) and np.array_equal(r, r_ref)
    assert (p_ref > -1).any()


def test_native_tables_crowded_images():
    """More than 100 detections per image and category (the maxDets cut), heavy score ties, images without labels,
    labels without detections, a class that never occurs."""
    rng = np.random.RandomState(5)
    gts, dts = [], []
    for img in range(6):
        ng = [8, 0, 3, 12, 1, 5][img]
        nd = [150, 40, 0, 260, 7, 120][img]
        g = np.zeros((ng,), EVAL_BBOX_DTYPE)
        g['w'], g['h'] = rng.uniform(5, 150, ng), rng.uniform(5, 120, ng)
        g['x'], g['y'] = rng.uniform(0, 150, ng), rng.uniform(0, 100, ng)
        g['class_id'] = rng.randint(0, 2, ng)
        d = np.zeros((nd,), EVAL_BBOX_DTYPE)
        src = rng.randint(0, max(ng, 1), nd)
        if ng:
            d['x'], d['y'] = g['x'][src] + rng.uniform(-6, 6, nd), g['y'][src] + rng.uniform(-6, 6, nd)
            d['w'], d['h'] = g['w'][src] * rng.uniform(0.8, 1.2, nd), g['h'][src] * rng.uniform(0.8, 1.2, nd)
        else:
            d['x'], d['y'], d['w'], d['h'] = rng.uniform(0, 200, nd), rng.uniform(0, 150, nd), rng.uniform(5, 60, nd), rng.uniform(5, 60, nd)
        d['class_id'] = rng.randint(0, 2, nd)
        d['class_confidence'] = np.round(rng.uniform(0, 1, nd) * 16) / 16
        gts.append(g)
        dts.append(d)
    anns, res = oc.to_coco_records(gts, dts)
    p_ref, r_ref = oc.coco_tables(anns, res, len(gts), 3)
    p, r = _native_tables(gts, dts, 3)
    assert np.array_equal(p, p_ref) and np.array_equal(r, r_ref)
    assert (p[:, :, 2] == -1).all() and (p[:, :, :2, 0] > -1).all()


def _rec(rows):
    a = np.zeros((len(rows),), EVAL_BBOX_DTYPE)
    for i, r in enumerate(rows):
        a[i]['t'], a[i]['x'], a[i]['y'], a[i]['w'], a[i]['h'], a[i]['class_id'], a[i]['class_confidence'] = r
    return a


def test_known_answers():
    """Hand-checkable cases, for the oracle and the native evaluator alike."""
    from leod_amd.utils.evaluation.prophesee.metrics.coco_eval import evaluate_detection
    t = 1000000
    gt = _rec([(t, 10, 10, 50, 50, 0, 1.0), (t, 100, 100, 40, 40, 0, 1.0)])
    # 1. detections identical to the labels: every AP that is defined is 1; both boxes are 'medium' (32^2..96^2)
    dt = gt.copy()
    dt['class_confidence'] = [0.9, 0.8]
    for fn in (lambda g, d: oc.evaluate_detection([g], [d], 2), lambda g, d: evaluate_detection([g], [d])):
        m = fn(gt, dt)
        assert m['AP'] == 1.0 and m['AP_50'] == 1.0 and m['AP_75'] == 1.0 and m['AP_M'] == 1.0
        assert m['AP_S'] == -1.0 and m['AP_L'] == -1.0
    # 2. hit (0.9), miss (0.8), hit (0.7): tp = 1,1,2  fp = 0,1,1  -> precision envelope 1, 2/3, 2/3 at recall .5, .5, 1
    #    101-point AP = (51 * 1 + 50 * 2/3) / 101 at every IoU threshold (the hits have IoU 1)
    dt = _rec([(t, 10, 10, 50, 50, 0, 0.9), (t, 200, 10, 40, 40, 0, 0.8), (t, 100, 100, 40, 40, 0, 0.7)])
    want = (51 + 50 * (2.0 / (1 + 2 + np.spacing(1)))) / 101
    for fn in (lambda g, d: oc.evaluate_detection([g], [d], 2), lambda g, d: evaluate_detection([g], [d])):
        m = fn(gt, dt)
        assert abs(m['AP'] - want) < 1e-12 and abs(m['AP_50'] - want) < 1e-12 and abs(m['AP_75'] - want) < 1e-12
    # 3. a detection of IoU 0.62 counts at thresholds 0.5, 0.55, 0.6 only: AP = 3/10, AP_50 = 1 (= 1/(1+eps)), AP_75 = 0
    g1 = _rec([(t, 0, 0, 100, 100, 1, 1.0)])
    d1 = _rec([(t, 0, 0, 100, 62, 1, 0.5)])
    for fn in (lambda g, d: oc.evaluate_detection([g], [d], 2), lambda g, d: evaluate_detection([g], [d])):
        m = fn(g1, d1)
        assert abs(m['AP_50'] - 1.0) < 1e-12 and m['AP_75'] == 0.0 and abs(m['AP'] - 0.3) < 1e-12 and m['AP_L'] == m['AP']
    # 4. wrong class never matches; detections outside the +-50 ms window are not seen at all
    d2 = _rec([(t, 0, 0, 100, 100, 0, 0.9)])
    d3 = _rec([(t + 50001, 0, 0, 100, 100, 1, 0.9)])
    d4 = _rec([(t + 50000, 0, 0, 100, 100, 1, 0.9)])
    for fn in (lambda g, d: oc.evaluate_detection([g], [d], 2), lambda g, d: evaluate_detection([g], [d])):
        assert fn(g1, d2)['AP'] == 0.0
        assert fn(g1, d3) == {k: 0.0 for k in oc.OUT_KEYS}        # no detections in any window: the all-zero dictionary
        assert abs(fn(g1, d4)['AP'] - 1.0) < 1e-12


@pytest.mark.parametrize('ci', range(len(EVAL_CASES)))
def test_evaluator_buffer_equals_oracle(ci):
    """PropheseeEvaluator.evaluate_buffer (overall + per class, filters on) == the oracle, key for key, bit for bit."""
    from leod_amd.utils.evaluation.prophesee.evaluator import PropheseeEvaluator
    case = EVAL_CASES[ci]
    labels, dets = synth_eval_sequences(**case)
    gen4 = case.get('n_cls', 2) == 3
    ev = PropheseeEvaluator('gen4' if gen4 else 'gen1', downsample_by_2=gen4)
    assert not ev.has_data()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        assert ev.evaluate_buffer(240, 304) is None and len(w) == 1
    ev.add_labels(labels[:1])
    ev.add_predictions(dets[:1])
    ev.add_labels(labels[1:])
    ev.add_predictions(dets[1:])
    hw = case.get('hw', (240, 304))
    got = ev.evaluate_buffer(hw[0], hw[1])
    want = oc.evaluate_buffer(labels, dets, 'gen4' if gen4 else 'gen1', gen4)
    assert got == want
    names = ('ped', 'cyc', 'car') if gen4 else ('car', 'ped')
    assert set(got) == set(oc.OUT_KEYS) | {f'{k}_{n}' for k in oc.OUT_KEYS for n in names}
    ev.reset_buffer()
    assert not ev.has_data()


def test_evaluation_throughput():
    """A validation-set sized buffer (20k labelled frames as single-frame records, the way the module feeds it) evaluates in
    seconds."""
    import time
    from leod_amd.utils.evaluation.prophesee.evaluation import evaluate_list
    labels, dets = synth_eval_sequences(seed=9, n_seq=40, n_frames=500, dt_per_gt=(0, 3))
    t0 = time.time()
    m = evaluate_list(dets, labels, 240, 304, 'gen1')
    dt = time.time() - t0
    assert 0.0 < m['AP'] < 1.0 and dt < 20.0, (m, dt)
