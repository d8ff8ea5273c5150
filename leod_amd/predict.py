"""Pseudo-label generation driver (what predict.py:167-246 of the reference does with ``pl.Trainer.predict``), for 1..N ranks.

    module = fetch_model_module(config)            # PseudoLabeler
    summary = run_pseudo_labeling(config, module, data_module)

The reference asserts a single GPU (predict.py:167-169).  Recordings are independent, so N ranks simply shard them -- one
process per GPU, no collective on the data path: ``ShardedStreamingDataPipe`` keeps a recording and its time-reversed TTA copy on
one rank, every rank runs ``PseudoLabeler.predict_step`` over its own stream and writes its own recordings
(``EventSeqData.save``).  Only the end of the run exchanges anything: the per-rank counts and the detection records of the frames
whose GT was withheld (pseudo_labeler.py:756-763) are gathered with ``all_gather_object`` and rank 0 computes the quality KPIs."""
import os
from typing import Any, Dict, Optional

import torch
import torch.distributed as dist

from leod_amd.modules.utils.detection import Mode


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def run_pseudo_labeling(config, module, data_module, device: Optional[torch.device] = None, save: bool = True,
                        process_group=None, prefetch_to_device: bool = True) -> Dict[str, Any]:
    """Returns (on every rank) {'num_sequences': total, 'num_sequences_rank': [...], 'metrics': KPI dict | None,
    'saved': [paths written by THIS rank]}."""
    rank, world = _world(process_group)
    if device is None:
        device = next(module.parameters()).device
    module.eval()
    module.setup('predict')
    data_module.setup('predict')
    loader = data_module.predict_dataloader()
    if prefetch_to_device and torch.device(device).type == 'cuda':
        from leod_amd.modules.data.prefetch import DevicePrefetcher
        batches = DevicePrefetcher(loader, module, device)
    else:
        batches = (module.transfer_batch_to_device(b, device, 0) for b in loader)
    pipelined = hasattr(module, 'flush_predictions') and torch.device(device).type == 'cuda'
    if pipelined:
        module.pipelined = True                # host bookkeeping of chunk i - 1 under the device work of chunk i
    with torch.no_grad():
        for i, batch in enumerate(batches):
            module.predict_step(batch, i)
        if pipelined:
            module.flush_predictions()
            module.pipelined = False
    saved = []
    if save and module.save_dir:
        if rank == 0:
            os.makedirs(module.save_dir, exist_ok=True)
        if world > 1:
            dist.barrier(group=process_group)
        for ev_data in module.ev_path_2_ev_data.values():
            assert ev_data.eoe, 'some data are not evaluated in full sequence'
            saved.append(ev_data.save(save_dir=module.save_dir, dst_name=module.dst_name))
    # ---- end of run: one small gather ----------------------------------------------------------------------------------
    evaluator = module.mode_2_psee_evaluator.get(Mode.TEST)
    mine = dict(ev_cnt=module.ev_cnt, paths=sorted(module.ev_path_2_ev_data),
                quality={k: (m.sum, m.count) for k, m in getattr(module, 'metrics', {}).items()},
                results={k: list(v) for k, v in getattr(module, 'results', {}).items()},
                labels=evaluator._buffer[evaluator.LABELS] if evaluator is not None else [],
                predictions=evaluator._buffer[evaluator.PREDICTIONS] if evaluator is not None else [],
                hw=module.mode_2_hw[Mode.TEST], batch_size=module.mode_2_batch_size[Mode.TEST])
    if world > 1:
        everyone = [None] * world
        dist.all_gather_object(everyone, mine, group=process_group)
    else:
        everyone = [mine]
    all_paths = [p for r in everyone for p in r['paths']]
    assert len(all_paths) == len(set(all_paths)), 'a recording was processed by more than one rank'
    metrics = None
    if rank == 0 and evaluator is not None and any(r['labels'] for r in everyone):
        evaluator.reset_buffer()
        for r in everyone:
            if r['labels']:
                evaluator.add_labels(r['labels'])
                evaluator.add_predictions(r['predictions'])
        hw = next(r['hw'] for r in everyone if r['hw'] is not None)
        metrics = evaluator.evaluate_buffer(img_height=hw[0], img_width=hw[1])
    if world > 1:
        box = [metrics]
        dist.broadcast_object_list(box, src=0, group=process_group)
        metrics = box[0]
    # precision / recall of the pseudo labels on the frames whose GT was withheld (PseudoLabeler._evaluate_pseudo_label): the ranks' running
    # means merge by their weights; the raw (best IoU, confidence) lists are written next to the generated dataset as the reference does
    # (predict.py:226-230: <parent of save_dir>/model_results.pkl)
    sums: Dict[str, Any] = {}
    for r in everyone:
        for k, (sm, n) in r['quality'].items():
            a = sums.setdefault(k, [0., 0])
            a[0] += sm
            a[1] += n
    label_quality = {k: v[0] / max(v[1], 1) for k, v in sums.items()} or None
    results_fn = None
    if save and module.save_dir and rank == 0:
        merged: Dict[str, list] = {}
        for r in everyone:
            for k, v in r['results'].items():
                merged.setdefault(k, []).extend(v)
        if merged:
            import pickle
            import numpy as np
            results_fn = os.path.join(os.path.dirname(module.save_dir.rstrip('/')) or '.', 'model_results.pkl')
            with open(results_fn, 'wb') as f:
                pickle.dump({k: np.array(v) for k, v in merged.items()}, f)
    return {'num_sequences': sum(r['ev_cnt'] for r in everyone), 'num_sequences_rank': [r['ev_cnt'] for r in everyone],
            'metrics': metrics, 'label_quality': label_quality, 'model_results': results_fn, 'saved': saved}


# ---- integrity check of a generated recording (the reference's own end-to-end verifier, predict.py:35-115) ---------------------------
def read_old_and_new_data(new_dir: str, old_dir: Optional[str] = None):
    """The frame count / size of the source recording and the (frame index, label) tables of the source and the generated recording
    (predict.py:35-55; the source defaults to ``datasets/<gen1|gen4>/train/<recording>`` as there)."""
    import numpy as np
    from leod_amd.data.utils import misc
    new_dir = new_dir[:-1] if new_dir[-1] == '/' else new_dir
    dst_name = 'gen1' if 'gen1' in new_dir else 'gen4'
    if old_dir is None:
        old_dir = os.path.join('datasets', dst_name, 'train', os.path.basename(new_dir))
    old_ev_dir = misc.get_ev_dir(old_dir)
    raw = misc.resolve_link(misc.get_ev_raw_fn(old_ev_dir, dst_name))
    frames = misc.RawFrames(raw) if os.path.exists(raw) else misc.H5Frames(misc.resolve_link(misc.get_ev_h5_fn(old_ev_dir, dst_name)))
    shape = tuple(frames.data.shape)
    frames.close()
    new_labels, new_idx = misc.read_npz_labels(new_dir)
    old_labels, old_idx = misc.read_npz_labels(old_dir)
    return (shape, np.asarray(misc.read_objframe_idx_2_repr_idx(new_dir)), new_labels, new_idx,
            np.asarray(misc.read_objframe_idx_2_repr_idx(old_dir)), old_labels, old_idx)


def get_label(labels, objframe_idx_2_label_idx, objframe_idx: int, hw, ds_by2: bool):
    """The boxes of one labelled frame, clamped to the frame, as numpy-backed ``ObjectLabels`` (predict.py:57-64)."""
    from leod_amd.data.genx_utils.labels import ObjectLabels
    start = objframe_idx_2_label_idx[objframe_idx]
    end = objframe_idx_2_label_idx[objframe_idx + 1] if objframe_idx < len(objframe_idx_2_label_idx) - 1 else labels.shape[0]
    lab = ObjectLabels.from_structured_array(labels[start:end], hw)
    if ds_by2:
        lab.scale_(scaling_multiplier=0.5)
    lab.clamp_to_frame_()
    lab.numpy_()
    return lab


def verify_data(new_dir: str, ratio: float = -1, ds_by2: bool = False, old_dir: Optional[str] = None, label_list=None) -> int:
    """Is a generated (pseudo-labelled) recording consistent with its source?  The reference runs this on 10 % of the recordings it writes
    (predict.py:67-115, called at :246-256): the labelled-frame table is sorted and inside the recording; every frame whose GT the sparse
    label list (``ssod_<ratio>-off0.pkl``, or ``label_list``) keeps is still there with all eight fields unchanged to 1e-6; every other frame
    of the source that survived holds pseudo labels only (t == 0); confidences lie in [0, 1].  -> number of retained GT frames checked."""
    import numpy as np
    from leod_amd.data.genx_utils.labels import ObjectLabels
    shape, new_o2r, new_labels, new_o2l, old_o2r, old_labels, old_o2l = read_old_and_new_data(new_dir, old_dir)
    hw = tuple(shape[-2:])
    if ds_by2:
        hw = tuple(s * 2 for s in hw)
    if label_list is None:
        if 0. < ratio < 1.:
            import pickle
            from leod_amd.data.genx_utils.dataset_streaming import SPLITS_DIR
            dst_name = 'gen1' if 'gen1' in new_dir else 'gen4'
            with open(os.path.join(SPLITS_DIR, dst_name, f'ssod_{ratio:.3f}-off0.pkl'), 'rb') as f:
                label_list = pickle.load(f)[os.path.basename(new_dir.rstrip('/'))]
        else:
            label_list = list(range(len(old_o2r)))
    label_set = set(int(i) for i in label_list)
    assert new_o2r[-1] <= shape[0], (new_o2r[-1], shape[0])
    assert (new_o2r >= 0).all()
    assert all(idx == new_o2r[:i + 1].max() for i, idx in enumerate(new_o2r)), 'labelled frames not sorted'
    assert all(old_o2r[i] in new_o2r for i in label_set), 'a GT frame of the sparse label list is missing'
    checked = 0
    for old_frame_idx, repr_idx in enumerate(old_o2r):
        hit = np.where(new_o2r == repr_idx)[0]
        if len(hit) == 0:
            assert old_frame_idx not in label_set, 'GT not retained'
            continue
        new_label = get_label(new_labels, new_o2l, int(hit[0]), hw=hw, ds_by2=ds_by2)
        assert (new_label.objectness >= 0).all() and (new_label.objectness <= 1).all()
        assert (new_label.class_confidence >= 0).all() and (new_label.class_confidence <= 1).all()
        if old_frame_idx not in label_set:
            assert new_label.is_pseudo_label().all(), 'should not contain GT'
            continue
        old_label = get_label(old_labels, old_o2l, old_frame_idx, hw=hw, ds_by2=ds_by2)
        for k in ObjectLabels.keys():
            assert np.abs(old_label.get(k) - new_label.get(k)).max() < 1e-6, (k, old_frame_idx)
        checked += 1
    return checked


# ---- quality of a generated dataset against the labels that were withheld (val_dst.py:36-160 of the reference) ----------------------------
def filter_bbox(pred, obj_thresh=0.9, cls_thresh=0.9, ignore_label=1024):
    """Keep the boxes above the (per-class) objectness AND class-confidence thresholds that are not ignore boxes (val_dst.py:36-46; in place)."""
    from leod_amd.modules.utils.ssod import filter_w_thresh
    o = pred.object_labels
    cls_idx = o[:, 5]
    obj_t = obj_thresh if isinstance(obj_thresh, float) else list(obj_thresh)
    cls_t = cls_thresh if isinstance(cls_thresh, float) else list(cls_thresh)
    keep = filter_w_thresh(o[:, 7], cls_idx, obj_t) & filter_w_thresh(o[:, 6], cls_idx, cls_t) & (cls_idx != ignore_label)
    pred.object_labels = o[keep]
    return pred


def evaluate_pseudo_dataset(config, pseudo_path: str, original_path: str, skip_gt: Optional[bool] = None, data_module_factory=None) -> Dict[str, float]:
    """``val_dst.py`` as a function: both datasets are streamed label-only (``dataset.only_load_labels``, one recording per batch, whole
    recordings) -- the generated one with every label, the original one with the sparse-label regime of ``config`` (``dataset.ratio`` or
    ``dataset.train_ratio``) so that the withheld GT arrives as ``SKIPPED_OBJLABELS_SEQ``.  Checked per frame as there (:66-71): a frame
    whose GT was kept holds exactly that GT in the generated dataset, any other labelled frame holds pseudo labels only.  Scored on the
    frames whose GT was withheld: precision / recall of the thresholded pseudo labels (``evaluate_label``), class-frame-weighted over the
    recordings.  -> {'ssod/teacher_AR@50_car': ..}"""
    import copy
    import numpy as np
    from leod_amd.data.genx_utils.labels import ObjectLabels
    from leod_amd.data.utils.types import DataType
    from leod_amd.modules.utils.detection import DATA_KEY
    from leod_amd.modules.utils.fetch import fetch_data_module
    from leod_amd.modules.utils.ssod import evaluate_label, AverageMeter
    make = data_module_factory or fetch_data_module
    cfg = copy.deepcopy(config)
    dst_name = cfg.dataset.name
    cfg.batch_size.eval = 1
    cfg.dataset.sequence_length = 320 if dst_name == 'gen1' else 128
    cfg.dataset.only_load_labels = True
    cfg.dataset.data_augmentation.stream.start_from_zero = True
    sparse_ratio, subseq_ratio = cfg.dataset.ratio, cfg.dataset.train_ratio
    if sparse_ratio == -1:
        assert 0. < subseq_ratio < 1., 'neither dataset.ratio nor dataset.train_ratio describes a sparse-label regime'
        cfg.dataset.train_ratio = -1
    else:
        assert 0. < sparse_ratio < 1.
        cfg.dataset.ratio = -1
    if skip_gt is None:
        skip_gt = 'all_pse' in pseudo_path
    cfg.dataset.path = pseudo_path
    pse_dm = make(cfg)
    pse_dm.setup('predict')
    pse_loader = pse_dm.predict_dataloader()
    cfg2 = copy.deepcopy(cfg)
    cfg2.dataset.ratio, cfg2.dataset.train_ratio = sparse_ratio, subseq_ratio
    cfg2.dataset.path = original_path
    dm = make(cfg2)
    dm.setup('predict')
    loader = dm.predict_dataloader()
    pl_cfg = cfg.model.pseudo_label
    meters: Dict[str, Any] = {}
    for pse_batch, batch in zip(pse_loader, loader):
        pse_data, data = pse_batch[DATA_KEY], batch[DATA_KEY]
        assert os.path.basename(pse_data[DataType.PATH][0]) == os.path.basename(data[DataType.PATH][0]), 'the two datasets stream different recordings'
        pse_l = [lbl[0] for lbl in pse_data[DataType.OBJLABELS_SEQ]]
        gt_l = [lbl[0] for lbl in data[DataType.OBJLABELS_SEQ]]
        sk_l = [lbl[0] for lbl in data[DataType.SKIPPED_OBJLABELS_SEQ]]
        kept_pse, kept_gt = [], []
        for pse, gt, sk in zip(pse_l, gt_l, sk_l):
            if gt is not None and not skip_gt:
                assert gt == pse, 'GT labels mismatch'
            elif pse is not None:
                assert bool(pse.is_pseudo_label().all()), 'Contain GT labels'
            if sk is not None:
                kept_gt.append(sk)
                if pse is None:
                    pse = ObjectLabels(sk.object_labels.new_zeros((0, 8)), sk.input_size_hw)
                else:
                    pse = filter_bbox(copy.deepcopy(pse), obj_thresh=pl_cfg.obj_thresh, cls_thresh=pl_cfg.cls_thresh, ignore_label=pl_cfg.ignore_label)
                kept_pse.append(pse)
        if not kept_gt:
            continue
        m = evaluate_label(kept_gt, kept_pse, np.ones(len(kept_gt), dtype=bool), num_cls=cfg.model.head.num_classes, prefix='ssod/')
        for k, v in m.items():
            if not k.startswith('num_'):
                meters.setdefault(k, AverageMeter()).update(v, n=m[f'num_{k.split("_")[-1]}'])
    return {k: v.avg for k, v in meters.items()}
