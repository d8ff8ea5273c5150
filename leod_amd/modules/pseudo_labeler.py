"""Pseudo-label generation loop with the reference's surface (modules/pseudo_labeler.py:27-796):
``BBOX_DTYPE``, ``tta_postprocess`` (ObjectLabels flavour), ``EventSeqData`` and ``PseudoLabeler``.

The per-batch hot loop (hflip copy, frame masks, backbone over L, head + postprocess + pred2label on the
selected frames) runs on the HIP kernels; the ragged per-sequence bookkeeping stays host Python as in the
reference.  Tracker-based filtering / in-painting (modules/tracking, pseudo_labeler.py:201-333) runs in the native
library (``leod_amd.modules.tracking`` -> ``leod_track_filter``): ``EventSeqData.save`` aggregates the TTA views,
applies the tracking post-filter configured in ``filter_config`` and writes the labels."""
import copy
import os
import os.path as osp
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch as th

from leod_amd.data.genx_utils.labels import ObjectLabels, SparselyBatchedObjectLabels, BBOX_DTYPE  # noqa: F401
from leod_amd.data.utils import misc
from leod_amd.data.utils.types import DataType
from leod_amd import ops
from leod_amd.models.detection.yolox.utils.boxes import postprocess, postprocess_padded  # noqa: F401
from .detection import Module
from .utils.detection import BackboneFeatureSelector, SeqLens, Mode, DATA_KEY
from .utils.ssod import pred2label, pred2label_padded, filter_pred_boxes  # noqa: F401
from .utils.tta import tta_postprocess as _tta_rows
from .tracking import track as _native_track


def tta_postprocess(preds: List[ObjectLabels], conf_thre: float = 0.7, nms_thre: float = 0.45,
                    class_agnostic: bool = False) -> List[ObjectLabels]:
    """Merge the (already concatenated) TTA predictions of each frame; frames that hold GT are passed through."""
    if len(preds) == 0:
        return preds
    out: List[ObjectLabels] = [preds[0].new_zeros()] * len(preds)
    rows, keep_idx = [], []
    for i, p in enumerate(preds):
        if len(p) and bool(p.is_gt_label().any()):
            out[i] = p
            continue
        if len(p) == 0:
            continue
        rows.append(p.get_labels_as_tensors(format_='prophesee'))
        keep_idx.append(i)
    merged = _tta_rows(rows, conf_thre, nms_thre, class_agnostic, pad=None)
    for i, det in zip(keep_idx, merged):
        if det is None:
            continue
        # pseudo labels carry t == 0, so the time column of the merged boxes is 0 as well
        lab = th.cat([th.zeros_like(det[:, :1]), det[:, 0:2], det[:, 2:4] - det[:, 0:2], det[:, 6:7], det[:, 5:6], det[:, 4:5]], 1)
        out[i] = ObjectLabels(lab, preds[i].input_size_hw)
    return out


class EventSeqData:
    """Accumulates the labels of one recording across batches and TTA views."""

    def __init__(self, path: str, scale_ratio: float, filter_config, postproc_cfg):
        self.path, self.scale_ratio = path, scale_ratio
        self.filter_config, self.postproc_cfg = filter_config, postproc_cfg
        self._eoe, self._aug = False, False
        self.frame_idx_2_labels: Dict[int, ObjectLabels] = {}

    def update(self, labels: List[Optional[ObjectLabels]], ev_idx: List[int], is_last_sample: bool,
               is_padded_mask: List[bool], is_hflip: bool, is_tflip: bool, tflip_offset: int) -> None:
        self._eoe = is_last_sample
        if is_hflip:                       # un-flip: x <- W - 1 - x - w
            for l in labels:
                if l is not None:
                    l.flip_lr_()
            self._aug = True
        if is_tflip:
            ev_idx = [i + tflip_offset for i in ev_idx]
            self._aug = True
        for tidx, (label, frame_idx) in enumerate(zip(labels, ev_idx)):
            if frame_idx < 0 or label is None or len(label) == 0:
                continue
            assert not is_padded_mask[tidx]
            label.scale_(self.scale_ratio)  # labels are stored at the recording's native resolution (reference :139)
            if frame_idx in self.frame_idx_2_labels:
                if bool(label.is_gt_label().any()):
                    continue               # GT is stored once (a differing copy only draws a warning in the reference, :144-152)
                self.frame_idx_2_labels[frame_idx] = self.frame_idx_2_labels[frame_idx] + label
            else:
                self.frame_idx_2_labels[frame_idx] = label

    @property
    def eoe(self) -> bool:
        return self._eoe

    @property
    def aug(self) -> bool:
        return self._aug

    def _aggregate_results(self, num_frames: int) -> None:
        assert self._eoe, 'Cannot aggregate results before the sequence ends.'
        self.frame_idx = sorted(i for i in self.frame_idx_2_labels if 0 <= i < num_frames)
        self.labels = [self.frame_idx_2_labels[i] for i in self.frame_idx]
        if self._aug and self.labels:
            self.labels = tta_postprocess(self.labels, conf_thre=self.postproc_cfg.confidence_threshold,
                                          nms_thre=self.postproc_cfg.nms_threshold)

    @staticmethod
    def _track(labels: List[ObjectLabels], frame_idx: List[int], min_track_len: int = 6, inpaint: bool = False):
        """Track the boxes of one recording and report those on short tracklets (reference :201-258; same signature and
        return values: ``remove_idx`` and, with ``inpaint``, ``{frame: [n,8] label rows}``).  One native call."""
        assert len(labels) == len(frame_idx)
        if len(labels) == 0:
            return []
        boxes = [l.get_xywh(format_='center', add_class_id=True).detach().cpu().numpy() for l in labels]
        is_gt = [l.is_gt_label().detach().cpu().numpy() for l in labels]
        return _native_track(boxes, is_gt, frame_idx, labels[0].input_size_hw, min_track_len, inpaint)

    def _track_filter(self) -> None:
        """Forward / forward-and-backward tracking; boxes on short tracklets get ``ignore_label`` as class id,
        missed detections of surviving tracklets are in-painted with ``ignore_label`` (reference :260-333)."""
        if len(self.labels) == 0:
            return
        fc = self.filter_config
        min_track_len = fc.min_track_len
        if min_track_len <= 0:
            return
        track_method = fc.track_method
        assert track_method in ['forward', 'forward or backward'], f'Unknown tracking post-processing {track_method}'
        remove_idx, inpainted = self._track(self.labels, self.frame_idx, min_track_len=min_track_len, inpaint=fc.inpaint)
        if 'backward' in track_method:
            rev_labels = [l.get_reverse() for l in self.labels[::-1]]
            rev_frame_idx = [max(self.frame_idx) - i for i in self.frame_idx[::-1]]
            bg_remove_idx, _ = self._track(rev_labels, rev_frame_idx, min_track_len=min_track_len, inpaint=False)
            nlabels = sum(len(l) for l in self.labels)
            remove_idx = list(set(remove_idx) & {nlabels - i - 1 for i in bg_remove_idx})
        remove = set(remove_idx)
        bbox_idx = 0
        for label in self.labels:
            n = len(label)
            hit = [i for i in range(n) if bbox_idx + i in remove]
            if hit:
                assert bool(label.is_pseudo_label().all()), 'Ignoring GT!'
                label.object_labels[hit, 5] = float(fc.ignore_label)
            bbox_idx += n
        if not inpainted:
            return
        hw = self.labels[0].input_size_hw
        ref = self.labels[0].object_labels
        for f_idx in range(max(self.frame_idx) + 1):
            if f_idx not in inpainted:
                continue
            rows = torch.from_numpy(inpainted[f_idx]).to(device=ref.device, dtype=ref.dtype)
            rows[:, 5] = float(fc.ignore_label)
            extra = ObjectLabels(rows, hw)
            if f_idx in self.frame_idx:
                k = self.frame_idx.index(f_idx)
                assert bool(self.labels[k].is_pseudo_label().all()), 'Inpaint ignored bbox at labeled frames!'
                self.labels[k] = self.labels[k] + extra
            else:
                self.frame_idx.append(f_idx)
                self.labels.append(extra)
        order = sorted(range(len(self.frame_idx)), key=lambda k: self.frame_idx[k])
        self.labels = [self.labels[k] for k in order]
        self.frame_idx = [self.frame_idx[k] for k in order]

    def _summarize(self):
        labels, cnt, f2l, f2r = [], 0, [], []
        for label, fidx in zip(self.labels, self.frame_idx):
            f2l.append(cnt)
            cnt += len(label)
            labels.append(label.to_structured_array())
            f2r.append(fidx)
        labels = np.concatenate(labels) if labels else np.zeros((0,), dtype=BBOX_DTYPE)
        return labels, np.array(f2l, dtype=np.int64), np.array(f2r, dtype=np.int64)

    def save(self, save_dir: str, dst_name: str, num_frames: Optional[int] = None) -> str:
        """Write the recording into the new dataset tree exactly where the reference's loaders look for it (:335-397):

            save_dir/<recording>/event_representations_v2/<ev_repr_name>/<frame file>   soft link to the source recording's
            save_dir/<recording>/event_representations_v2/<ev_repr_name>/objframe_idx_2_repr_idx.npy
            save_dir/<recording>/labels_v2/labels.npz                                   labels, objframe_idx_2_label_idx
            dirname(save_dir)/{val,test}                                                soft links to the source splits

        Refuses to overwrite (``os.makedirs(exist_ok=False)``).  ``num_frames`` defaults to the frame count of the source
        recording's frame file (HDF5 or its raw .npy twin)."""
        assert dst_name in ('gen1', 'gen4')
        assert 'train' in save_dir and dst_name in save_dir
        assert 'train' in self.path and dst_name in self.path
        path = osp.normpath(self.path)
        base_dir = osp.dirname(osp.dirname(path))
        sources = [(fn, misc.resolve_link(fn)) for fn in misc.ev_repr_files(path, dst_name)]
        if num_frames is None:
            frames = misc.open_ev_repr(path, dst_name)
            num_frames = len(frames)
            frames.close()
        new_base_dir = osp.dirname(osp.normpath(save_dir))
        new_seq_dir = osp.join(save_dir, osp.basename(path))
        new_ev_dir = misc.get_ev_dir(new_seq_dir)
        new_labels_npz_fn = misc.get_labels_npz_fn(new_seq_dir)
        os.makedirs(new_ev_dir, exist_ok=False)
        os.makedirs(osp.dirname(new_labels_npz_fn), exist_ok=False)
        for fn, target in sources:                      # the frames are linked, never copied
            os.symlink(osp.abspath(target), osp.join(new_ev_dir, osp.basename(fn)))
        self._aggregate_results(num_frames=num_frames)
        self._track_filter()
        labels, f2l, f2r = self._summarize()
        np.save(misc.get_objframe_idx_2_repr_idx_fn(new_ev_dir), f2r)
        np.savez(new_labels_npz_fn, labels=labels, objframe_idx_2_label_idx=f2l)
        # link the evaluation splits once, for completeness (:386-397).  Idempotent per split: with N ranks saving after the same
        # barrier (leod_amd/predict.py) a check-then-create races (two ranks both see 'val' missing, or one sees 'val' before 'test').
        for split in ('val', 'test'):
            try:
                os.symlink(osp.abspath(misc.resolve_link(osp.join(base_dir, split))), osp.join(new_base_dir, split))
            except FileExistsError:
                pass
        return new_seq_dir


class PseudoLabeler(Module):
    control_tensors_on_host = True          # flags / indices / masks of a chunk stay host tensors (Module.transfer_batch_to_device)

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.mode_2_seq_lens = SeqLens()
        self.ev_path_2_ev_data: Dict[str, EventSeqData] = {}
        self.ev_cnt = 0
        self.metrics: Dict[str, Any] = {}                  # per-class precision / recall of the pseudo labels on withheld GT (reference :449)
        self.results: Dict[str, List[float]] = {}          # best IoU / confidences of the pseudo boxes (reference :450)
        self.dst_name = self.dst_config.name
        self.ds_by2 = self.dst_config.downsample_by_factor_2
        assert self.dst_name in ('gen1', 'gen4')
        self.save_dir = self.full_config.get('save_dir', '')
        if self.save_dir:
            assert self.dst_name in self.save_dir and 'train' in self.save_dir
            assert not osp.exists(self.save_dir), f'{self.save_dir} already exists'
        self.use_gt = self.full_config.get('use_gt', True)
        self.tta_cfg = self.full_config.tta
        self.filter_bbox_fn = lambda b: filter_pred_boxes(b, dataset_name=self.dst_name, downsampled_by_2=self.ds_by2)
        self.pipelined = False                 # host half of chunk i - 1 under the device half of chunk i (see predict_step)
        self._pending = None
        self._host_slots: List[Any] = []
        self._host_next = 0

    def get_data_from_batch(self, batch: Any):
        """hflip TTA: the flipped copy is concatenated on the batch dim (B -> 2B) and labels are flipped; padding is
        applied by the stem kernel afterwards, i.e. to the right/bottom of the FLIPPED tensor as in the reference
        (pseudo_labeler.py:469-470,493)."""
        data = batch[DATA_KEY]
        assert DataType.AUGM_STATE not in data
        frames = data[DataType.EV_REPR]
        hflip = self.tta_cfg.enable and self.tta_cfg.hflip
        B = frames[0].shape[0]
        if hflip and frames[0].is_cuda and frames[0].element_size() == 1:
            ev = ops.stack_hflip_u8([f.contiguous() for f in frames])        # raw uint8 voxels: frames + flipped copies in one pass
        else:
            ev = th.stack(frames)
            if hflip:
                ev = th.cat([ev, th.flip(ev, dims=[-1])], dim=1)
        data['is_hflip'] = np.array([False] * B, dtype=bool)
        if hflip:
            new = {}
            for k in (DataType.IS_FIRST_SAMPLE, DataType.IS_LAST_SAMPLE, DataType.IS_REVERSED):
                new[k] = th.cat([data[k]] * 2, dim=-1)
            for k in (DataType.EV_IDX, DataType.IS_PADDED_MASK):
                new[k] = [th.cat([d] * 2, dim=-1) for d in data[k]]
            new[DataType.PATH] = data[DataType.PATH] * 2
            for k in (DataType.OBJLABELS_SEQ, DataType.SKIPPED_OBJLABELS_SEQ):
                labels, flipped = data[k], copy.deepcopy(data[k])
                for i, (l, lf) in enumerate(zip(labels, flipped)):
                    lf.flip_lr_()
                    labels[i] = l + lf
                new[k] = labels
            new['is_hflip'] = np.array([False] * B + [True] * B, dtype=bool)
            data = new
        data[DataType.EV_REPR] = list(ev.unbind(0))
        return data

    def _get_pred_mask(self, worker_id: int, data: Dict):
        obj_labels, skipped = data[DataType.OBJLABELS_SEQ], data[DataType.SKIPPED_OBJLABELS_SEQ]
        L, B = len(obj_labels), len(obj_labels[0])
        skip = np.zeros((L, B), dtype=bool)
        gt_mask = np.zeros((L, B), dtype=bool)
        skipped_gt = np.zeros((L, B), dtype=bool)
        skip_len = max(self.mdl_config.pseudo_label.skip_first_t, 1)
        prev_lens = self.mode_2_seq_lens.get_lens(worker_id=worker_id)
        for b in range(B):
            if prev_lens[b] < skip_len:
                skip[:skip_len - int(prev_lens[b]), b] = True
        for t in range(L):
            for b in range(B):
                has_gt = (obj_labels[t][b] is not None) and self.use_gt
                has_sk = skipped[t][b] is not None
                assert not (has_gt and has_sk)
                gt_mask[t, b] = has_gt
                skip[t, b] = has_gt
                skipped_gt[t, b] = has_sk
        skip[th.stack(data[DataType.IS_PADDED_MASK]).cpu().numpy()] = True
        return ~skip, gt_mask, skipped_gt

    # no_grad rather than the reference's inference_mode (:622): the labels are edited in place afterwards (un-flip,
    # tracking filter), which inference tensors only allow inside Lightning's own inference-mode predict loop
    @torch.no_grad()
    def _predict_step_impl(self, batch: Any, mode: Mode = Mode.TEST):
        if self.dst_config.get('only_load_labels', False):
            # tracking-only post-processing of an existing pseudo dataset (reference :625-637, predict.py:137-155): the loader delivers the labels
            # without event frames, nothing is predicted, EventSeqData applies the tracker filter when the recordings are saved
            data = batch[DATA_KEY]
            assert all(lbl.is_empty() for lbl in data[DataType.SKIPPED_OBJLABELS_SEQ])
            labels_bl = [list(lbl) for lbl in zip(*data[DataType.OBJLABELS_SEQ])]                  # L x B -> B x L
            B = len(labels_bl)
            out = (labels_bl, data[DataType.PATH], th.stack(data[DataType.EV_IDX]).transpose(1, 0).cpu().numpy().tolist(),
                   data[DataType.IS_FIRST_SAMPLE].cpu().numpy().tolist(), data[DataType.IS_LAST_SAMPLE].cpu().numpy().tolist(),
                   th.stack(data[DataType.IS_PADDED_MASK]).transpose(1, 0).cpu().numpy().tolist(),
                   data['is_hflip'] if 'is_hflip' in data else [False] * B, data[DataType.IS_REVERSED].cpu().numpy().tolist())
            return lambda: out
        data = self.get_data_from_batch(batch)
        worker_id = self.get_worker_id_from_batch(batch)
        ev_seq = data[DataType.EV_REPR]
        obj_labels, skipped = data[DataType.OBJLABELS_SEQ], data[DataType.SKIPPED_OBJLABELS_SEQ]
        is_first = data[DataType.IS_FIRST_SAMPLE]
        if is_first.device.type == 'cpu':                     # the loader's tensor (``control_tensors_on_host``): no device round trip at all
            first_host = is_first.numpy().tolist()
            dev0 = ev_seq[0].device
            is_first_dev = is_first.pin_memory().to(dev0, non_blocking=True) if dev0.type == 'cuda' else is_first
        else:
            # a device tensor: this copy is enqueued behind whatever the caller has in flight -- in pipelined mode the PREVIOUS chunk's
            # kernels, which the host then waits for (28.0 instead of 25.x ms per chunk in bench.py --pseudo)
            first_host = is_first.cpu().numpy().tolist()
            is_first_dev = is_first
        L, B = len(obj_labels), len(obj_labels[0])
        assert L > 0 and B > 0
        if self.mode_2_batch_size[mode] is None:
            self.mode_2_batch_size[mode] = B
        else:
            assert self.mode_2_batch_size[mode] == B
        hw = tuple(ev_seq[0].shape[-2:])
        if self.mode_2_hw[mode] is None:
            self.mode_2_hw[mode] = hw
        else:
            assert self.mode_2_hw[mode] == hw
        rnn = self.mode_2_rnn_states[mode]
        rnn.reset(worker_id=worker_id, indices_or_bool_tensor=is_first_dev)
        prev = rnn.get_states(worker_id=worker_id)
        self.mode_2_seq_lens.reset(worker_id=worker_id, indices_or_bool_tensor=th.tensor(first_host, dtype=th.bool))     # host lengths, host mask
        pse_mask, gt_mask, skipped_gt_mask = self._get_pred_mask(worker_id, data)
        skipped_gt_labels: List[ObjectLabels] = [skipped[t][b] for t in range(L) for b in range(B) if skipped_gt_mask[t, b]]
        gt_labels: List[ObjectLabels] = []
        if self.use_gt:
            for t in range(L):
                cur, _ = obj_labels[t].get_valid_labels_and_batch_indices()
                gt_labels.extend(cur)
        in_features = self.mdl.fpn.in_features
        rows = tuple(t * B + int(b) for t in range(L) for b in np.where(pse_mask[t])[0])
        feats = None
        if self.time_batched:
            # stage-major: every stage sees all L*B frames per launch, only the ConvLSTM walks over t (same values as the
            # per-timestep loop of pseudo_labeler.py:687-722, see RNNDetector.forward_sequence)
            ev = self._stack_frames(ev_seq)
            feats_all, prev = self.mdl.backbone.forward_sequence(ev, prev)
            if len(rows) == L * B:                            # every frame is predicted on (no GT frames in the chunk): the maps as they are
                feats = {k: feats_all[k] for k in in_features}
            elif rows:
                ridx = self._row_index(rows, ev.device)
                feats = {k: feats_all[k].permute(0, 2, 3, 1).index_select(0, ridx).permute(0, 3, 1, 2) for k in in_features}
        else:
            selector = BackboneFeatureSelector()
            for t in range(L):
                f_t, prev = self.mdl.forward_backbone(x=ev_seq[t], previous_states=prev)
                idx = np.where(pse_mask[t])[0].tolist()
                if idx:
                    selector.add_backbone_features({k: f_t[k] for k in in_features}, idx)
            feats = selector.get_batched_backbone_features()
        rnn.save_states_and_detach(worker_id=worker_id, states=prev)
        self.mode_2_seq_lens.update_lens(worker_id=worker_id, lens=torch.ones(B).long() * L)
        lab = lcnt = ready = None
        if feats is not None:
            preds, _ = self.mdl.forward_detect(backbone_features=feats)
            # postprocess + pred2label (reference :724-741) in their padded, sync-free forms: NMS and the pseudo-label filters of all
            # L * B frames are two launches -- the list forms cost one slice-copy launch per frame (672 per chunk of 32 streams x 21
            # frames) on top of their own synchronisation
            det, cnt = postprocess_padded(preds, self.num_classes, self.mdl_config.postprocess.confidence_threshold,
                                          self.mdl_config.postprocess.nms_threshold)
            lab, lcnt = pred2label_padded(det, cnt, self.mdl_config.pseudo_label.obj_thresh, self.mdl_config.pseudo_label.cls_thresh,
                                          self.dst_name, self.ds_by2)
            # The label rows and their counts go to the HOST in one asynchronous copy each, enqueued HERE -- behind this chunk's kernels
            # and in front of whatever the caller enqueues next: everything that follows is per-frame bookkeeping (un-flip, merge of
            # the TTA views, GT checks, scaling -- EventSeqData.update), which as device ops was ~1000 tiny launches and
            # synchronisations per chunk (22 of 57 ms).  The final TTA merge moves its padded batch to the device again (utils/tta.py).
            slot = self._host_slots[self._host_next] if self._host_slots else None
            if slot is None or slot[0].shape != lab.shape:
                self._host_slots = [(th.empty(lab.shape, dtype=lab.dtype).pin_memory(), th.empty(lcnt.shape, dtype=lcnt.dtype).pin_memory())
                                    for _ in range(2)]
                self._host_next = 0
                slot = self._host_slots[0]
            self._host_next = (self._host_next + 1) % 2
            slot[0].copy_(lab, non_blocking=True)
            slot[1].copy_(lcnt, non_blocking=True)
            ready = th.cuda.Event()
            ready.record()
            lab, lcnt = slot

        def finish():
            """The host half of the step: waits for THIS chunk's copies only (a pipelined caller has the next chunk enqueued already)."""
            pse_labels: List[ObjectLabels] = []
            if lab is not None:
                ready.synchronize()
                counts = lcnt.tolist()
                if counts and min(counts) < 0:
                    raise ops.LeodHipError('pred2label: an image overflowed the LDS candidate arrays and no workspace was given')
                hw_lab = tuple(self.dst_config.ev_repr_hw)
                rows = lab[:, :max(1, max(counts))].clone()        # the pinned slot is reused two chunks later
                pse_labels = [ObjectLabels(rows[i, :n], hw_lab) for i, n in enumerate(counts)]
            all_labels = [[None] * L for _ in range(B)]
            skipped_gt_pse_labels: List[ObjectLabels] = []
            gi = pi = 0
            for t in range(L):
                for b in range(B):
                    if skipped_gt_mask[t, b]:
                        assert pse_mask[t, b], 'should predict on skipped GT frames'
                        skipped_gt_pse_labels.append(pse_labels[pi])
                    if pse_mask[t, b]:
                        all_labels[b][t] = pse_labels[pi]
                        pi += 1
                    elif gt_mask[t, b]:
                        all_labels[b][t] = gt_labels[gi]
                        gi += 1
            assert pi == int(pse_mask.sum()) and gi == int(gt_mask.sum()) and len(skipped_gt_pse_labels) == len(skipped_gt_labels)
            if skipped_gt_labels:
                self._evaluate_pseudo_label(skipped_gt_labels, skipped_gt_pse_labels)
            if skipped_gt_labels and mode in self.mode_2_psee_evaluator:
                # quality of the pseudo labels on frames whose GT was withheld (reference :753-763): detection KPIs at the end of the
                # run (``run_psee_evaluator``; gathered over ranks by ``leod_amd.predict.run_pseudo_labeling``)
                from leod_amd.utils.evaluation.prophesee.io.box_loading import to_prophesee
                labels_proph, preds_proph = to_prophesee(skipped_gt_labels, skipped_gt_pse_labels)
                self.mode_2_psee_evaluator[mode].add_labels(labels_proph)
                self.mode_2_psee_evaluator[mode].add_predictions(preds_proph)
            return (all_labels, data[DataType.PATH], ev_idx_l, first_l, last_l, padding_l, data['is_hflip'], reversed_l)

        # small host-side lists of the chunk, read now (host tensors from the loader; device ones cost their copy here, not later)
        ev_idx_l = th.stack(data[DataType.EV_IDX]).transpose(1, 0).cpu().numpy().tolist()
        padding_l = th.stack(data[DataType.IS_PADDED_MASK]).transpose(1, 0).cpu().numpy().tolist()
        first_l = first_host
        last_l = data[DataType.IS_LAST_SAMPLE].cpu().numpy().tolist()
        reversed_l = data[DataType.IS_REVERSED].cpu().numpy().tolist()
        return finish

    def _consume(self, out) -> None:
        for labels, path, ev_idx, is_first, is_last, padded, hflip, tflip in zip(*out):
            if not path:                                      # padding slot of the streaming loader
                assert not is_first and not is_last and all(padded) and all(i == -1 for i in ev_idx), 'invalid empty data'
                continue
            if path not in self.ev_path_2_ev_data:
                assert is_first, 'should load the first sample first'
                self.ev_path_2_ev_data[path] = EventSeqData(path=path, scale_ratio=2. if self.ds_by2 else 1,
                                                            filter_config=self.mdl_config.pseudo_label,
                                                            postproc_cfg=self.mdl_config.postprocess)
                self.ev_cnt += 1
            self.ev_path_2_ev_data[path].update(labels=labels, ev_idx=ev_idx, is_last_sample=is_last,
                                                is_padded_mask=padded, is_hflip=bool(hflip), is_tflip=bool(tflip),
                                                tflip_offset=self.dst_config.data_augmentation.tflip_offset)

    def _evaluate_pseudo_label(self, gt_obj_labels, pse_obj_labels) -> None:
        """Precision / recall of the pseudo labels against the GT that was withheld, and the (best IoU, confidence) pairs of the pseudo boxes
        (reference :591-620): running per-class means in ``self.metrics``, raw lists in ``self.results`` (capped at 1e5 boxes)."""
        from leod_amd.modules.utils.ssod import evaluate_label, get_scores_ious, AverageMeter
        mask = np.ones(len(gt_obj_labels), dtype=bool)
        m = evaluate_label(gt_obj_labels, pse_obj_labels, pred_mask=mask, num_cls=self.num_classes, prefix='ssod/')
        for k, v in m.items():
            if k.startswith('num_'):
                continue
            self.metrics.setdefault(k, AverageMeter()).update(v, n=m[f'num_{k.split("_")[-1]}'])
        if self.results and len(self.results['ssod/true_ious_all']) > 1e5:
            return
        for k, v in get_scores_ious(gt_obj_labels, pse_obj_labels, pred_mask=mask, num_cls=self.num_classes, prefix='ssod/').items():
            self.results.setdefault(k, []).extend(v)

    def flush_predictions(self) -> None:
        """Pipelined mode: finish the chunk whose host half is still outstanding (call after the last ``predict_step``)."""
        pend, self._pending = self._pending, None
        if pend is not None:
            self._consume(pend())

    def predict_step(self, batch: Any, batch_idx: int = 0) -> None:
        """One chunk (reference :622-796).  ``self.pipelined`` (set by ``leod_amd.predict.run_pseudo_labeling``): the device work of chunk
        i is enqueued, THEN the host bookkeeping of chunk i - 1 runs under it; ``flush_predictions()`` finishes the last chunk.  Default:
        both halves here, the results are in ``ev_path_2_ev_data`` when the call returns."""
        finish = self._predict_step_impl(batch=batch, mode=Mode.TEST)
        if not self.pipelined:
            self._consume(finish())
            return
        pend, self._pending = self._pending, finish
        if pend is not None:
            self._consume(pend())
