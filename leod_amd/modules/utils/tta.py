"""TTA merge helpers with the reference's interface (modules/utils/tta.py:18-61): the boxes that several views of one
frame produced are merged by confidence filtering + batched NMS -- here ONE kernel launch for all frames."""
from typing import List

import torch as th

from leod_amd import ops
from leod_amd.models.detection.yolox.utils.boxes import GPU_VANILLA_LIMIT


def tta_postprocess_padded(rows: th.Tensor, conf_thre=0.7, nms_thre=0.45, class_agnostic=False):
    """rows [F, N, 7] = (xyxy, obj, cls_conf, cls_id), padding rows must carry obj < 0.  -> (det [F,N,7], cnt [F])."""
    return ops.postprocess_nms(rows, 0, conf_thre, nms_thre, class_agnostic, None, GPU_VANILLA_LIMIT)


def tta_postprocess(preds: List[th.Tensor], conf_thre: float = 0.7, nms_thre: float = 0.45, class_agnostic: bool = False,
                    pad=None) -> List[th.Tensor]:
    """preds: list of [n_i,7] (xyxy, obj_conf, cls_conf, cls_idx) -> same with NMS applied (score order)."""
    out = [pad] * len(preds)
    live = [i for i, p in enumerate(preds) if p is not None and p.shape[0] > 0]
    if not live:
        return out
    nmax = max(preds[i].shape[0] for i in live)
    dev = preds[live[0]].device
    on_host = dev.type == 'cpu'            # host-resident label rows (the pseudo-label loop keeps them there): assemble on the host,
    rows = th.zeros((len(live), nmax, 7), dtype=th.float32, device=dev)       # one copy to the device, one copy of the result back
    rows[:, :, 4] = -1.0
    for j, i in enumerate(live):
        rows[j, :preds[i].shape[0]] = preds[i]
    if on_host:
        rows = rows.to(th.device('cuda', th.cuda.current_device()))
    det, cnt = tta_postprocess_padded(rows, conf_thre, nms_thre, class_agnostic)
    counts = ops.host_counts(cnt, 'tta_postprocess')
    if on_host:
        det = det[:, :max(1, max(counts))].cpu()
    for j, (i, n) in enumerate(zip(live, counts)):
        if n > 0:
            out[i] = det[j, :n]
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Evaluation with test-time augmentation (reference: modules/utils/tta.py:64-387): every recording is seen in up to four
# views (plain, horizontally flipped, time-reversed, both); the detections of all views are mapped back to the plain
# frame, concatenated per labelled frame and merged by NMS before they go to the Prophesee evaluator.
# ---------------------------------------------------------------------------------------------------------------------
import os  # noqa: E402
from typing import Any, Dict, Tuple  # noqa: E402

import numpy as np  # noqa: E402

from leod_amd.data.genx_utils.labels import ObjectLabels  # noqa: E402
from leod_amd.data.utils.types import DataType, DatasetSamplingMode  # noqa: E402
from leod_amd.models.detection.yolox.utils.boxes import postprocess  # noqa: E402
from leod_amd.modules.detection import Module  # noqa: E402
from leod_amd.modules.utils.detection import BackboneFeatureSelector, Mode, DATA_KEY  # noqa: E402
from leod_amd.utils.evaluation.prophesee.io.box_loading import to_prophesee  # noqa: E402


class EventSeqResult:
    """Detections and labels of one recording, keyed by frame index, across TTA views (:64-195)."""

    def __init__(self, path: str, img_hw: Tuple[int, int], postproc_cfg):
        self.path, self.img_hw, self.postproc_cfg = path, tuple(img_hw), postproc_cfg
        self._eoe, self._aug = False, False
        self.ev_idx_2_pred: Dict[int, th.Tensor] = {}
        self.ev_idx_2_gt: Dict[int, ObjectLabels] = {}

    def update(self, is_hflip: bool, is_tflip: bool, preds: List, gts: List, ev_idx: List[int], is_last_sample: bool,
               tflip_offset: int) -> None:
        """One view of one batch.  Only frames that carry labels count; the labels themselves are taken from the plain
        view, and only the plain view can end the recording."""
        keep = [k for k, gt in enumerate(gts) if isinstance(gt, ObjectLabels) and len(gt) > 0]
        boxes = [preds[k].get_labels_as_tensors(format_='prophesee') if isinstance(preds[k], ObjectLabels) else preds[k]
                 for k in keep]
        frames = [ev_idx[k] for k in keep]
        if is_hflip:
            boxes = self._hflip_bbox(boxes)
        if is_tflip:                                   # a reversed frame i shows what the plain frame i + offset shows
            frames = [f + tflip_offset for f in frames]
        if is_hflip or is_tflip:
            self._aug = True
        else:
            assert not self._eoe, 'Cannot update a finished sequence.'
            for f, k in zip(frames, keep):
                assert f not in self.ev_idx_2_gt, 'Duplicate label.'
                assert self.img_hw == tuple(gts[k].input_size_hw), 'Inconsistent image size.'
                self.ev_idx_2_gt[f] = gts[k]
        for f, b in zip(frames, boxes):
            self.ev_idx_2_pred[f] = b if f not in self.ev_idx_2_pred else th.cat([self.ev_idx_2_pred[f], b], dim=0)
        if not (is_hflip or is_tflip):
            self._eoe = is_last_sample

    def _hflip_bbox(self, bboxes: List) -> List[th.Tensor]:
        """Back to the un-flipped frame: x1 <- W - 1 - x1 - w (in place on tensors, like the reference :130-147)."""
        out = []
        for b in bboxes:
            if isinstance(b, ObjectLabels):
                b.flip_lr_()
                b = b.get_labels_as_tensors(format_='prophesee')
            else:
                w = b[:, 2] - b[:, 0]
                b[:, 0] = self.img_hw[1] - 1 - b[:, 0] - w
                b[:, 2] = b[:, 0] + w
            out.append(b)
        return out

    def aggregate_results(self):
        """-> (label records, detection records) per labelled frame in frame order; multi-view detections NMS-merged."""
        assert self._eoe, 'Cannot aggregate results before the sequence ends.'
        frames = sorted(self.ev_idx_2_pred)
        assert frames == sorted(self.ev_idx_2_gt), 'Missing labels.'
        preds = [self.ev_idx_2_pred[f] for f in frames]
        if self._aug:
            preds = tta_postprocess(preds, conf_thre=self.postproc_cfg.confidence_threshold,
                                    nms_thre=self.postproc_cfg.nms_threshold)
        return to_prophesee([self.ev_idx_2_gt[f] for f in frames], preds)

    @property
    def aug(self) -> bool:
        return self._aug

    @property
    def eoe(self) -> bool:
        return self._eoe


class TTAModule(Module):
    """Test-only wrapper of the detection module (:198-387): ``test_step`` collects per-recording results over the TTA
    views the loader delivers (time-reversed copies) or that are made here (horizontal flip, concatenated on the batch
    dimension), ``on_test_epoch_end`` merges them and runs the Prophesee evaluator."""

    control_tensors_on_host = True          # flags / indices / masks of a chunk stay host tensors (Module.transfer_batch_to_device)

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.ev_path_2_ev_pred: Dict[str, EventSeqResult] = {}
        self.tta_cfg = self.full_config.tta
        self.postproc_cfg = self.mdl_config.postprocess

    def get_data_from_batch(self, batch: Any):
        """The flipped copy doubles the batch; labels are duplicated unflipped -- only the plain view's are used.  The
        event tensors stay uint8 / unpadded: the stem kernel casts and pads the right / bottom of the flipped frame, which
        is where the reference pads it (:239)."""
        data = batch[DATA_KEY]
        assert DataType.AUGM_STATE not in data, 'should not apply data augmentation in testing'
        ev = th.stack(data[DataType.EV_REPR])
        B = ev.shape[1]
        data['is_hflip'] = np.zeros(B, dtype=bool)
        if self.tta_cfg.enable and self.tta_cfg.hflip:
            ev = th.cat([ev, th.flip(ev, dims=[-1])], dim=1)
            twice = {k: th.cat([data[k]] * 2, dim=-1) for k in (DataType.IS_FIRST_SAMPLE, DataType.IS_LAST_SAMPLE, DataType.IS_REVERSED)}
            twice[DataType.EV_IDX] = [th.cat([idx] * 2, dim=-1) for idx in data[DataType.EV_IDX]]
            twice[DataType.PATH] = data[DataType.PATH] * 2
            twice[DataType.OBJLABELS_SEQ] = [labels + labels for labels in data[DataType.OBJLABELS_SEQ]]
            twice['is_hflip'] = np.array([False] * B + [True] * B, dtype=bool)
            data = twice
        data[DataType.EV_REPR] = list(ev.unbind(0))
        return data

    @th.no_grad()
    def _test_step_impl(self, batch: Any, mode: Mode):
        data = self.get_data_from_batch(batch)
        worker_id = self.get_worker_id_from_batch(batch)
        assert mode in (Mode.VAL, Mode.TEST)
        assert self.mode_2_sampling_mode[mode] == DatasetSamplingMode.STREAM, 'Should always test on streaming mode event sequences'
        ev_seq, labels_seq = data[DataType.EV_REPR], data[DataType.OBJLABELS_SEQ]
        is_first = data[DataType.IS_FIRST_SAMPLE]
        L, B = len(labels_seq), len(labels_seq[0])
        assert L > 0 and B > 0
        if self.mode_2_batch_size[mode] is None:
            self.mode_2_batch_size[mode] = B
        else:
            assert self.mode_2_batch_size[mode] == B
        rnn = self.mode_2_rnn_states[mode]
        rnn.reset(worker_id=worker_id, indices_or_bool_tensor=is_first)
        states = rnn.get_states(worker_id=worker_id)
        hw = tuple(ev_seq[0].shape[-2:])
        if self.mode_2_hw[mode] is None:
            self.mode_2_hw[mode] = hw
        else:
            assert self.mode_2_hw[mode] == hw
        gts, where = [], []
        for t in range(L):
            cur, idx = labels_seq[t].get_valid_labels_and_batch_indices()
            gts.extend(cur)
            where.extend((t, b) for b in idx)
        in_features = self.mdl.fpn.in_features
        feats = None
        if self.time_batched:                                  # stage-major, all L frames of a stage per launch
            ev = self._stack_frames(ev_seq)
            feats_all, states = self.mdl.backbone.forward_sequence(ev, states)
            if where:
                ridx = self._row_index(tuple(t * B + b for t, b in where), ev.device)
                feats = {k: feats_all[k].permute(0, 2, 3, 1).index_select(0, ridx).permute(0, 3, 1, 2) for k in in_features}
        else:
            selector = BackboneFeatureSelector()
            by_t: Dict[int, List[int]] = {}
            for t, b in where:
                by_t.setdefault(t, []).append(b)
            for t in range(L):
                f_t, states = self.mdl.forward_backbone(x=ev_seq[t], previous_states=states)
                if t in by_t:
                    selector.add_backbone_features(backbone_features={k: f_t[k] for k in in_features}, selected_indices=by_t[t])
            feats = selector.get_batched_backbone_features()
        rnn.save_states_and_detach(worker_id=worker_id, states=states)
        if feats is None:
            assert len(gts) == 0
            return tuple([[]] * 8)
        predictions, _ = self.mdl.forward_detect(backbone_features=feats)
        dets = postprocess(prediction=predictions, num_classes=self.mdl_config.head.num_classes,
                           conf_thre=self.postproc_cfg.confidence_threshold, nms_thre=self.postproc_cfg.nms_threshold,
                           pad=th.zeros((0, 7), dtype=predictions.dtype, device=predictions.device))
        all_preds = [[1.0] * L for _ in range(B)]              # placeholders mark frames without labels
        all_gts = [[1.0] * L for _ in range(B)]
        for det, gt, (t, b) in zip(dets, gts, where):
            all_preds[b][t], all_gts[b][t] = det, gt
        ev_idx = th.stack(data[DataType.EV_IDX]).transpose(1, 0).cpu().numpy().tolist()
        return (all_preds, all_gts, data[DataType.PATH], ev_idx, is_first.cpu().numpy().tolist(),
                data[DataType.IS_LAST_SAMPLE].cpu().numpy().tolist(), data['is_hflip'].tolist(),
                data[DataType.IS_REVERSED].cpu().numpy().tolist())

    def training_step(self, batch: Any, batch_idx: int = 0):
        raise NotImplementedError('Only used for testing')

    def validation_step(self, batch: Any, batch_idx: int = 0):
        raise NotImplementedError('Only used for testing')

    def test_step(self, batch: Any, batch_idx: int = 0) -> None:
        out = self._test_step_impl(batch=batch, mode=Mode.TEST)
        for preds, gts, path, ev_idx, first, last, hflip, tflip in zip(*out):
            if not path:                                      # padding slot of the streaming loader
                assert not first and not last and all(i == -1 for i in ev_idx), 'invalid empty data'
                continue
            key = os.path.basename(path)
            if key not in self.ev_path_2_ev_pred:
                assert first, 'should load the first sample first'
                self.ev_path_2_ev_pred[key] = EventSeqResult(path=key, img_hw=tuple(self.dst_config.ev_repr_hw),
                                                             postproc_cfg=self.postproc_cfg)
            self.ev_path_2_ev_pred[key].update(is_hflip=hflip, is_tflip=tflip, preds=preds, gts=gts, ev_idx=ev_idx,
                                               is_last_sample=last,
                                               tflip_offset=self.dst_config.data_augmentation.tflip_offset)

    def on_test_epoch_end(self):
        mode = Mode.TEST
        assert not self.mode_2_psee_evaluator[mode].has_data()
        for result in self.ev_path_2_ev_pred.values():
            labels, preds = result.aggregate_results()
            self.mode_2_psee_evaluator[mode].add_labels(labels)
            self.mode_2_psee_evaluator[mode].add_predictions(preds)
        return self.run_psee_evaluator(mode=mode)
