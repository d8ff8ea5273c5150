"""Detection module with the reference's LightningModule method surface (modules/detection.py:25-594):
``setup, forward, get_data_from_batch, training_step, validation_step, test_step, configure_optimizers,
predict_one_seq, load_weight``.  pytorch_lightning is optional: without it the class is a plain
``nn.Module`` with the same methods, and ``leod_amd.engine.TrainEngine`` drives it.

Hot-loop differences from the reference (same results): the event tensor stays uint8 and unpadded (the cast
and the bottom/right zero padding are folded into the stem kernel instead of materialising a
[L,B,20,256,320] fp32 tensor, detection.py:132-135), and the detection head/loss never synchronises with
the host."""
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch as th

try:  # pragma: no cover
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except ImportError:
    _Base = th.nn.Module

from leod_amd.data.genx_utils.labels import ObjectLabels
from leod_amd.data.utils.types import DataType, DatasetSamplingMode, ObjDetOutput
from leod_amd.models.detection.yolox.utils.boxes import postprocess
from leod_amd.models.detection.yolox_extension.models.detector import YoloXDetector
from leod_amd.utils.evaluation.prophesee.evaluator import PropheseeEvaluator
from leod_amd.utils.evaluation.prophesee.io.box_loading import to_prophesee
from leod_amd.utils.padding import InputPadderFromShape
from .utils.detection import (BackboneFeatureSelector, EventReprSelector, Mode, RNNStates, mode_2_string,
                              merge_mixed_batches, WORKER_ID_KEY, DATA_KEY)
from .utils.ssod import get_subsample_label_idx


class Module(_Base):
    def __init__(self, full_config, ssod: bool = False):
        super().__init__()
        self.full_config = full_config
        self.mdl_config = full_config.model
        self.num_classes = self.mdl_config.head.num_classes
        self.in_res_hw = tuple(self.mdl_config.backbone.in_res_hw)
        self.input_padder = InputPadderFromShape(desired_hw=self.in_res_hw)
        self.mdl = YoloXDetector(self.mdl_config, ssod=ssod)
        self.dst_config = full_config.dataset
        self.label_subsample_idx = get_subsample_label_idx(L=self.dst_config.sequence_length,
                                                           use_every=self.mdl_config.get('use_label_every', 1))
        self.mode_2_rnn_states: Dict[Mode, RNNStates] = {m: RNNStates() for m in Mode}
        self.mode_2_hw: Dict[Mode, Optional[Tuple[int, int]]] = {m: None for m in Mode}
        self.mode_2_batch_size: Dict[Mode, Optional[int]] = {m: None for m in Mode}
        self.mode_2_sampling_mode: Dict[Mode, Any] = {}
        self.mode_2_psee_evaluator: Dict[Mode, PropheseeEvaluator] = {}
        self.started_training = True
        self.train_vis_every = int(1e9)
        self.train_eval_every = None

    # ---- Lightning-compatible plumbing -----------------------------------------------------------------
    def setup(self, stage: Optional[str] = None) -> None:
        dst = self.full_config.dataset
        new_evaluator = lambda: PropheseeEvaluator(dataset=dst.name, downsample_by_2=dst.downsample_by_factor_2)  # noqa: E731
        if stage == 'fit':
            self.train_config = self.full_config.training
            metrics_cfg = self.full_config.get('logging', {}).get('train', {}).get('metrics', None)
            if metrics_cfg is not None and metrics_cfg.get('compute', False):      # detection KPIs on the training stream
                self.train_eval_every = metrics_cfg.get('detection_metrics_every_n_steps', None)
                self.mode_2_psee_evaluator[Mode.TRAIN] = new_evaluator()
            self.mode_2_psee_evaluator[Mode.VAL] = new_evaluator()
            self.mode_2_sampling_mode[Mode.TRAIN] = dst.train.sampling
            self.mode_2_sampling_mode[Mode.VAL] = dst.eval.sampling
            self.started_training = False
        elif stage == 'validate':
            self.mode_2_psee_evaluator[Mode.VAL] = new_evaluator()
            self.mode_2_sampling_mode[Mode.VAL] = dst.eval.sampling
        elif stage in ('test', 'predict'):
            self.mode_2_psee_evaluator[Mode.TEST] = new_evaluator()
            self.mode_2_sampling_mode[Mode.TEST] = dst.eval.sampling
        else:
            raise NotImplementedError(f'Stage {stage} not implemented.')

    def forward(self, event_tensor, previous_states=None, retrieve_detections: bool = True, targets=None):
        return self.mdl(x=event_tensor, previous_states=previous_states, retrieve_detections=retrieve_detections,
                        targets=targets)

    def get_worker_id_from_batch(self, batch: Any) -> int:
        return batch[WORKER_ID_KEY]

    def get_data_from_batch(self, batch: Any):
        """Event reprs are kept as loaded ([B,C,H,W] uint8 or float, unpadded): the stem kernel pads/casts on the fly.
        Pseudo labels outside ``label_subsample_idx`` are dropped, GT never (reference :136-147)."""
        data = batch[DATA_KEY]
        if not self.training:
            return data
        seq = data[DataType.OBJLABELS_SEQ]
        for tidx in range(len(seq)):
            if tidx not in self.label_subsample_idx:
                seq[tidx].set_non_gt_labels_to_none_()
        return data

    # ---- the hot loop -----------------------------------------------------------------------------------
    def _run_sequence(self, mode: Mode, data, worker_id: int, training: bool, ignore_kwargs=None):
        ev_seq = data[DataType.EV_REPR]
        labels_seq = data[DataType.OBJLABELS_SEQ]
        is_first = data[DataType.IS_FIRST_SAMPLE]
        rnn = self.mode_2_rnn_states[mode]
        rnn.reset(worker_id=worker_id, indices_or_bool_tensor=is_first)
        L, B = len(ev_seq), len(labels_seq[0])
        assert L > 0
        if self.mode_2_batch_size[mode] is None:
            self.mode_2_batch_size[mode] = B
        else:
            assert self.mode_2_batch_size[mode] == B
        prev_states = rnn.get_states(worker_id=worker_id)
        selector, ev_selector, obj_labels = BackboneFeatureSelector(), EventReprSelector(), []
        for tidx in range(L):
            ev = ev_seq[tidx]
            if self.mode_2_hw[mode] is None:
                self.mode_2_hw[mode] = tuple(ev.shape[-2:])
            else:
                assert self.mode_2_hw[mode] == tuple(ev.shape[-2:])
            feats, states = self.mdl.forward_backbone(x=ev, previous_states=prev_states)
            prev_states = states
            cur, idx = labels_seq[tidx].get_valid_labels_and_batch_indices(**(ignore_kwargs or {}))
            if len(cur) > 0:
                selector.add_backbone_features(backbone_features=feats, selected_indices=idx)
                obj_labels.extend(cur)
                ev_selector.add_ev_repr(ev_repr=ev, selected_indices=idx)
        rnn.save_states_and_detach(worker_id=worker_id, states=prev_states)
        return selector, ev_selector, obj_labels, B

    def training_step(self, batch: Any, batch_idx: int = 0, log: bool = True):
        batch = merge_mixed_batches(batch)
        data = self.get_data_from_batch(batch)
        worker_id = self.get_worker_id_from_batch(batch)
        self.started_training = True
        ign = dict(ignore=self.mdl_config.get('ignore_image', False),
                   ignore_label=self.mdl_config.head.get('ignore_label', 1024))
        selector, _, obj_labels, B = self._run_sequence(Mode.TRAIN, data, worker_id, True, ign)
        assert len(obj_labels) > 0
        feats = selector.get_batched_backbone_features()
        labels_yolox = ObjectLabels.get_labels_as_batched_tensor(obj_label_list=obj_labels, format_='yolox')
        labels_yolox = labels_yolox.to(device=next(iter(feats.values())).device, dtype=torch.float32)
        predictions, losses = self.mdl.forward_detect(backbone_features=feats, targets=labels_yolox)
        assert losses is not None and 'loss' in losses
        output = {'loss': losses['loss'],
                  'log_dict': {f'{mode_2_string[Mode.TRAIN]}/{k}': v for k, v in losses.items()}}
        if hasattr(self, 'log_dict') and log and _Base is not th.nn.Module:  # pragma: no cover
            self.log_dict(output['log_dict'], on_step=True, on_epoch=True, batch_size=B, sync_dist=False, rank_zero_only=True)
        return output

    @torch.no_grad()
    def _val_test_step_impl(self, batch: Any, mode: Mode):
        data = self.get_data_from_batch(batch)
        worker_id = self.get_worker_id_from_batch(batch)
        assert mode in (Mode.VAL, Mode.TEST)
        selector, ev_selector, obj_labels, _ = self._run_sequence(mode, data, worker_id, False)
        if len(obj_labels) == 0:
            return {ObjDetOutput.SKIP_VIZ: True}
        predictions, _ = self.mdl.forward_detect(backbone_features=selector.get_batched_backbone_features())
        pred_processed = postprocess(prediction=predictions, num_classes=self.num_classes,
                                     conf_thre=self.mdl_config.postprocess.confidence_threshold,
                                     nms_thre=self.mdl_config.postprocess.nms_threshold)
        # Prophesee records [t, x, y, w, h, class_id, class_confidence], (x, y) = top-left corner (reference :384-399)
        labels_proph, preds_proph = to_prophesee(obj_labels, pred_processed)
        if self.started_training and mode in self.mode_2_psee_evaluator:
            self.mode_2_psee_evaluator[mode].add_labels(labels_proph)
            self.mode_2_psee_evaluator[mode].add_predictions(preds_proph)
        return {ObjDetOutput.LABELS_PROPH: labels_proph[-1], ObjDetOutput.PRED_PROPH: preds_proph[-1],
                ObjDetOutput.EV_REPR: ev_selector.get_ev_repr_as_list(start_idx=-1)[0], ObjDetOutput.SKIP_VIZ: False}

    def run_psee_evaluator(self, mode: Mode, log: bool = True, reset_buffer: bool = True, ret_pr_curve: bool = False):
        """Prophesee / COCO KPIs over everything buffered for ``mode`` (reference :409-463); ``{'val/AP': tensor, ...}``.
        With Lightning the dictionary is also logged; without it (or with ``log=False``) it is returned."""
        from warnings import warn
        evaluator = self.mode_2_psee_evaluator.get(mode)
        if evaluator is None:
            warn(f'{mode=} psee_evaluator is None', UserWarning, stacklevel=2)
            return None
        if mode == Mode.VAL:
            assert reset_buffer, 'Not reseting evaluator in validation'
        hw, batch_size = self.mode_2_hw[mode], self.mode_2_batch_size[mode]
        assert hw is not None and batch_size is not None
        if not evaluator.has_data():
            warn(f'{mode=} psee_evaluator no data', UserWarning, stacklevel=2)
            return None
        metrics = evaluator.evaluate_buffer(img_height=hw[0], img_width=hw[1], ret_pr_curve=ret_pr_curve)
        assert metrics is not None
        if reset_buffer:
            evaluator.reset_buffer()
        pr_curves = {k: metrics.pop(k) for k in [k for k in metrics if 'PR' in k]} if ret_pr_curve else None
        device = next(self.parameters()).device
        log_dict = {f'{mode_2_string[mode]}/{k}': torch.as_tensor(v).to(device) for k, v in metrics.items()}
        if log and hasattr(self, 'log_dict') and _Base is not th.nn.Module:  # pragma: no cover
            self.log_dict(log_dict, on_step=False, on_epoch=True, batch_size=batch_size, sync_dist=True)
            return pr_curves
        if ret_pr_curve:
            return pr_curves
        log_dict['batch_size'] = batch_size
        return log_dict

    def on_train_epoch_end(self):
        if Mode.TRAIN in self.mode_2_psee_evaluator and self.train_eval_every is None and self.mode_2_hw[Mode.TRAIN] is not None:
            return self.run_psee_evaluator(mode=Mode.TRAIN)

    def on_validation_epoch_end(self):
        if self.started_training:
            assert self.mode_2_psee_evaluator[Mode.VAL].has_data()
            return self.run_psee_evaluator(mode=Mode.VAL)

    def on_test_epoch_end(self):
        assert self.mode_2_psee_evaluator[Mode.TEST].has_data()
        return self.run_psee_evaluator(mode=Mode.TEST, reset_buffer=False)

    def validation_step(self, batch: Any, batch_idx: int = 0):
        return self._val_test_step_impl(batch=batch, mode=Mode.VAL)

    def test_step(self, batch: Any, batch_idx: int = 0):
        return self._val_test_step_impl(batch=batch, mode=Mode.TEST)

    def configure_optimizers(self) -> Any:
        """With Lightning: torch AdamW + OneCycleLR exactly as the reference (:485-518).  The native path
        (leod_amd.engine.TrainEngine) uses the fused HIP AdamW on the flat parameter buffer instead."""
        tc = self.full_config.training
        opt = th.optim.AdamW(self.mdl.parameters(), lr=tc.learning_rate, weight_decay=tc.weight_decay)
        sp = tc.lr_scheduler
        if not sp.use:
            return opt
        sch = torch.optim.lr_scheduler.OneCycleLR(optimizer=opt, max_lr=tc.learning_rate, div_factor=sp.div_factor,
                                                  final_div_factor=sp.final_div_factor / sp.div_factor,
                                                  total_steps=sp.total_steps, pct_start=sp.pct_start,
                                                  cycle_momentum=False, anneal_strategy='linear')
        return {'optimizer': opt, 'lr_scheduler': {'scheduler': sch, 'interval': 'step', 'frequency': 1, 'strict': True,
                                                   'name': 'learning_rate'}}

    @torch.no_grad()
    def predict_one_seq(self, batch: Any, head_every: int = 128):
        """B=1 full sequence: backbone per timestep, head every ``head_every`` timesteps (reference :520-581)."""
        data = self.get_data_from_batch(batch)
        ev_seq = data[DataType.EV_REPR]
        labels_seq = data[DataType.OBJLABELS_SEQ]
        prev, feats_buf, preds = None, [], []
        L = len(ev_seq)
        for tidx in range(L):
            feats, prev = self.mdl.forward_backbone(x=ev_seq[tidx], previous_states=prev)
            feats_buf.append(feats)
            if (tidx + 1) % head_every == 0 or tidx == L - 1:
                cat = {k: th.cat([f[k] for f in feats_buf]) for k in feats}
                p, _ = self.mdl.forward_detect(backbone_features=cat)
                preds.extend(postprocess(prediction=p, num_classes=self.num_classes,
                                         conf_thre=self.mdl_config.postprocess.confidence_threshold,
                                         nms_thre=self.mdl_config.postprocess.nms_threshold))
                feats_buf = []
        return preds, th.stack([e[0] for e in ev_seq]), [l[0] for l in labels_seq]

    def load_weight(self, ckpt_path: str, strict: bool = True) -> None:
        ckpt = torch.load(ckpt_path, map_location='cpu')
        if 'state_dict' in ckpt:
            ckpt = ckpt['state_dict']
        self.load_state_dict(ckpt, strict=strict)
