"""Detection module with the reference's LightningModule method surface (modules/detection.py:25-594):
``setup, forward, get_data_from_batch, training_step, validation_step, test_step, configure_optimizers,
predict_one_seq, load_weight``.  pytorch_lightning is optional: without it the class is a plain
``nn.Module`` with the same methods, driven by ``leod_amd.optim.fit_step`` (what Lightning's automatic
optimisation does with one batch).

This surface IS the fast path: ``training_step`` runs the stage-major time-batched schedule, and
``configure_optimizers`` returns ``leod_amd.optim.FlatAdamW`` (one fused clip + AdamW launch on the flat parameter
buffer, the data-parallel all-reduce and SyncBatchNorm switch included) plus torch's own ``OneCycleLR``.

Hot-loop differences from the reference (same results): the event tensor stays uint8 and unpadded (the cast
and the bottom/right zero padding are folded into the stem kernel instead of materialising a
[L,B,20,256,320] fp32 tensor, detection.py:132-135), the L timesteps are evaluated stage by stage instead of
frame by frame, and the detection head/loss never synchronises with the host."""
import os
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch as th

try:  # pragma: no cover
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except ImportError:
    _Base = th.nn.Module

from leod_amd import ops
from leod_amd.functions import WgradSide
from leod_amd.data.genx_utils.labels import ObjectLabels, SparselyBatchedObjectLabels
from leod_amd.data.utils.types import DataType, DatasetSamplingMode, ObjDetOutput
from leod_amd.models.detection.yolox.utils.boxes import postprocess
from leod_amd.models.detection.yolox_extension.models.detector import YoloXDetector
from leod_amd.utils.evaluation.prophesee.evaluator import PropheseeEvaluator
from leod_amd.utils.evaluation.prophesee.io.box_loading import to_prophesee
from leod_amd.utils.padding import InputPadderFromShape
from leod_amd.utils.host import bound_host_threads
from .utils.detection import (BackboneFeatureSelector, Mode, RNNStates, mode_2_string,
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
        # 'batched' (default): stage-major, all L timesteps of a stage per launch; anything else: the reference's
        # timestep-major loop over ``forward_backbone`` (same values, ~5x more and smaller launches)
        self.time_batched = True          # stage-major time-batched schedule (False: the reference's timestep-major loop, for state inspection)
        self.wgrad_side = True            # weight-gradient kernels on a side HIP stream (False: on the launch stream)
        self.plan_head_eager = False      # option for N > 1: planned backbone, eager PAFPN + head (see _head_eager)
        self._row_idx_cache: Dict[Any, th.Tensor] = {}
        # launch plans (modules/step_plan.py): a training-step geometry seen twice is captured once and replayed from C afterwards --
        # this build's counterpart of the reference's ``backbone.compile`` switch (torch.compile(mode='reduce-overhead') = CUDA graphs,
        # modules/detection.py:43-44).  LEOD_PLAN=0 keeps every step eager.
        from .step_plan import TrainStepPlans
        self.plan_mode = os.environ.get('LEOD_PLAN', '1') == '1'
        self._plans = TrainStepPlans()
        self._flat = None                                # FlatParams of configure_optimizers (owner of the bf16 weight shadow)
        self._pin_ring: List[Any] = []
        self._pin_next = 0

    # ---- Lightning-compatible plumbing -----------------------------------------------------------------
    def setup(self, stage: Optional[str] = None) -> None:
        # training.precision (reference: Trainer(precision=config.training.precision), train.py:240): 16 -> bf16 contraction
        # operands with fp32 accumulation / statistics / state, 32 -> fp32 end to end
        ops.set_precision(ops.precision_from_config(self.full_config.get('training', None)))
        bound_host_threads()            # the launch thread must not be throttled by a spinning intra-op pool (utils/host.py)
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
        states = data.get(DataType.AUGM_STATE, None)
        if states is not None and any(s.apply_h_flip or s.zoom_in.active or s.zoom_out.active for s in states):
            # spatial augmentation of the whole batch in ONE gather launch (the loaders transformed the labels on the host
            # and left the frames untouched; the reference flips / zooms every frame in the dataloader workers)
            from leod_amd.data.utils.augmentor import augment_events
            f0 = data[DataType.EV_REPR][0]
            if not (th.is_tensor(f0) and f0.is_cuda):
                # the labels of this batch ARE transformed already: training on them over un-warped frames would be silently wrong
                from leod_amd._lib import LeodHipError
                raise LeodHipError('batch carries an active AUGM_STATE but its frames are not on the device: move the batch first '
                                   '(Module.transfer_batch_to_device / DevicePrefetcher); there is no host fallback for the frame warp')
            ev = augment_events(self._stack_frames(data[DataType.EV_REPR]).contiguous(), states)
            data[DataType.EV_REPR] = [ev[t] for t in range(ev.shape[0])]
        return data

    # ---- the hot loop -----------------------------------------------------------------------------------
    @staticmethod
    def _stack_frames(ev_seq) -> th.Tensor:
        """The loader hands over L frame tensors [B,C,H,W] (reference: a Python list, detection.py:132); when they are
        consecutive slices of one [L,B,C,H,W] buffer (``leod_amd.data`` loaders, ``HostFeeder``) that buffer is used as is,
        otherwise the frames are stacked (one device copy of the uint8 batch)."""
        if th.is_tensor(ev_seq):
            return ev_seq
        f0 = ev_seq[0]
        base = f0._base
        step = f0.numel() * f0.element_size()
        if (base is not None and tuple(base.shape) == (len(ev_seq),) + tuple(f0.shape) and base.is_contiguous()
                and base.dtype == f0.dtype and base.data_ptr() == f0.data_ptr()
                and all(e.shape == f0.shape and e.is_contiguous() and e.data_ptr() == f0.data_ptr() + t * step
                        for t, e in enumerate(ev_seq))):
            return base
        return th.stack(list(ev_seq))

    def _run_sequence(self, mode: Mode, data, worker_id: int, ignore_kwargs=None):
        """reset states -> backbone over the L frames -> features of the labelled frames (reference: the ``for tidx in
        range(L)`` loop of detection.py:188-226).  Default schedule: stage-major and time-batched
        (``RNNDetector.forward_sequence``: every per-frame layer sees L*B frames per launch, only the ConvLSTM walks over
        t) -- same values as the per-timestep loop, which stays available (``self.time_batched = False``)."""
        if self._flat is not None:
            self._flat.ensure_shadow()                   # bf16 copy of the flat parameters for the Linear kernels (stale after every optimiser step)
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
        hw = tuple(ev_seq[0].shape[-2:])
        if self.mode_2_hw[mode] is None:
            self.mode_2_hw[mode] = hw
        else:
            assert self.mode_2_hw[mode] == hw
        prev_states = rnn.get_states(worker_id=worker_id)
        obj_labels, where = [], []
        for tidx in range(L):
            cur, idx = labels_seq[tidx].get_valid_labels_and_batch_indices(**(ignore_kwargs or {}))
            obj_labels.extend(cur)
            where.extend((tidx, b) for b in idx)
        in_features = self.mdl.fpn.in_features
        if self.time_batched:
            ev = self._stack_frames(ev_seq)
            feats = None
            if where:
                rows = self._row_index(tuple(t * B + b for t, b in where), ev.device)
                _, states, feats = self.mdl.backbone.forward_sequence(ev, prev_states, select_rows=rows, select_stages=tuple(in_features))
            else:
                _, states = self.mdl.backbone.forward_sequence(ev, prev_states)
        else:
            selector = BackboneFeatureSelector()
            states = prev_states
            by_t: Dict[int, List[int]] = {}
            for t, b in where:
                by_t.setdefault(t, []).append(b)
            for tidx in range(L):
                f_t, states = self.mdl.forward_backbone(x=ev_seq[tidx], previous_states=states)
                if tidx in by_t:
                    selector.add_backbone_features(backbone_features={k: f_t[k] for k in in_features},
                                                   selected_indices=by_t[tidx])
            feats = selector.get_batched_backbone_features()
        rnn.save_states_and_detach(worker_id=worker_id, states=states)
        return feats, obj_labels, where, B

    def _upload_pinned(self, t: th.Tensor, device) -> th.Tensor:
        """Host tensor -> device through a ring of 4 pinned staging buffers (asynchronous copy on the launch stream; a
        buffer is reused only after the copy that read it has completed -- normally three steps earlier)."""
        n = t.numel()
        if not self._pin_ring or self._pin_ring[0][0].numel() < n or self._pin_ring[0][0].dtype != t.dtype:
            self._pin_ring = [[th.empty(max(2 * n, 4096), dtype=t.dtype).pin_memory(), None] for _ in range(4)]
            self._pin_next = 0
        slot = self._pin_ring[self._pin_next]
        self._pin_next = (self._pin_next + 1) % len(self._pin_ring)
        if slot[1] is not None:
            slot[1].synchronize()
        slot[0][:n].copy_(t.reshape(-1))
        out = slot[0][:n].to(device, non_blocking=True).view(t.shape)
        slot[1] = th.cuda.Event()
        slot[1].record()
        return out

    def _row_index(self, rows, device):
        key = (rows, str(device))
        if key not in self._row_idx_cache:
            if len(self._row_idx_cache) > 256:
                self._row_idx_cache.clear()
            self._row_idx_cache[key] = th.tensor(rows, dtype=th.long, device=device)
        return self._row_idx_cache[key]

    def training_step(self, batch: Any, batch_idx: int = 0, log: bool = True):
        batch = merge_mixed_batches(batch)
        data = self.get_data_from_batch(batch)
        worker_id = self.get_worker_id_from_batch(batch)
        self.started_training = True
        ign = dict(ignore=self.mdl_config.get('ignore_image', False),
                   ignore_label=self.mdl_config.head.get('ignore_label', 1024))
        device = data[DataType.EV_REPR][0].device
        if self.plan_mode and self.time_batched and device.type == 'cuda' and th.is_grad_enabled() and self._plans.allowed():
            planned = self._training_step_planned(data, worker_id, ign, log)
            if planned is not None:
                return planned
            self._plans.eager_steps += 1               # EVERY eager fallback of a plan-mode step is counted here, whatever its reason
        ops.StatArena.begin_step(device)             # one memset for every BatchNorm statistic accumulator of the step
        feats, obj_labels, _, B = self._run_sequence(Mode.TRAIN, data, worker_id, ign)
        assert len(obj_labels) > 0
        labels_yolox = ObjectLabels.get_labels_as_batched_tensor(obj_label_list=obj_labels, format_='yolox')
        if labels_yolox.device != device:            # host labels: pinned + asynchronous, the launch thread never waits for the GPU
            labels_yolox = self._upload_pinned(labels_yolox, device)
        predictions, losses = self.mdl.forward_detect(backbone_features=feats, targets=labels_yolox.to(torch.float32))
        assert losses is not None and 'loss' in losses
        # the weight-gradient kernels of the backward pass that follows run on a side HIP stream until the optimiser joins it
        WgradSide.active = self.wgrad_side and torch.is_grad_enabled() and not torch.cuda.is_current_stream_capturing()
        output = {'loss': losses['loss'],
                  'log_dict': {f'{mode_2_string[Mode.TRAIN]}/{k}': v for k, v in losses.items()}}
        if hasattr(self, 'log_dict') and log and _Base is not th.nn.Module:  # pragma: no cover
            self.log_dict(output['log_dict'], on_step=True, on_epoch=True, batch_size=B, sync_dist=False, rank_zero_only=True)
        return output

    def _training_step_planned(self, data, worker_id: int, ign, log: bool):
        """The same step through a launch plan (modules/step_plan.py), or None when this batch has to run eagerly: the first
        occurrence of its geometry, host-resident frames, no labelled frame, or a geometry whose capture failed."""
        from .step_plan import PlanLossFn, BackbonePlan, LOSS_KEYS
        ev_seq = data[DataType.EV_REPR]
        labels_seq = data[DataType.OBJLABELS_SEQ]
        is_first = data[DataType.IS_FIRST_SAMPLE]
        L, B = len(ev_seq), len(labels_seq[0])
        obj_labels, where = [], []
        for tidx in range(L):
            cur, idx = labels_seq[tidx].get_valid_labels_and_batch_indices(**ign)
            obj_labels.extend(cur)
            where.extend((tidx, b) for b in idx)
        if not where:
            return None
        ev = self._stack_frames(ev_seq)
        labels_yolox = ObjectLabels.get_labels_as_batched_tensor(obj_label_list=obj_labels, format_='yolox').to(th.float32)
        plans = self._plans
        key = plans.key_of(ev)
        hit = plans.lookup(key)
        if hit is None:
            return None
        mode = Mode.TRAIN
        hw = tuple(ev.shape[-2:])
        if self.mode_2_batch_size[mode] is None or self.mode_2_hw[mode] is None:
            return None                                  # the eager step records them (reference asserts, detection.py:176-179,196-199)
        assert self.mode_2_batch_size[mode] == B and self.mode_2_hw[mode] == hw
        rnn = self.mode_2_rnn_states[mode]
        fresh = False                                    # a plan of this step was captured in this very call
        if hit == 'capture':
            like = rnn.get_states(worker_id) or next(iter(rnn.states.values()), None)
            if like is None:
                return None
            hit = plans.build(key, self, ev, like, self.wgrad_side)
            if hit is None:
                return None
            fresh = True
        bb: BackbonePlan = hit
        if self._head_eager():
            return self._planned_backbone_eager_head(bb, plans, fresh, ev, labels_yolox, where, is_first, worker_id, rnn, B, log)
        # the data-dependent part: PAFPN + head for THIS labelled-frame count (captured the first time the count is seen; nothing has been
        # executed yet, so a failed capture simply leaves the step to the eager path)
        hkey, nmax_pad = plans.head_key_of(len(where), labels_yolox.shape[1])
        hd = bb.head(hkey)
        if hd is None:
            hd = plans.build_head(bb, hkey, self, len(where), nmax_pad, self.wgrad_side)
            fresh = True
        if hd is None or hd == 'eager':
            return None
        self.started_training = True
        rows_host = tuple(t * B + b for t, b in where)
        bb.load_states(rnn, worker_id)
        bb.stage_inputs(ev, rows_host, self._row_index(rows_host + (-1,) * (bb.n_max - len(rows_host)), ev.device), is_first)
        hd.stage_labels(labels_yolox)
        bb.run_forward()
        hd.run_forward()
        rnn.save_states_and_detach(worker_id=worker_id, states=bb.states)
        plans.steps += 1
        plans.replays += 0 if fresh else 1
        if plans.anchor is None or plans.anchor.device != ev.device:
            plans.anchor = th.zeros(1, device=ev.device, requires_grad=True)
        entry = hd
        out6 = entry.losses6.clone()                     # the static loss buffer is overwritten by the next replay
        loss = PlanLossFn.apply(bb, hd, plans.anchor, out6[0])
        losses = {k: out6[i] for i, k in enumerate(LOSS_KEYS)}
        losses['loss'] = loss
        output = {'loss': loss, 'log_dict': {f'{mode_2_string[Mode.TRAIN]}/{k}': v for k, v in losses.items()}}
        if hasattr(self, 'log_dict') and log and _Base is not th.nn.Module:  # pragma: no cover
            self.log_dict(output['log_dict'], on_step=True, on_epoch=True, batch_size=B, sync_dist=False, rank_zero_only=True)
        return output

    def _head_eager(self) -> bool:
        """Planned backbone + eager head (``module.plan_head_eager = True``): an option for N > 1, where every SyncBatchNorm exchange inside a captured
        head is a plan segment boundary (the backbone has no BatchNorm: its plans stay whole apart from the gradient buckets).  Measured
        with every collective issued on a one-rank RCCL communicator (profiles/r05_p_rccl_force_collectives.txt): 16.98 ms per step against
        17.3-17.4 with the captured head and 16.3-16.5 all-eager (15.4 without collectives) -- but 10.5 instead of 4.1 ms of host time per
        step, i.e. it depends on the host the way eager launches do; the captured head stays the default."""
        return bool(self.plan_head_eager)

    def _planned_backbone_eager_head(self, bb, plans, fresh, ev, labels_yolox, where, is_first, worker_id, rnn, B, log):
        from .step_plan import EagerHeadGate
        from leod_amd import functions as Fn
        device = ev.device
        self.started_training = True
        rows_host = tuple(t * B + b for t, b in where)
        bb.load_states(rnn, worker_id)
        bb.stage_inputs(ev, rows_host, self._row_index(rows_host + (-1,) * (bb.n_max - len(rows_host)), device), is_first)
        if labels_yolox.device != device:
            labels_yolox = self._upload_pinned(labels_yolox, device)
        bb.run_forward()
        rnn.save_states_and_detach(worker_id=worker_id, states=bb.states)
        plans.steps += 1
        plans.replays += 0 if fresh else 1
        if plans.anchor is None or plans.anchor.device != device:
            plans.anchor = th.zeros(1, device=device, requires_grad=True)
        ops.StatArena.begin_step(device)             # the eager head's BatchNorm statistic accumulators (the backbone plan has its own arena)
        in_features = tuple(self.mdl.fpn.in_features)
        sel = EagerHeadGate.apply(bb, plans.anchor, self._row_index(rows_host, device), *in_features)
        _, losses = self.mdl.forward_detect(backbone_features={k: Fn.as_nchw(v) for k, v in zip(in_features, sel)},
                                            targets=labels_yolox.to(torch.float32))
        WgradSide.active = self.wgrad_side and torch.is_grad_enabled()
        output = {'loss': losses['loss'], 'log_dict': {f'{mode_2_string[Mode.TRAIN]}/{k}': v for k, v in losses.items()}}
        if hasattr(self, 'log_dict') and log and _Base is not th.nn.Module:  # pragma: no cover
            self.log_dict(output['log_dict'], on_step=True, on_epoch=True, batch_size=B, sync_dist=False, rank_zero_only=True)
        return output

    def backward(self, loss: th.Tensor, *args, **kwargs) -> None:
        """Lightning's backward hook (and what ``leod_amd.optim.fit_step`` calls)."""
        loss.backward(*args, **kwargs)

    @torch.no_grad()
    def _val_test_step_impl(self, batch: Any, mode: Mode):
        data = self.get_data_from_batch(batch)
        worker_id = self.get_worker_id_from_batch(batch)
        assert mode in (Mode.VAL, Mode.TEST)
        feats, obj_labels, where, _ = self._run_sequence(mode, data, worker_id)
        if len(obj_labels) == 0:
            return {ObjDetOutput.SKIP_VIZ: True}
        predictions, _ = self.mdl.forward_detect(backbone_features=feats)
        pred_processed = postprocess(prediction=predictions, num_classes=self.num_classes,
                                     conf_thre=self.mdl_config.postprocess.confidence_threshold,
                                     nms_thre=self.mdl_config.postprocess.nms_threshold, host=True)
        # Prophesee records [t, x, y, w, h, class_id, class_confidence], (x, y) = top-left corner (reference :384-399)
        labels_proph, preds_proph = to_prophesee(obj_labels, pred_processed)
        if self.started_training and mode in self.mode_2_psee_evaluator:
            self.mode_2_psee_evaluator[mode].add_labels(labels_proph)
            self.mode_2_psee_evaluator[mode].add_predictions(preds_proph)
        t_last, b_last = where[-1]
        return {ObjDetOutput.LABELS_PROPH: labels_proph[-1], ObjDetOutput.PRED_PROPH: preds_proph[-1],
                ObjDetOutput.EV_REPR: data[DataType.EV_REPR][t_last][b_last], ObjDetOutput.SKIP_VIZ: False}

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
        """AdamW + OneCycleLR with the reference's hyper-parameters (:485-518).  The optimiser is ``FlatAdamW``: a
        ``torch.optim.Optimizer`` whose ``step`` is the flat-gradient all-reduce (N > 1 ranks) + ONE fused
        value-clip / AdamW launch; the scheduler is torch's ``OneCycleLR`` acting on its ``param_groups``."""
        from leod_amd.optim import FlatAdamW
        tc = self.full_config.training
        opt = FlatAdamW(self.mdl, lr=tc.learning_rate, weight_decay=tc.weight_decay,
                        clip_value=tc.get('gradient_clip_val', None))
        self._flat = opt.flat
        sp = tc.lr_scheduler
        if not sp.use:
            return opt
        sch = torch.optim.lr_scheduler.OneCycleLR(optimizer=opt, max_lr=tc.learning_rate, div_factor=sp.div_factor,
                                                  final_div_factor=sp.final_div_factor / sp.div_factor,
                                                  total_steps=sp.total_steps, pct_start=sp.pct_start,
                                                  cycle_momentum=False, anneal_strategy='linear')
        return {'optimizer': opt, 'lr_scheduler': {'scheduler': sch, 'interval': 'step', 'frequency': 1, 'strict': True,
                                                   'name': 'learning_rate'}}

    # ---- Lightning hooks that keep the Trainer's generic machinery off the flat buffers -----------------------------------
    def configure_gradient_clipping(self, optimizer, gradient_clip_val=None, gradient_clip_algorithm=None) -> None:
        """``Trainer(gradient_clip_val=1.0, gradient_clip_algorithm='value')`` (train.py:236-237): the clip is fused into
        the optimiser kernel, AFTER the cross-rank gradient sum, so the Trainer's own clipping pass is replaced by this."""
        if gradient_clip_val is None:
            return
        assert gradient_clip_algorithm in (None, 'value'), 'the reference clips by value (train.py:237)'
        optimizer = getattr(optimizer, 'optimizer', optimizer)          # LightningOptimizer wrapper
        optimizer.clip_value = float(gradient_clip_val)

    # The inference drivers (PseudoLabeler, TTAModule) read the small per-chunk tensors of a batch (first / last / reversed flags, frame
    # indices, padding masks) on the HOST: on the device every such read is a blocking copy behind the chunk's kernels, i.e. the host
    # cannot run a chunk ahead of the GPU.  They set this, and the batch keeps those tensors where the loader made them; the one the
    # device needs as well (the state-reset mask) is uploaded through pinned memory by the step.
    control_tensors_on_host = False

    def transfer_batch_to_device(self, batch: Any, device, dataloader_idx: int = 0) -> Any:
        """Tensors go to the device asynchronously; box labels stay on the host (their per-frame bookkeeping is host work,
        the padded target tensor is uploaded once per step in ``training_step``)."""
        host_ctl = self.control_tensors_on_host

        def move(o, key=None):
            if th.is_tensor(o):
                if host_ctl and o.device.type == 'cpu' and o.numel() <= 4096:
                    return o                             # flags / indices / masks of the chunk: read by the host bookkeeping (see the class attribute)
                return o.to(device, non_blocking=True)
            if isinstance(o, (ObjectLabels, SparselyBatchedObjectLabels)) or o is None or isinstance(o, (str, int, float, bool)):
                return o
            if key == DataType.EV_REPR and isinstance(o, list) and len(o) and th.is_tensor(o[0]):
                # the L frames of a batch are views of one (pinned) [L,B,C,H,W] buffer: ONE PCIe copy, then views again
                ev = self._stack_frames(o).to(device, non_blocking=True)
                return [ev[t] for t in range(ev.shape[0])]
            if isinstance(o, dict):
                return {k: move(v, k) for k, v in o.items()}
            if isinstance(o, (list, tuple)):
                return type(o)(move(v) for v in o)
            return o
        return move(batch)

    @torch.no_grad()
    def predict_one_seq(self, batch: Any, head_every: int = 128):
        """B=1 full sequence (reference :520-581): the backbone over chunks of ``head_every`` timesteps (time-batched, LSTM
        state carried from chunk to chunk), the head + postprocess once per chunk."""
        mode = Mode.TEST
        data = self.get_data_from_batch(batch)
        worker_id = self.get_worker_id_from_batch(batch)
        ev_seq = data[DataType.EV_REPR]
        labels_seq = data[DataType.OBJLABELS_SEQ]
        is_first = data[DataType.IS_FIRST_SAMPLE]
        L = len(ev_seq)
        assert L > 0 and ev_seq[0].shape[0] == len(labels_seq[0]) == is_first.shape[0] == 1, 'one recording at a time'
        self.mode_2_rnn_states[mode].reset(worker_id=worker_id, indices_or_bool_tensor=is_first)
        hw = tuple(ev_seq[0].shape[-2:])
        if self.mode_2_hw[mode] is None:
            self.mode_2_hw[mode] = hw
        else:
            assert self.mode_2_hw[mode] == hw
        prev, preds = None, []                           # (the reference starts the recording from a zero state, :544)
        in_features = self.mdl.fpn.in_features
        for lo in range(0, L, head_every):
            chunk = ev_seq[lo:lo + head_every]
            if self.time_batched:
                feats_all, prev = self.mdl.backbone.forward_sequence(th.stack(list(chunk)), prev)
                cat = {k: feats_all[k] for k in in_features}
            else:
                buf = []
                for ev in chunk:
                    f_t, prev = self.mdl.forward_backbone(x=ev, previous_states=prev)
                    buf.append(f_t)
                cat = {k: th.cat([f[k] for f in buf]) for k in in_features}
            p, _ = self.mdl.forward_detect(backbone_features=cat)
            preds.extend(postprocess(prediction=p, num_classes=self.num_classes,
                                     conf_thre=self.mdl_config.postprocess.confidence_threshold,
                                     nms_thre=self.mdl_config.postprocess.nms_threshold))
        return preds, th.stack([e[0] for e in ev_seq]), [l[0] for l in labels_seq]

    def load_weight(self, ckpt_path: str, strict: bool = True) -> None:
        ckpt = torch.load(ckpt_path, map_location='cpu')
        if 'state_dict' in ckpt:
            ckpt = ckpt['state_dict']
        self.load_state_dict(ckpt, strict=strict)
