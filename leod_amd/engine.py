"""Native training / pseudo-labelling drivers for the LEOD hot path on MI355X (what Lightning's
Trainer does for the reference, train.py:228-251, minus logging/checkpoint UI).

``TrainEngine.step(ev_seq, labels, is_first)`` = one full training step with inputs already on the device:
reset LSTM rows -> T backbone timesteps -> head + SimOTA + loss on labelled frames -> backward (HIP wgrad
kernels accumulate into the flat gradient buffer) -> RCCL all-reduce -> fused value-clip + AdamW ->
OneCycle LR -> detach states.  No host synchronisation happens inside a step.
"""
from typing import Dict, List, Optional, Sequence

import torch

from . import ops
from .parallel import FlatParams, DataParallel, one_cycle_lr
from .models.detection.yolox.utils.boxes import postprocess_padded
from .modules.utils.ssod import pred2label_padded


class TrainEngine:
    def __init__(self, detector: torch.nn.Module, lr=2e-4, weight_decay=0.0, total_steps=400000, pct_start=0.005,
                 div_factor=20, final_div_factor=10000, clip_value=1.0, process_group=None, sync_bn=True):
        self.det = detector
        self.det.train()
        self.flat = FlatParams(detector)
        self.dp = DataParallel(self.flat, process_group, sync_bn=sync_bn)
        self.dp.broadcast_parameters()
        self.hp = dict(lr=lr, weight_decay=weight_decay, total_steps=total_steps, pct_start=pct_start,
                       div_factor=div_factor, final_div_factor=final_div_factor, clip_value=clip_value)
        self.global_step = 0
        self.states = None
        self.last_losses = None

    def current_lr(self):
        h = self.hp
        return one_cycle_lr(self.global_step, h['lr'], h['total_steps'], h['pct_start'], h['div_factor'], h['final_div_factor'])

    def forward_loss(self, ev_seq: torch.Tensor, labels: torch.Tensor, label_tb: Sequence[Sequence[int]],
                     is_first: Optional[torch.Tensor] = None):
        """ev_seq [T,B,C,H,W] uint8/fp32 on the device; ``label_tb[t]`` = batch indices with labels at timestep t (host
        lists, static); labels [B',N,7] yolox targets in (t, b) order, already on the device."""
        T = ev_seq.shape[0]
        if self.states is not None and is_first is not None:
            for h, c in self.states:                    # RNNStates.reset: zero rows that start a new sequence
                h[is_first] = 0
                c[is_first] = 0
        states = self.states
        sel: Dict[int, List[torch.Tensor]] = {}
        for t in range(T):
            feats, states = self.det.forward_backbone(ev_seq[t], states)
            idx = label_tb[t]
            if len(idx):
                for k in self.det.fpn.in_features:
                    v = feats[k].permute(0, 2, 3, 1)    # NHWC view of the channels-last map
                    sel.setdefault(k, []).append(v if len(idx) == v.shape[0] else v[list(idx)])
        feats = {k: torch.cat(v, 0).permute(0, 3, 1, 2) for k, v in sel.items()}
        preds, losses = self.det.forward_detect(feats, targets=labels)
        self._new_states = [(h.detach(), c.detach()) for h, c in states]
        return preds, losses

    def step(self, ev_seq, labels, label_tb, is_first=None):
        self.flat.zero_grad()
        _, losses = self.forward_loss(ev_seq, labels, label_tb, is_first)
        losses['loss'].backward()
        scale = self.dp.all_reduce_gradients()
        self.flat.adamw_step(self.current_lr(), self.hp['weight_decay'], self.hp['clip_value'], grad_scale=scale)
        self.global_step += 1
        self.states = self._new_states
        self.last_losses = losses
        return losses


class PseudoLabelEngine:
    """Inference loop of modules/pseudo_labeler.py:622-770 for sequences already on the device: optional hflip
    copy on the batch dim, backbone over T, one batched head pass over every (t, b) frame, batched NMS and the
    pseudo-label filters -- all device-side, no host sync until the caller reads the counts."""

    def __init__(self, detector: torch.nn.Module, num_classes: int, conf_thre=0.01, nms_thre=0.45, obj_thresh=(0.6, 0.3),
                 cls_thresh=(0.6, 0.3), dataset_name='gen1', downsampled_by_2=False, hflip=True, max_det=256):
        self.det = detector.eval()
        self.nc, self.conf, self.nms = num_classes, conf_thre, nms_thre
        self.obj_thresh, self.cls_thresh = list(obj_thresh), list(cls_thresh)
        self.dataset_name, self.ds2, self.hflip, self.max_det = dataset_name, downsampled_by_2, hflip, max_det
        self.states = None

    @torch.no_grad()
    def step(self, ev_seq: torch.Tensor, is_first: Optional[torch.Tensor] = None):
        """ev_seq [T,B,C,H,W] -> (labels [T*B', max_det, 8], counts [T*B'], detections, det counts); B' = 2B with hflip."""
        if self.hflip:
            ev_seq = torch.cat([ev_seq, torch.flip(ev_seq, dims=[-1])], dim=1)
            if is_first is not None:
                is_first = torch.cat([is_first, is_first])
        T = ev_seq.shape[0]
        if self.states is not None and is_first is not None:
            for h, c in self.states:
                h[is_first] = 0
                c[is_first] = 0
        states = self.states
        per_t: Dict[int, List[torch.Tensor]] = {}
        for t in range(T):
            feats, states = self.det.forward_backbone(ev_seq[t], states)
            for k in self.det.fpn.in_features:
                per_t.setdefault(k, []).append(feats[k].permute(0, 2, 3, 1))
        self.states = states
        feats = {k: torch.cat(v, 0).permute(0, 3, 1, 2) for k, v in per_t.items()}
        preds, _ = self.det.forward_detect(feats)
        det, cnt = postprocess_padded(preds, self.nc, self.conf, self.nms, max_det=self.max_det)
        lab, lcnt = pred2label_padded(det, cnt, self.obj_thresh, self.cls_thresh, self.dataset_name, self.ds2)
        return lab, lcnt, det, cnt
