"""Decoupled YOLOX head, HIP-backed (mirror of the reference's
models/detection/yolox/models/yolo_head.py:21-332; same module tree / state-dict keys and loss dict).

Stems and towers are BaseConv kernels; the three 1x1 prediction convs, the grid decode, SimOTA
assignment, the loss terms and their gradient are fused into four kernels without a single host
synchronisation (the reference syncs >= 2x per image, yolo_head.py:455,764)."""
import math
from typing import Dict, Optional

import torch
import torch.nn as nn

from leod_amd import functions as Fn
from leod_amd import ops
from .losses import IOUloss, FocalLoss
from .network_blocks import BaseConv, DWConv

LOSS_KEYS = ('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss', 'num_fg')


class YOLOXHead(nn.Module):
    def __init__(self, num_classes=80, strides=(8, 16, 32), in_channels=(256, 512, 1024), act="silu", depthwise=False,
                 compile_cfg: Optional[Dict] = None, obj_focal_loss=False, bbox_loss_weighting='', ignore_bg_k=-1,
                 reg_weight=5.0, obj_weight=1.0, cls_weight=1.0, ignore_bbox_thresh=None, ignore_label=1024):
        super().__init__()
        if ignore_bg_k is not None and ignore_bg_k > 1:
            raise ValueError('ignore_bg_k is a fraction of the background anchors')
        if bbox_loss_weighting and bbox_loss_weighting.split('-', 1)[0] not in ('obj', 'cls', 'objxcls'):
            raise NotImplementedError(f'Unknow {bbox_loss_weighting=}')
        self.num_classes = num_classes
        self.decode_in_inference = True
        self.cls_convs, self.reg_convs = nn.ModuleList(), nn.ModuleList()
        self.cls_preds, self.reg_preds, self.obj_preds = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.stems = nn.ModuleList()
        hidden_dim = int(256 * (in_channels[-1] / 1024))
        self.hidden_dim = hidden_dim
        self.depthwise = bool(depthwise)
        Conv = DWConv if depthwise else BaseConv          # the tower convs (yolo_head.py:52)
        for c in in_channels:
            self.stems.append(BaseConv(c, hidden_dim, ksize=1, stride=1, act=act))
            self.cls_convs.append(nn.Sequential(Conv(hidden_dim, hidden_dim, 3, 1, act=act),
                                                Conv(hidden_dim, hidden_dim, 3, 1, act=act)))
            self.reg_convs.append(nn.Sequential(Conv(hidden_dim, hidden_dim, 3, 1, act=act),
                                                Conv(hidden_dim, hidden_dim, 3, 1, act=act)))
            self.cls_preds.append(nn.Conv2d(hidden_dim, num_classes, 1, 1, 0))
            self.reg_preds.append(nn.Conv2d(hidden_dim, 4, 1, 1, 0))
            self.obj_preds.append(nn.Conv2d(hidden_dim, 1, 1, 1, 0))
        self.use_l1 = False
        self.obj_focal_loss = bool(obj_focal_loss)
        self.obj_loss_fn = FocalLoss(alpha=0.25, gamma=2.0) if obj_focal_loss else nn.BCEWithLogitsLoss(reduction="none")
        self.cls_loss_fn = nn.BCEWithLogitsLoss(reduction="none")
        self.iou_loss = IOUloss(reduction="mean")
        self.strides = tuple(int(s) for s in strides)
        self.reg_weight, self.obj_weight, self.cls_weight = reg_weight, obj_weight, cls_weight
        self.ignore_bg_k = ignore_bg_k if ignore_bg_k is not None else -1
        self.bbox_loss_weighting = bbox_loss_weighting      # property: parsed and validated once per assignment (below)
        self.ignore_bbox_thresh = ignore_bbox_thresh
        self.ignore_label = ignore_label
        self.last_assignment = None
        self.last_losses6 = None
        self.hw = None
        self.initialize_biases(prior_prob=0.01)

    @property
    def bbox_loss_weighting(self) -> str:
        return self._bbox_loss_weighting

    @bbox_loss_weighting.setter
    def bbox_loss_weighting(self, bbox_loss_weighting) -> None:
        bbox_loss_weighting = bbox_loss_weighting or ''
        if bbox_loss_weighting and bbox_loss_weighting.split('-', 1)[0] not in ('obj', 'cls', 'objxcls'):
            raise NotImplementedError(f'Unknow {bbox_loss_weighting=}')
        self._bbox_loss_weighting = bbox_loss_weighting
        self._blw = None                 # (which confidence, compiled expression of w)
        if bbox_loss_weighting:
            val, expr = bbox_loss_weighting.split('-', 1) if '-' in bbox_loss_weighting else (bbox_loss_weighting, 'w')
            try:
                code = compile(expr, '<model.head.bbox_loss_weighting>', 'eval')
            except SyntaxError as e:
                raise ValueError(f'model.head.bbox_loss_weighting: {expr!r} is not an expression ({e})') from e
            bad = set(code.co_names) - {'w', 'torch', 'math'} - set(dir(torch)) - set(dir(math)) - set(dir(torch.Tensor))
            if bad:
                raise ValueError(f'model.head.bbox_loss_weighting: {expr!r} may only use w, torch and math (found {sorted(bad)})')
            probe = eval(code, {'__builtins__': {}, 'torch': torch, 'math': math}, {'w': torch.full((2, 3), 0.5)})  # noqa: S307
            if not torch.is_tensor(probe) or probe.shape != (2, 3):
                raise ValueError(f'model.head.bbox_loss_weighting: {expr!r} must be elementwise in w (a tensor of w\'s shape)')
            self._blw = (val, code)

    def initialize_biases(self, prior_prob):
        v = -math.log((1 - prior_prob) / prior_prob)
        for conv in list(self.cls_preds) + list(self.obj_preds):
            with torch.no_grad():
                conv.bias.fill_(v)

    @torch.no_grad()
    def _bbox_label_weights(self, labels):
        """``bbox_loss_weighting`` (reference :358-381): 'obj' | 'cls' | 'objxcls', optionally followed by '-<expression of w>'
        ('cls-w**2'), evaluated here on every label ROW (labels [B,N,7]) -- the expression is elementwise, so weighting a foreground
        anchor by the value of its matched row equals the reference's evaluate-after-gather; the kernel does the gather and the
        division by the batch mean.  None when the option is off."""
        if self._blw is None:
            return None
        val, code = self._blw
        obj_conf, cls_conf = labels[:, :, 5], labels[:, :, 6]
        w = obj_conf if val == 'obj' else cls_conf if val == 'cls' else obj_conf * cls_conf
        # the reference evaluates the configured string per step (:376); here the compiled, validated expression without builtins
        w = eval(code, {'__builtins__': {}, 'torch': torch, 'math': math}, {'w': w})  # noqa: S307
        return w.to(torch.float32).contiguous()

    @torch.no_grad()
    def _ignore_bbox(self, labels):
        """Low-confidence pseudo boxes get ``ignore_label`` as class id (reference :383-401; mutates labels)."""
        if not self.ignore_bbox_thresh:
            return labels
        cls_idx, obj_conf, cls_conf = labels[:, :, 0], labels[:, :, 5], labels[:, :, 6]
        ign = torch.zeros_like(cls_idx, dtype=torch.bool)
        for idx, th in enumerate(self.ignore_bbox_thresh):
            ign |= (cls_idx == idx) & ((obj_conf < th) | (cls_conf < th))
        ign &= labels.sum(dim=2) > 0
        labels[:, :, 0] = torch.where(ign, torch.full_like(cls_idx, float(self.ignore_label)), cls_idx)
        return labels

    def _towers(self, xin):
        if self.depthwise:                                # DWConv towers: two BaseConvs per layer, evaluated layer by layer
            feats = []
            for k in range(len(xin)):
                x = self.stems[k].forward_nhwc(xin[k])
                c = r = x
                for d in (0, 1):
                    c = self.cls_convs[k][d].forward_nhwc(c)
                    r = self.reg_convs[k][d].forward_nhwc(r)
                feats += [c, r]
            return feats
        # the three levels are independent: layers of equal depth form one group -- one launch per kernel kind for the six tower convs of a
        # depth (functions.base_conv_group -> ops.conv3x3_group_* / bn_silu_*_group), one SyncBatchNorm exchange per group.  (Rounds 3-5 ran
        # the levels on their own HIP streams in eager steps instead: 15.09 ms against 14.84 for the grouped launches, round 6.)
        n = len(xin)
        xs = Fn.base_conv_group(list(self.stems), list(xin))
        for d in (0, 1):
            ys = Fn.base_conv_group([self.cls_convs[k][d] for k in range(n)] + [self.reg_convs[k][d] for k in range(n)],
                                    (xs + xs) if d == 0 else ys)
        feats = []
        for k in range(n):
            feats += [ys[k], ys[n + k]]
        return feats

    def forward(self, xin, labels=None, pred_probs=None):
        """xin: per-level feature maps, NCHW-logical (or NHWC contiguous when ``xin_is_nhwc``).
        Returns (decoded predictions [B, A, 5+nc] with sigmoid scores, loss dict | None)."""
        assert pred_probs is None
        xin = [Fn.to_nhwc(x) for x in xin]
        return self.forward_nhwc(xin, labels)

    def forward_nhwc(self, xin, labels=None):
        feats = self._towers(xin)
        self.hw = [tuple(f.shape[1:3]) for f in feats[0::2]]
        params = []
        for k in range(len(self.strides)):
            params += [self.cls_preds[k].weight, self.cls_preds[k].bias, self.reg_preds[k].weight, self.reg_preds[k].bias,
                       self.obj_preds[k].weight, self.obj_preds[k].bias]
        if self.training:
            assert labels is not None
            labels = labels.to(dtype=torch.float32).contiguous()
            losses, out = Fn.HeadTailFn.apply(self, labels, *feats, *params)
            self.last_losses6 = losses.detach()              # the six scalars as one tensor (launch plans read them back from here)
            d = {k: losses[i] for i, k in enumerate(LOSS_KEYS)}
            d['loss'] = Fn.PickLossFn.apply(losses)           # the one differentiable entry, without a SelectBackward node
            return out, d
        B = feats[0].shape[0]
        A = sum(h * w for h, w in self.hw)
        out = torch.empty((B, A, 5 + self.num_classes), dtype=torch.float32, device=feats[0].device)
        a0 = 0
        with torch.no_grad():
            for k in range(len(self.strides)):
                cw, cb, rw, rb, ow, ob = params[6 * k:6 * k + 6]
                ops.head_pred_fwd(feats[2 * k], feats[2 * k + 1], cw.view(self.num_classes, -1), cb, rw.view(4, -1), rb,
                                  ow.view(1, -1), ob, None, out, self.strides[k], a0)
                a0 += self.hw[k][0] * self.hw[k][1]
        return out, None
