"""YoloXDetector facade (mirror of the reference's
models/detection/yolox_extension/models/detector.py:18-91): recurrent backbone + PAFPN + YOLOX head
behind ``forward_backbone`` / ``forward_detect`` / ``forward`` with unchanged signatures and state-dict keys
(``backbone.*``, ``fpn.*``, ``yolox_head.*``)."""
from typing import Dict, Optional, Tuple, Union

import torch as th

from leod_amd import functions as Fn
from ...recurrent_backbone import build_recurrent_backbone
from .build import build_yolox_fpn, build_yolox_head


class YoloXDetector(th.nn.Module):
    def __init__(self, model_cfg, ssod: bool = False):
        super().__init__()
        self.backbone = build_recurrent_backbone(model_cfg.backbone)
        in_channels = self.backbone.get_stage_dims(tuple(model_cfg.fpn.in_stages))
        self.fpn = build_yolox_fpn(model_cfg.fpn, in_channels=in_channels)
        strides = self.backbone.get_strides(tuple(model_cfg.fpn.in_stages))
        self.yolox_head = build_yolox_head(model_cfg.head, in_channels=in_channels, strides=strides, ssod=ssod)

    def forward_backbone(self, x: th.Tensor, previous_states=None, token_mask: Optional[th.Tensor] = None):
        """x [B,C,H,W] -> ({stage: [B,C,h,w]}, [(h, c)] * 4)."""
        return self.backbone(x, previous_states, token_mask)

    def forward_detect(self, backbone_features: Dict[int, th.Tensor], targets: Optional[th.Tensor] = None,
                       soft_targets: Optional[th.Tensor] = None) -> Tuple[th.Tensor, Union[Dict[str, th.Tensor], None]]:
        """-> (outputs [B, N, 4+1+num_cls], losses dict | None)."""
        assert soft_targets is None
        x2, x1, x0 = (Fn.to_nhwc(backbone_features[f]) for f in self.fpn.in_features)
        if self.training:
            # data-parallel training: the backward pass leaves the PAFPN + head here -> their gradient bucket is exchanged
            x2, x1, x0 = Fn.bucket_boundary(-1, x2, x1, x0)
            Fn.sync_bn_begin(x0.shape[0], x0.device)     # SyncBatchNorm: images over all ranks, once per pass (no-op on one rank)
        fpn_feats = self.fpn.forward_nhwc(x2, x1, x0)
        if self.training:
            assert targets is not None
            return self.yolox_head.forward_nhwc(fpn_feats, targets)
        outputs, losses = self.yolox_head.forward_nhwc(fpn_feats)
        assert losses is None
        return outputs, losses

    def forward(self, x: th.Tensor, previous_states=None, retrieve_detections: bool = True, targets=None):
        backbone_features, states = self.forward_backbone(x, previous_states)
        if not retrieve_detections:
            assert targets is None
            return None, None, states
        outputs, losses = self.forward_detect(backbone_features=backbone_features, targets=targets)
        return outputs, losses, states
