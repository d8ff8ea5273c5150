"""PAFPN neck on channels-last maps, HIP-backed (mirror of the reference's
models/detection/yolox_extension/models/yolo_pafpn.py:18-140; same module tree / state-dict keys)."""
from typing import Dict, Optional, Tuple

import torch as th
import torch.nn as nn

from leod_amd import functions as Fn
from ...yolox.models.network_blocks import BaseConv, CSPLayer, DWConv


def upsample2_nhwc(x: th.Tensor) -> th.Tensor:
    """nearest-exact x2 (yolo_pafpn.py:47) on [B,H,W,C]: out[2y+dy, 2x+dx] = in[y, x]."""
    B, H, W, C = x.shape
    return x[:, :, None, :, None, :].expand(B, H, 2, W, 2, C).reshape(B, 2 * H, 2 * W, C)


class YOLOPAFPN(nn.Module):
    def __init__(self, depth: float = 1.0, in_stages: Tuple[int, ...] = (2, 3, 4),
                 in_channels: Tuple[int, ...] = (256, 512, 1024), depthwise: bool = False, act: str = "silu",
                 compile_cfg: Optional[Dict] = None):
        super().__init__()
        assert len(in_stages) == len(in_channels) == 3
        Conv = DWConv if depthwise else BaseConv          # bottom-up convs and the Bottleneck 3x3s (yolo_pafpn.py:37)
        self.in_features = tuple(in_stages)
        self.in_channels = tuple(in_channels)
        n = round(3 * depth)
        c0, c1, c2 = in_channels
        self.lateral_conv0 = BaseConv(c2, c1, 1, 1, act=act)
        self.C3_p4 = CSPLayer(2 * c1, c1, n, False, depthwise=depthwise, act=act)
        self.reduce_conv1 = BaseConv(c1, c0, 1, 1, act=act)
        self.C3_p3 = CSPLayer(2 * c0, c0, n, False, depthwise=depthwise, act=act)
        self.bu_conv2 = Conv(c0, c0, 3, 2, act=act)
        self.C3_n3 = CSPLayer(2 * c0, c1, n, False, depthwise=depthwise, act=act)
        self.bu_conv1 = Conv(c1, c1, 3, 2, act=act)
        self.C3_n4 = CSPLayer(2 * c1, c2, n, False, depthwise=depthwise, act=act)

    def forward_nhwc(self, x2, x1, x0):
        fpn_out0 = self.lateral_conv0.forward_nhwc(x0)
        f_out0 = self.C3_p4.forward_nhwc(Fn.cat2_nhwc(fpn_out0, x1, up=True))        # cat([upsample(fpn_out0), x1]) in one launch
        fpn_out1 = self.reduce_conv1.forward_nhwc(f_out0)
        pan_out2 = self.C3_p3.forward_nhwc(Fn.cat2_nhwc(fpn_out1, x2, up=True))
        p_out1 = Fn.cat2_nhwc(self.bu_conv2.forward_nhwc(pan_out2), fpn_out1)
        pan_out1 = self.C3_n3.forward_nhwc(p_out1)
        p_out0 = Fn.cat2_nhwc(self.bu_conv1.forward_nhwc(pan_out1), fpn_out0)
        pan_out0 = self.C3_n4.forward_nhwc(p_out0)
        return pan_out2, pan_out1, pan_out0

    def forward(self, input: Dict[int, th.Tensor]):
        """input: {stage: [B,C,h,w]} -> tuple of 3 NCHW-logical maps (channels-last memory)."""
        x2, x1, x0 = (Fn.to_nhwc(input[f]) for f in self.in_features)
        return tuple(Fn.as_nchw(t) for t in self.forward_nhwc(x2, x1, x0))
