from typing import Tuple

from leod_amd.config import to_container
from .yolo_pafpn import YOLOPAFPN
from ...yolox.models.yolo_head import YOLOXHead


def build_yolox_head(head_cfg, in_channels: Tuple[int, ...], strides: Tuple[int, ...], ssod: bool = False):
    assert not ssod
    d = to_container(head_cfg)
    d.pop('name')
    d.pop('version', None)
    d.update(in_channels=in_channels, strides=strides, compile_cfg=d.pop('compile', None))
    return YOLOXHead(**d)


def build_yolox_fpn(fpn_cfg, in_channels: Tuple[int, ...]):
    d = to_container(fpn_cfg)
    name = d.pop('name')
    if name not in {'PAFPN', 'pafpn'}:
        raise NotImplementedError(name)
    d.update(in_channels=in_channels, compile_cfg=d.pop('compile', None))
    return YOLOPAFPN(**d)
