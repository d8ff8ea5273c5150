"""Runtime-derived config fields, same results as the reference's config/modifier.py:10-131:
padded input resolution (multiple of 32*partition_split_32), partition size, class count and the
gen1->gen4 remapping of per-class thresholds."""
import math

from .dictconfig import open_dict

_HW = {'gen1': (240, 304), 'gen4': (720, 1280)}


def dataloading_hw(dst_cfg):
    hw = _HW[dst_cfg.name]
    return tuple(x // 2 for x in hw) if dst_cfg.downsample_by_factor_2 else hw


def _remap_gen4(t):
    """gen1 order (car, ped) -> gen4 order (ped, cyc, car); cyclists share the pedestrian value."""
    if not isinstance(t, float) and t is not None and len(t) == 2:
        return type(t)([t[1], t[1], t[0]])
    return t


def dynamically_modify_train_config(config):
    with open_dict(config):
        dst = config.dataset
        if dst.name not in _HW:
            raise AssertionError(f'dataset {dst.name} not supported')
        num_classes = 2 if dst.name == 'gen1' else 3
        dst.num_classes = num_classes
        hw = dataloading_hw(dst)
        dst.ev_repr_hw = hw
        if not config.get('is_train', True) and 'tta' in config:
            dst.tta = config.tta
        mdl = config.model
        bb = mdl.backbone
        if bb.name != 'MaxViTRNN':
            raise NotImplementedError(bb.name)
        split = bb.partition_split_32
        assert split in (1, 2, 4)
        mult = 32 * split
        mdl_hw = tuple(math.ceil(x / mult) * mult for x in hw)
        bb.in_res_hw = mdl_hw
        part = tuple(x // mult for x in mdl_hw)
        assert (mdl_hw[0] // 32) % part[0] == 0 and (mdl_hw[1] // 32) % part[1] == 0
        bb.stage.attention.partition_size = part
        bb.vit_size = {64: 'base', 48: 'small', 32: 'tiny'}.get(bb.embed_dim, f'dim{bb.embed_dim}')
        mdl.head.num_classes = num_classes
        if 'pseudo_label' in mdl and dst.name == 'gen4':
            mdl.pseudo_label.obj_thresh = _remap_gen4(mdl.pseudo_label.obj_thresh)
            mdl.pseudo_label.cls_thresh = _remap_gen4(mdl.pseudo_label.cls_thresh)
        thr = mdl.head.get('ignore_bbox_thresh', None)
        if thr and dst.name == 'gen4':
            mdl.head.ignore_bbox_thresh = _remap_gen4(thr)
    return config
