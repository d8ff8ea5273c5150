"""The reference's Hydra config tree as Python data (same keys/values as config/*.yaml of the
reference: model/maxvit_yolox/default.yaml, model/rnndet.yaml, model/pseudo_labeler.yaml,
experiment/gen{1,4}/{default,tiny,small,base}.yaml, dataset/{base,gen1,gen4}.yaml, general.yaml),
composed without Hydra: ``full_config(dataset='gen1', size='small', model='rnndet', overrides=...)``.
Hydra/OmegaConf users can keep passing their own DictConfig -- only the keys matter."""
import copy
from typing import Mapping, Optional

from .dictconfig import create

_SIZES = {  # experiment/gen*/{tiny,small,base}.yaml
    'tiny': dict(embed_dim=32, dim_head=32, fpn_depth=0.33),
    'small': dict(embed_dim=48, dim_head=24, fpn_depth=0.33),
    'base': dict(embed_dim=64, dim_head=32, fpn_depth=0.67),
}


def _compile_off():
    return dict(enable=False, args=dict(mode='reduce-overhead'))


def model_config(size: str = 'small', name: str = 'rnndet', partition_split_32: int = 1, **over) -> dict:
    s = dict(_SIZES[size]) if size in _SIZES else dict(_SIZES['small'])
    s.update({k: over.pop(k) for k in list(over) if k in ('embed_dim', 'dim_head', 'fpn_depth')})
    cfg = dict(
        name=name,
        backbone=dict(
            name='MaxViTRNN', compile=_compile_off(), input_channels=20, enable_masking=False,
            partition_split_32=partition_split_32, embed_dim=s['embed_dim'], dim_multiplier=[1, 2, 4, 8],
            num_blocks=[1, 1, 1, 1], T_max_chrono_init=[4, 8, 16, 32], stem=dict(patch_size=4),
            stage=dict(
                downsample=dict(type='patch', overlap=True, norm_affine=True),
                attention=dict(use_torch_mha=False, partition_size=None, dim_head=s['dim_head'], attention_bias=True,
                               mlp_activation='gelu', mlp_gated=False, mlp_bias=True, mlp_ratio=4, drop_mlp=0,
                               drop_path=0, ls_init_value=1e-5),
                lstm=dict(dws_conv=False, dws_conv_only_hidden=True, dws_conv_kernel_size=3, drop_cell_update=0))),
        fpn=dict(name='PAFPN', compile=_compile_off(), depth=s['fpn_depth'], in_stages=[2, 3, 4], depthwise=False,
                 act='silu'),
        head=dict(name='YoloX', compile=_compile_off(), depthwise=False, act='silu', obj_focal_loss=False,
                  bbox_loss_weighting='', ignore_bbox_thresh=None, ignore_label=1024, ignore_bg_k=0),
        postprocess=dict(confidence_threshold=0.1, nms_threshold=0.45),
        use_label_every=1, ignore_image=False)
    # the shipped model groups (config/model/*.yaml): `model=rnndet-soft` (self-training rounds: pseudo boxes below the
    # per-class obj/cls confidence are ignored, not suppressed), its 1 Mpx WSOD variant, and the two pseudo-labellers
    if name in ('rnndet-soft', 'rnndet-soft-gen4-wsod'):
        cfg['name'] = 'rnndet'
        cfg['head']['ignore_bbox_thresh'] = [0.7, 0.35] if name == 'rnndet-soft' else [0.7, 0.55]
    if name in ('pseudo_labeler', 'pseudo_labeler-gen4-wsod'):
        cfg['name'] = 'pseudo_labeler'
        thr = [0.6, 0.3] if name == 'pseudo_labeler' else [0.6, 0.5]
        cfg['pseudo_label'] = dict(skip_first_t=0, obj_thresh=list(thr), cls_thresh=list(thr), min_track_len=6,
                                   track_method='forward or backward', inpaint=True, ignore_label=1024)
    for k, v in over.items():
        cfg[k] = v
    return cfg


_DATASETS = {
    'gen1': dict(name='gen1', path='./datasets/gen1/', ev_repr_name='stacked_histogram_dt=50_nbins=10',
                 sequence_length=21, resolution_hw=[240, 304], downsample_by_factor_2=False,
                 only_load_end_labels=False, tflip_offset=-1),
    'gen4': dict(name='gen4', path='./datasets/gen4/', ev_repr_name='stacked_histogram_dt=50_nbins=10',
                 sequence_length=5, resolution_hw=[720, 1280], downsample_by_factor_2=True,
                 only_load_end_labels=False, tflip_offset=-2),
}


def dataset_config(name: str) -> dict:
    d = dict(_DATASETS[name])
    tfo = d.pop('tflip_offset')
    rot = dict(prob=0, min_angle_deg=2, max_angle_deg=6)
    d.update(ssod=False, ratio=-1, train_ratio=-1, val_ratio=-1, test_ratio=-1, only_load_labels=False, reverse_event_order=False,
             train=dict(sampling='mixed', random=dict(weighted_sampling=False), mixed=dict(w_stream=1, w_random=1)),
             eval=dict(sampling='stream'),
             # config/dataset/base.yaml:19-58
             data_augmentation=dict(
                 tflip_offset=tfo,
                 random=dict(prob_hflip=0.5, prob_tflip=0, rotate=dict(rot),
                             zoom=dict(prob=0.8, zoom_in=dict(weight=8, factor=dict(min=1, max=1.5)),
                                       zoom_out=dict(weight=2, factor=dict(min=1, max=1.2)))),
                 stream=dict(start_from_zero=False, prob_hflip=0.5, prob_tflip=0, rotate=dict(rot),
                             zoom=dict(prob=0.5, zoom_out=dict(factor=dict(min=1, max=1.2))))))
    return d


def _deep_update(dst: dict, src: Mapping):
    for k, v in src.items():
        if isinstance(v, Mapping) and isinstance(dst.get(k), dict):
            _deep_update(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)


def full_config(dataset: str = 'gen1', size: str = 'small', model: str = 'rnndet', is_train: bool = True,
                overrides: Optional[Mapping] = None):
    """Equivalent of ``train.py model=<model> dataset=<dataset> +experiment/<dataset>=<size>.yaml``."""
    gen4 = dataset == 'gen4'
    cfg = dict(
        is_train=is_train,
        reproduce=dict(seed_everything=None, deterministic_flag=False, benchmark=True),
        training=dict(precision=16, max_epochs=10000, max_steps=400000, learning_rate=0.000346 if gen4 else 0.0002,
                      weight_decay=0, gradient_clip_val=1.0, limit_train_batches=1.0,
                      lr_scheduler=dict(use=True, total_steps=400000, pct_start=0.005, div_factor=20,
                                        final_div_factor=10000)),
        validation=dict(limit_val_batches=1.0, val_check_interval=20000, check_val_every_n_epoch=None),
        batch_size=dict(train=12 if gen4 else 8, eval=12 if gen4 else 8),
        hardware=dict(num_workers=dict(train=8, eval=4 if gen4 else 8), gpus=0, dist_backend='nccl'),
        logging=dict(ckpt_every_min=18,
                     train=dict(metrics=dict(compute=False, detection_metrics_every_n_steps=None),
                                log_model_every_n_steps=5000, log_every_n_steps=100,
                                high_dim=dict(enable=True, every_n_steps=5000, n_samples=4)),
                     validation=dict(high_dim=dict(enable=True, every_n_epochs=1, n_samples=8))),
        suffix='', weight='', checkpoint='',
        tta=dict(enable=False, hflip=True, tflip=True), use_gt=True, save_dir='',
        dataset=dataset_config(dataset),
        model=model_config(size, model, partition_split_32=2 if gen4 else 1))
    if overrides:
        _deep_update(cfg, overrides)
    return create(cfg)
