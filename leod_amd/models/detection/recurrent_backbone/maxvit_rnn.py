"""4-stage recurrent MaxViT backbone (mirror of the reference's
models/detection/recurrent_backbone/maxvit_rnn.py:23-201; same module tree => same state-dict keys
``stages.{i}.{downsample_cf2cl,att_blocks.{j}.att_{window,grid},lstm}``).

Per stage and timestep the HIP path launches 13 kernels: conv, LayerNorm, 2 x (LN+qkv, attention,
proj+LayerScale+residual, LN+fc1+GELU, fc2+LayerScale+residual) and the fused ConvLSTM cell.
Feature maps are returned as logical [B,C,h,w] tensors backed by channels-last memory."""
from typing import Dict, Optional, Tuple

import torch as th
import torch.nn as nn

from ...layers.rnn import DWSConvLSTM2d
from ...layers.maxvit.maxvit import PartitionAttentionCl, nhwC_2_nChw, get_downsample_layer_Cf2Cl, PartitionType
from .base import BaseDetector


class MaxVitAttentionPairCl(nn.Module):
    def __init__(self, dim: int, skip_first_norm: bool, attention_cfg):
        super().__init__()
        self.att_window = PartitionAttentionCl(dim=dim, partition_type=PartitionType.WINDOW,
                                               attention_cfg=attention_cfg, skip_first_norm=skip_first_norm)
        self.att_grid = PartitionAttentionCl(dim=dim, partition_type=PartitionType.GRID,
                                             attention_cfg=attention_cfg, skip_first_norm=False)

    def forward(self, x):
        return self.att_grid(self.att_window(x))


class RNNDetectorStage(nn.Module):
    def __init__(self, dim_in: int, stage_dim: int, spatial_downsample_factor: int, num_blocks: int,
                 enable_token_masking: bool, T_max_chrono_init: Optional[int], stage_cfg):
        super().__init__()
        assert isinstance(num_blocks, int) and num_blocks > 0
        lstm_cfg = stage_cfg.lstm
        self.downsample_cf2cl = get_downsample_layer_Cf2Cl(dim_in=dim_in, dim_out=stage_dim,
                                                           downsample_factor=spatial_downsample_factor,
                                                           downsample_cfg=stage_cfg.downsample)
        self.att_blocks = nn.ModuleList([
            MaxVitAttentionPairCl(dim=stage_dim, skip_first_norm=(i == 0 and self.downsample_cf2cl.output_is_normed()),
                                  attention_cfg=stage_cfg.attention) for i in range(num_blocks)])
        self.lstm = DWSConvLSTM2d(dim=stage_dim, dws_conv=lstm_cfg.dws_conv,
                                  dws_conv_only_hidden=lstm_cfg.dws_conv_only_hidden,
                                  dws_conv_kernel_size=lstm_cfg.dws_conv_kernel_size,
                                  cell_update_dropout=lstm_cfg.get('drop_cell_update', 0))
        # learnable [MASK] token put to masked pixels of the stage's input map (maxvit_rnn.py:174-180; off in every shipped config)
        self.mask_token = nn.Parameter(th.zeros(1, 1, 1, stage_dim), requires_grad=True) if enable_token_masking else None
        if self.mask_token is not None:
            th.nn.init.normal_(self.mask_token, std=.02)

    def _mask_tokens(self, x: th.Tensor, token_mask: Optional[th.Tensor]) -> th.Tensor:
        """x[token_mask] = mask_token (maxvit_rnn.py:190-192); x [N,H,W,C], token_mask [N,H,W] bool"""
        if token_mask is None:
            return x
        assert self.mask_token is not None, 'No mask token present in this stage'
        from leod_amd import functions as Fn
        return Fn.TokenMaskFn.apply(x.contiguous(), token_mask.to(th.bool).contiguous(), self.mask_token)

    def forward(self, x: th.Tensor, h_and_c_previous=None, token_mask: Optional[th.Tensor] = None, padded_hw=None):
        x = self.downsample_cf2cl(x, padded_hw=padded_hw)          # -> N H W C
        x = self._mask_tokens(x, token_mask)
        for blk in self.att_blocks:
            x = blk(x)
        h_c = self.lstm(nhwC_2_nChw(x), h_and_c_previous)          # zero-copy view, no .contiguous()
        return h_c[0], h_c

    def forward_sequence(self, x: th.Tensor, T: int, h_and_c_previous=None, padded_hw=None, token_mask: Optional[th.Tensor] = None):
        """All T timesteps of a sequence batch at once: x [T*B,...].  Downsampling and the attention blocks are per-frame
        maps, so they run ONCE on the T*B batch (21x larger launches instead of 21x more of them); only the ConvLSTM
        recurrence walks over t (``DWSConvLSTM2d.forward_sequence``).  Same values as T chained ``forward`` calls."""
        x = self.downsample_cf2cl(x, padded_hw=padded_hw)
        x = self._mask_tokens(x, token_mask)
        for blk in self.att_blocks:
            x = blk(x)
        return self.lstm.forward_sequence(nhwC_2_nChw(x), T, h_and_c_previous)


def _ensure_shadows():
    """16-bit weight shadows of flat parameter buffers (leod_amd.parallel.FlatParams) follow in-place edits of the weights on EVERY forward
    entry point, not only on the ones that go through Module._run_sequence"""
    from leod_amd.parallel import FlatParams
    FlatParams.ensure_all_shadows()


class RNNDetector(BaseDetector):
    def __init__(self, mdl_config):
        super().__init__()
        in_channels = mdl_config.input_channels
        embed_dim = mdl_config.embed_dim
        dim_multiplier = tuple(mdl_config.dim_multiplier)
        num_blocks = tuple(mdl_config.num_blocks)
        t_max = tuple(mdl_config.T_max_chrono_init)          # parsed, never used by the reference either
        assert len(num_blocks) == 4 == len(dim_multiplier) == len(t_max)
        assert isinstance(embed_dim, int)
        self.in_res_hw = tuple(mdl_config.in_res_hw) if mdl_config.get('in_res_hw', None) is not None else None
        patch_size = mdl_config.stem.patch_size
        self.stage_dims = [embed_dim * m for m in dim_multiplier]
        self.stages = nn.ModuleList()
        self.strides = []
        input_dim, stride = in_channels, 1
        for i, (nb, tm) in enumerate(zip(num_blocks, t_max)):
            factor = patch_size if i == 0 else 2
            self.stages.append(RNNDetectorStage(dim_in=input_dim, stage_dim=self.stage_dims[i],
                                                spatial_downsample_factor=factor, num_blocks=nb,
                                                enable_token_masking=mdl_config.enable_masking and i == 0,
                                                T_max_chrono_init=tm, stage_cfg=mdl_config.stage))
            stride *= factor
            self.strides.append(stride)
            input_dim = self.stage_dims[i]
        self.num_stages = 4

    def get_stage_dims(self, stages: Tuple[int, ...]) -> Tuple[int, ...]:
        idx = [s - 1 for s in stages]
        assert min(idx) >= 0 and max(idx) < len(self.stages), idx
        return tuple(self.stage_dims[i] for i in idx)

    def get_strides(self, stages: Tuple[int, ...]) -> Tuple[int, ...]:
        idx = [s - 1 for s in stages]
        assert min(idx) >= 0 and max(idx) < len(self.stages), idx
        return tuple(self.strides[i] for i in idx)

    def forward(self, x: th.Tensor, prev_states=None, token_mask: Optional[th.Tensor] = None):
        """x: [B,C,H,W] event voxels -- fp32 already padded to ``in_res_hw`` (reference convention) or the raw
        uint8/fp32 unpadded tensor (the zero padding is then folded into the stem kernel)."""
        if prev_states is None:
            prev_states = [None] * self.num_stages
        assert len(prev_states) == self.num_stages
        _ensure_shadows()
        padded_hw = self.in_res_hw if (self.in_res_hw is not None and tuple(x.shape[-2:]) != self.in_res_hw) else None
        states, output = [], {}
        for i, stage in enumerate(self.stages):
            x, state = stage(x, prev_states[i], token_mask if i == 0 else None, padded_hw if i == 0 else None)
            states.append(state)
            output[i + 1] = x
        return output, states

    def forward_sequence(self, x_seq: th.Tensor, prev_states=None, select_rows: Optional[th.Tensor] = None, select_stages=(), inject=None,
                         token_mask: Optional[th.Tensor] = None):
        """x_seq [T,B,C,H,W]: stage-major, time-batched evaluation of a whole sequence (see
        ``RNNDetectorStage.forward_sequence``).  Returns {stage: features of all timesteps [T*B,C,h,w]} and the final
        states -- the per-timestep loop of modules/detection.py:188-226 with the loops interchanged.
        ``select_rows`` (frame indices t*B + b of the labelled frames) + ``select_stages``: additionally returns {stage: features of
        those frames} (what ``BackboneFeatureSelector`` gathers, modules/utils/detection.py:27-58), selected inside the stage loop so
        that the two consumers of a stage output share one autograd node (``functions.ForkSelectFn``)."""
        T, B = x_seq.shape[:2]
        _ensure_shadows()
        if prev_states is None:
            prev_states = [None] * self.num_stages
        padded_hw = self.in_res_hw if (self.in_res_hw is not None and tuple(x_seq.shape[-2:]) != self.in_res_hw) else None
        x = x_seq.reshape((T * B,) + tuple(x_seq.shape[2:]))
        states, output = [], {}
        from leod_amd.functions import bucket_boundary, fork_select
        selected = {}
        for i, stage in enumerate(self.stages):
            if i > 0:
                # data-parallel training: when the backward pass arrives here, stage i is done -> its gradient bucket is exchanged
                x = bucket_boundary(i, x)
            x, state = stage.forward_sequence(x, T, prev_states[i], padded_hw if i == 0 else None,
                                              token_mask.reshape((T * B,) + tuple(token_mask.shape[2:])) if (token_mask is not None and i == 0) else None)
            states.append(state)
            if inject is not None and (i + 1) in select_stages:
                # a step recorded as separate backbone / head launch plans (modules/step_plan.py): the stage output is exposed to the head plan
                # and the labelled frames' gradient comes back through static buffers of ``inject``
                x = inject.fork(i + 1, x)
            elif select_rows is not None and (i + 1) in select_stages:
                x, selected[i + 1] = fork_select(x, select_rows)
            output[i + 1] = x
        return (output, states) if select_rows is None else (output, states, selected)
