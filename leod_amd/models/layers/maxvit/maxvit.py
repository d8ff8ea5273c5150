"""MaxViT pieces of the RVT backbone, HIP-backed (mirror of the reference's
models/layers/maxvit/maxvit.py: same class names, constructor arguments and state-dict keys).

Differences in *how* things run, not in what they compute:
  * tensors stay channels-last fp32 rows; window/grid partitioning is index math inside the attention
    kernel (leod_partition_attn_*), so ``window_partition`` & co. below are plain view helpers kept for
    API compatibility and tests;
  * ``PartitionAttentionCl.forward`` is one autograd node (functions.AttnBlockFn) = 5 forward kernels;
  * the attention block implements the configuration the reference ships (SelfAttentionCl, non-gated GELU MLP,
    LayerScale > 0, drop_path = drop_mlp = 0) and raises NotImplementedError loudly for anything else; the downsample layer
    takes both of its options (overlap, norm_affine).
"""
from enum import Enum, auto
from typing import Tuple

import torch
from torch import nn

from leod_amd import functions as Fn
from leod_amd import ops


class PartitionType(Enum):
    WINDOW = auto()
    GRID = auto()


def nChw_2_nhwC(x: torch.Tensor):
    assert x.ndim == 4
    return x.permute(0, 2, 3, 1)


def nhwC_2_nChw(x: torch.Tensor):
    assert x.ndim == 4
    return x.permute(0, 3, 1, 2)


# ---- partition helpers (maxvit.py:273-304 of the reference); pure views/copies, no compute ----------
def window_partition(x, window_size: Tuple[int, int]):
    B, H, W, C = x.shape
    assert H % window_size[0] == 0 and W % window_size[1] == 0
    x = x.reshape(B, H // window_size[0], window_size[0], W // window_size[1], window_size[1], C)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, window_size[0], window_size[1], C)


def window_reverse(windows, window_size: Tuple[int, int], img_size: Tuple[int, int]):
    H, W = img_size
    C = windows.shape[-1]
    x = windows.reshape(-1, H // window_size[0], W // window_size[1], window_size[0], window_size[1], C)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, H, W, C)


def grid_partition(x, grid_size: Tuple[int, int]):
    B, H, W, C = x.shape
    assert H % grid_size[0] == 0 and W % grid_size[1] == 0
    x = x.reshape(B, grid_size[0], H // grid_size[0], grid_size[1], W // grid_size[1], C)
    return x.permute(0, 2, 4, 1, 3, 5).reshape(-1, grid_size[0], grid_size[1], C)


def grid_reverse(windows, grid_size: Tuple[int, int], img_size: Tuple[int, int]):
    H, W = img_size
    C = windows.shape[-1]
    x = windows.reshape(-1, H // grid_size[0], W // grid_size[1], grid_size[0], grid_size[1], C)
    return x.permute(0, 3, 1, 4, 2, 5).reshape(-1, H, W, C)


class LayerNorm(nn.LayerNorm):
    """timm-style LayerNorm over the last dim (``num_channels`` ctor argument, parameters weight/bias)."""

    def __init__(self, num_channels, eps=1e-5, affine=True):
        super().__init__(num_channels, eps=eps, elementwise_affine=affine)

    def forward(self, x):
        y, _ = ops.layernorm_fwd(x.contiguous(), self.weight, self.bias, eps=self.eps)
        return y


class LayerScale(nn.Module):
    def __init__(self, dim: int, init_values: float = 1e-5, inplace: bool = False):
        super().__init__()
        self.inplace = inplace
        self.gamma = nn.Parameter(init_values * torch.ones(dim))


class MLP(nn.Module):
    """Linear -> GELU(erf) -> Linear (reference MLP with gated=False, maxvit.py:85-118)."""

    def __init__(self, dim: int, channel_last: bool, expansion_ratio: int, act_layer=nn.GELU, gated: bool = False,
                 bias: bool = True, drop_prob: float = 0.):
        super().__init__()
        if gated or not channel_last or not bias or drop_prob > 0 or act_layer is not nn.GELU:
            raise NotImplementedError('HIP MLP implements the shipped config: channels-last, non-gated, GELU, bias, no dropout')
        inner = int(dim * expansion_ratio)
        self.net = nn.Sequential(nn.Sequential(nn.Linear(dim, inner, bias=True), nn.GELU()), nn.Dropout(p=0.),
                                 nn.Linear(inner, dim, bias=True))


class SelfAttentionCl(nn.Module):
    """Channels-last MHSA parameters (qkv/proj Linear); standalone ``forward`` treats every leading
    [B, h, w] slab as one partition (maxvit.py:328-354)."""

    def __init__(self, dim: int, dim_head: int = 32, bias: bool = True):
        super().__init__()
        if not bias:
            raise NotImplementedError('attention_bias=False is not implemented')
        self.num_heads = dim // dim_head
        self.dim_head = dim_head
        self.scale = dim_head ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=bias)
        self.proj = nn.Linear(dim, dim, bias=bias)

    @torch.no_grad()
    def forward(self, x: torch.Tensor):
        """Inference-only convenience: x [Bp, ph, pw, C], each leading slab is one attention window."""
        x = x.contiguous()
        Bp, ph, pw, C = x.shape
        qkv, _, _ = ops.ln_linear_fwd(x, None, None, self.qkv.weight, self.qkv.bias)
        o, _ = ops.partition_attn_fwd(qkv, self.num_heads, (ph, pw), True)
        out, _, _ = ops.ln_linear_fwd(o, None, None, self.proj.weight, self.proj.bias)
        return out


class DownsampleBase(nn.Module):
    @staticmethod
    def output_is_normed():
        raise NotImplementedError


class ConvDownsampling_Cf2Cl(DownsampleBase):
    """Conv (no bias) + LayerNorm; NCHW (or raw event tensor) in, NHWC out (maxvit.py:143-182).  ``overlap`` (shipped): k = 2s - 1, pad k // 2;
    otherwise non-overlapping patches k = s, pad 0.  ``norm_affine=False``: LayerNorm without weight / bias (no such state-dict keys)."""

    def __init__(self, dim_in: int, dim_out: int, downsample_factor: int, downsample_cfg):
        super().__init__()
        assert downsample_factor in (2, 4, 8)
        overlap, affine = downsample_cfg.get('overlap', True), downsample_cfg.get('norm_affine', True)
        k = (downsample_factor - 1) * 2 + 1 if overlap else downsample_factor
        self.conv = nn.Conv2d(dim_in, dim_out, kernel_size=k, padding=k // 2 if overlap else 0, stride=downsample_factor, bias=False)
        self.norm = LayerNorm(num_channels=dim_out, eps=1e-5, affine=affine)
        if not affine:                       # the kernels take a scale / shift: constants, and their (discarded) gradients, outside the state dict
            self.register_buffer('_ln_one', torch.ones(dim_out), persistent=False)
            self.register_buffer('_ln_zero', torch.zeros(dim_out), persistent=False)
        self.stride = downsample_factor
        self.is_stem = downsample_factor == 4

    def forward(self, x: torch.Tensor, padded_hw=None):
        """x: [B,C,H,W].  Stem (factor 4): raw uint8/fp32 NCHW voxels, optionally still unpadded
        (``padded_hw`` = model input resolution).  Other stages: any-stride NCHW view of an NHWC map."""
        if self.is_stem:
            if x.dtype not in (torch.uint8, torch.float32):
                x = x.float()
            x = x.contiguous()
            padded_hw = tuple(padded_hw) if padded_hw is not None else tuple(x.shape[-2:])
        else:
            x = Fn.to_nhwc(x)
        ln_w, ln_b = (self.norm.weight, self.norm.bias) if self.norm.weight is not None else (self._ln_one, self._ln_zero)
        return Fn.ConvLNFn.apply(self, x, self.conv.weight, ln_w, ln_b, self.is_stem, self.stride, padded_hw)

    @staticmethod
    def output_is_normed():
        return True


def get_downsample_layer_Cf2Cl(dim_in: int, dim_out: int, downsample_factor: int, downsample_cfg) -> DownsampleBase:
    if downsample_cfg.type == 'patch':
        return ConvDownsampling_Cf2Cl(dim_in=dim_in, dim_out=dim_out, downsample_factor=downsample_factor,
                                      downsample_cfg=downsample_cfg)
    raise NotImplementedError


class PartitionAttentionCl(nn.Module):
    """Window or grid partition attention + MLP block on channels-last maps (maxvit.py:185-270)."""

    def __init__(self, dim: int, partition_type: PartitionType, attention_cfg, skip_first_norm: bool = False):
        super().__init__()
        norm_eps = attention_cfg.get('norm_eps', 1e-5)
        partition_size = attention_cfg.partition_size
        dim_head = attention_cfg.get('dim_head', 32)
        ls_init_value = attention_cfg.get('ls_init_value', 1e-5)
        if attention_cfg.use_torch_mha or attention_cfg.mlp_gated or attention_cfg.mlp_activation != 'gelu' \
                or attention_cfg.get('drop_path', 0.0) > 0 or attention_cfg.get('drop_mlp', 0.0) > 0 \
                or not ls_init_value > 0 or norm_eps != 1e-5:
            raise NotImplementedError('HIP PartitionAttentionCl implements the shipped config (SelfAttentionCl, '
                                      'GELU non-gated MLP, LayerScale>0, no drop path/mlp, eps 1e-5)')
        self.partition_size = (partition_size, partition_size) if isinstance(partition_size, int) else tuple(partition_size)
        assert len(self.partition_size) == 2
        assert isinstance(partition_type, PartitionType)
        self.partition_window = partition_type == PartitionType.WINDOW
        self.norm1 = nn.Identity() if skip_first_norm else LayerNorm(dim, eps=norm_eps)
        self.self_attn = SelfAttentionCl(dim, dim_head=dim_head, bias=attention_cfg.get('attention_bias', True))
        self.ls1 = LayerScale(dim=dim, init_values=ls_init_value)
        self.drop_path1 = nn.Identity()
        self.norm2 = LayerNorm(dim, eps=norm_eps)
        self.mlp = MLP(dim=dim, channel_last=True, expansion_ratio=attention_cfg.get('mlp_ratio', 4), act_layer=nn.GELU,
                       gated=False, bias=attention_cfg.get('mlp_bias', True), drop_prob=0.)
        self.ls2 = LayerScale(dim=dim, init_values=ls_init_value)
        self.drop_path2 = nn.Identity()

    def forward(self, x):
        """x: [B,H,W,C] channels-last."""
        n1 = self.norm1 if isinstance(self.norm1, nn.LayerNorm) else None
        sa, mlp = self.self_attn, self.mlp
        return Fn.AttnBlockFn.apply(
            self, x.contiguous(),
            n1.weight if n1 is not None else None, n1.bias if n1 is not None else None,
            sa.qkv.weight, sa.qkv.bias, sa.proj.weight, sa.proj.bias, self.ls1.gamma,
            self.norm2.weight, self.norm2.bias, mlp.net[0][0].weight, mlp.net[0][0].bias,
            mlp.net[2].weight, mlp.net[2].bias, self.ls2.gamma)
