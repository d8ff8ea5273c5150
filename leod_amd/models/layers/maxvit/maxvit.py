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


class GLU(nn.Module):
    """Gated linear unit parameters (maxvit.py:56-82): ``proj`` maps dim_in -> 2 * dim_out, forward = first half * act(second half)."""

    def __init__(self, dim_in: int, dim_out: int, channel_last: bool, act_layer, bias: bool = True):
        super().__init__()
        assert channel_last
        self.proj = nn.Linear(dim_in, dim_out * 2, bias=bias)
        self.act_layer = act_layer if isinstance(act_layer, nn.Module) else nn.Identity()


class MLP(nn.Module):
    """Parameters of the block's MLP (maxvit.py:85-118): Linear -> act -> Linear, or GLU -> Linear when ``gated`` (inner width
    floor(dim * ratio * 2 / 3 / 32) * 32 then).  ``act_layer``: an activation NAME of ``ops.ACTIVATIONS`` (or nn.GELU).  The shipped
    combination (non-gated, GELU, bias) runs inside the fused block kernels; anything else through ``functions``' composed nodes."""

    def __init__(self, dim: int, channel_last: bool, expansion_ratio: int, act_layer=nn.GELU, gated: bool = False,
                 bias: bool = True, drop_prob: float = 0.):
        super().__init__()
        act = 'gelu' if act_layer is nn.GELU else act_layer
        if not channel_last or drop_prob > 0 or not isinstance(act, str) or act not in ops.ACTIVATIONS:
            raise NotImplementedError(f'HIP MLP: channels-last, no dropout, activation one of {sorted(ops.ACTIVATIONS)} (got {act_layer!r})')
        self.act_name, self.gated, self.has_bias = act, bool(gated), bool(bias)
        inner = int(dim * expansion_ratio)
        if gated:
            import math
            inner = math.floor(inner * 2 / 3 / 32) * 32
            proj_in = GLU(dim_in=dim, dim_out=inner, channel_last=True, act_layer=None, bias=bias)
        else:
            proj_in = nn.Sequential(nn.Linear(dim, inner, bias=bias), nn.GELU() if act == 'gelu' else nn.Identity())
        self.inner = inner
        self.net = nn.Sequential(proj_in, nn.Dropout(p=0.), nn.Linear(inner, dim, bias=bias))

    @property
    def fc1(self) -> nn.Linear:
        return self.net[0].proj if self.gated else self.net[0][0]


class SelfAttentionCl(nn.Module):
    """Channels-last MHSA parameters (qkv/proj Linear); standalone ``forward`` treats every leading
    [B, h, w] slab as one partition (maxvit.py:328-354)."""

    def __init__(self, dim: int, dim_head: int = 32, bias: bool = True):
        super().__init__()
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


class TorchMHSAWrapperCl(nn.Module):
    """Parameters in ``nn.MultiheadAttention``'s layout (maxvit.py:307-325: keys ``mha.in_proj_weight`` [3C, C] = (q | k | v) rows,
    ``mha.out_proj``).  The same attention: the block re-orders the in-projection rows per head into the kernels' (q_h | k_h | v_h)
    interleaving (``interleave``), gradients flow back through that gather."""

    def __init__(self, dim: int, dim_head: int = 32, bias: bool = True):
        super().__init__()
        assert dim % dim_head == 0
        self.num_heads = dim // dim_head
        self.dim_head = dim_head
        self.mha = nn.MultiheadAttention(embed_dim=dim, num_heads=self.num_heads, bias=bias, batch_first=True)
        d, C = dim_head, dim
        idx = [part * C + h * d + j for h in range(self.num_heads) for part in range(3) for j in range(d)]
        self.register_buffer('interleave', torch.tensor(idx, dtype=torch.long), persistent=False)


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
        if attention_cfg.get('drop_path', 0.0) > 0 or attention_cfg.get('drop_mlp', 0.0) > 0 or norm_eps != 1e-5:
            raise NotImplementedError('HIP PartitionAttentionCl: no drop path / MLP dropout (0 in every config), LayerNorm eps 1e-5')
        use_mha, gated, act = bool(attention_cfg.use_torch_mha), bool(attention_cfg.mlp_gated), attention_cfg.mlp_activation
        attn_bias, mlp_bias = attention_cfg.get('attention_bias', True), attention_cfg.get('mlp_bias', True)
        # the shipped combination runs as ONE fused autograd node (functions.AttnBlockFn); every other one through the composed nodes
        self.generic = use_mha or gated or act != 'gelu' or not ls_init_value > 0 or not attn_bias or not mlp_bias
        self.partition_size = (partition_size, partition_size) if isinstance(partition_size, int) else tuple(partition_size)
        assert len(self.partition_size) == 2
        assert isinstance(partition_type, PartitionType)
        self.partition_window = partition_type == PartitionType.WINDOW
        self.norm1 = nn.Identity() if skip_first_norm else LayerNorm(dim, eps=norm_eps)
        self.self_attn = (TorchMHSAWrapperCl if use_mha else SelfAttentionCl)(dim, dim_head=dim_head, bias=attn_bias)
        self.ls1 = LayerScale(dim=dim, init_values=ls_init_value) if ls_init_value > 0 else nn.Identity()
        self.drop_path1 = nn.Identity()
        self.norm2 = LayerNorm(dim, eps=norm_eps)
        self.mlp = MLP(dim=dim, channel_last=True, expansion_ratio=attention_cfg.get('mlp_ratio', 4), act_layer=act,
                       gated=gated, bias=mlp_bias, drop_prob=0.)
        self.ls2 = LayerScale(dim=dim, init_values=ls_init_value) if ls_init_value > 0 else nn.Identity()
        self.drop_path2 = nn.Identity()
        self.dim = dim

    def _const(self, name: str, n: int, value: float, like: torch.Tensor) -> torch.Tensor:
        """constant vectors standing in for parameters an option removes (bias=False, no LayerScale): kept per module and device"""
        t = self.__dict__.get('_c_' + name)
        if t is None or t.device != like.device or t.numel() != n:
            t = torch.full((n,), value, dtype=torch.float32, device=like.device)
            self.__dict__['_c_' + name] = t
        return t

    def _forward_generic(self, x):
        """The block out of composed autograd nodes (functions.LNLinearFn / AttnCoreFn / LinearScaleResFn / ActGluFn) on fp32 rows."""
        B, H, W, C = x.shape
        x2 = x.contiguous().view(-1, C)
        n1 = self.norm1 if isinstance(self.norm1, nn.LayerNorm) else None
        sa, mlp = self.self_attn, self.mlp
        if isinstance(sa, TorchMHSAWrapperCl):
            wq = sa.mha.in_proj_weight.index_select(0, sa.interleave)
            bq = sa.mha.in_proj_bias.index_select(0, sa.interleave) if sa.mha.in_proj_bias is not None else self._const('b3', 3 * C, 0., x)
            wp, bp = sa.mha.out_proj.weight, sa.mha.out_proj.bias
        else:
            wq, bq, wp, bp = sa.qkv.weight, sa.qkv.bias, sa.proj.weight, sa.proj.bias
        bq = bq if bq is not None else self._const('b3', 3 * C, 0., x)
        bp = bp if bp is not None else self._const('b1', C, 0., x)
        g1 = self.ls1.gamma if isinstance(self.ls1, LayerScale) else self._const('one', C, 1., x)
        g2 = self.ls2.gamma if isinstance(self.ls2, LayerScale) else self._const('one', C, 1., x)
        qkv = Fn.LNLinearFn.apply(x2, n1.weight if n1 is not None else None, n1.bias if n1 is not None else None, wq, bq)
        o = Fn.AttnCoreFn.apply(qkv.view(B, H, W, 3 * C), sa.num_heads, self.partition_size, self.partition_window)
        y = Fn.LinearScaleResFn.apply(o.reshape(-1, C), wp, bp, g1, x2)
        fc1, fc2 = mlp.fc1, mlp.net[2]
        b1 = fc1.bias if fc1.bias is not None else self._const('bh', fc1.weight.shape[0], 0., x)
        b2 = fc2.bias if fc2.bias is not None else self._const('b1', C, 0., x)
        p = Fn.LNLinearFn.apply(y, self.norm2.weight, self.norm2.bias, fc1.weight, b1)
        h = Fn.ActGluFn.apply(p, mlp.act_name, mlp.gated)
        z = Fn.LinearScaleResFn.apply(h, fc2.weight, b2, g2, y)
        return z.view(B, H, W, C)

    def forward(self, x):
        """x: [B,H,W,C] channels-last."""
        if self.generic:
            return self._forward_generic(x)
        n1 = self.norm1 if isinstance(self.norm1, nn.LayerNorm) else None
        sa, mlp = self.self_attn, self.mlp
        return Fn.AttnBlockFn.apply(
            self, x.contiguous(),
            n1.weight if n1 is not None else None, n1.bias if n1 is not None else None,
            sa.qkv.weight, sa.qkv.bias, sa.proj.weight, sa.proj.bias, self.ls1.gamma,
            self.norm2.weight, self.norm2.bias, mlp.net[0][0].weight, mlp.net[0][0].bias,
            mlp.net[2].weight, mlp.net[2].bias, self.ls2.gamma)
