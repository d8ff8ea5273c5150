"""Oracle: RVT recurrent backbone (MaxViT window/grid attention + ConvLSTM), functional PyTorch-CPU
fp32.  TEST INFRASTRUCTURE (see oracle/__init__.py).

All functions take the model ``state_dict`` ``sd`` plus a key prefix, so the same tensors that are
loaded into the reference (when golden vectors are made) or into the HIP-backed modules are used
here unchanged.  ``path:line`` citations are relative to /root/reference.
"""
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------
# partitions (models/layers/maxvit/maxvit.py:273-304)
# ------------------------------------------------------------------------------------------------
def window_partition(x, ws):
    """maxvit.py:273-279: contiguous ws[0] x ws[1] tiles -> [B*nH*nW, ws0, ws1, C]."""
    B, H, W, C = x.shape
    x = x.reshape(B, H // ws[0], ws[0], W // ws[1], ws[1], C)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, ws[0], ws[1], C)


def window_reverse(win, ws, hw):
    """maxvit.py:282-287."""
    H, W = hw
    C = win.shape[-1]
    x = win.reshape(-1, H // ws[0], W // ws[1], ws[0], ws[1], C)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, H, W, C)


def grid_partition(x, gs):
    """maxvit.py:290-296: dilated grids; token (gy,gx) of partition (py,px) is pixel
    (gy*(H/gs0)+py, gx*(W/gs1)+px)."""
    B, H, W, C = x.shape
    x = x.reshape(B, gs[0], H // gs[0], gs[1], W // gs[1], C)
    return x.permute(0, 2, 4, 1, 3, 5).reshape(-1, gs[0], gs[1], C)


def grid_reverse(win, gs, hw):
    """maxvit.py:299-304."""
    H, W = hw
    C = win.shape[-1]
    x = win.reshape(-1, H // gs[0], W // gs[1], gs[0], gs[1], C)
    return x.permute(0, 3, 1, 4, 2, 5).reshape(-1, H, W, C)


# ------------------------------------------------------------------------------------------------
# layers
# ------------------------------------------------------------------------------------------------
def layer_norm(x, sd, prefix, eps=1e-5):
    """timm LayerNorm over the last dim (maxvit.py:172,201,229; layers/norm.py:44-56)."""
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + '.weight'], sd[prefix + '.bias'], eps)


def self_attention(x, sd, prefix, dim_head):
    """SelfAttentionCl.forward, maxvit.py:343-354.  x: [Bp, ph, pw, C].
    qkv columns are interleaved per head: head h owns [h*3d, (h+1)*3d) = (q | k | v)."""
    Bp = x.shape[0]
    restore = x.shape[:-1]
    C = x.shape[-1]
    heads = C // dim_head
    if (prefix + '.mha.in_proj_weight') in sd:
        # TorchMHSAWrapperCl (maxvit.py:307-325): nn.MultiheadAttention(batch_first) = the same attention with the in-projection rows
        # ordered (all q | all k | all v), head h owning channels [h d, (h + 1) d) of each
        qkv = F.linear(x.reshape(Bp, -1, C), sd[prefix + '.mha.in_proj_weight'], sd.get(prefix + '.mha.in_proj_bias'))
        q, k, v = (t.view(Bp, -1, heads, dim_head).transpose(1, 2) for t in qkv.chunk(3, dim=2))
        attn = ((q * (dim_head ** -0.5)) @ k.transpose(-2, -1)).softmax(dim=-1)
        o = (attn @ v).transpose(1, 2).reshape(restore + (-1,))
        return F.linear(o, sd[prefix + '.mha.out_proj.weight'], sd.get(prefix + '.mha.out_proj.bias'))
    qkv = F.linear(x, sd[prefix + '.qkv.weight'], sd.get(prefix + '.qkv.bias'))
    q, k, v = qkv.view(Bp, -1, heads, dim_head * 3).transpose(1, 2).chunk(3, dim=3)
    attn = (q @ k.transpose(-2, -1)) * (dim_head ** -0.5)
    attn = attn.softmax(dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(restore + (-1,))
    return F.linear(o, sd[prefix + '.proj.weight'], sd.get(prefix + '.proj.bias'))


ACTIVATIONS = {'gelu': F.gelu, 'silu': F.silu, 'swish': F.silu, 'relu': F.relu, 'sigmoid': torch.sigmoid, 'tanh': torch.tanh, 'relu6': F.relu6,
               'leaky_relu': F.leaky_relu, 'elu': F.elu, 'hard_sigmoid': F.hardsigmoid, 'hard_swish': F.hardswish, 'mish': F.mish,
               'selu': F.selu, 'celu': F.celu, 'hard_mish': lambda x: 0.5 * x * (x + 2).clamp(min=0, max=2)}


def mlp(x, sd, prefix, act='gelu'):
    """MLP.forward, maxvit.py:85-118: Linear -> act -> Linear, or -- gated, read off the state dict -- GLU (:56-82: first half of the
    projection times act(second half)) -> Linear.  ``act``: the `mlp_activation` name (timm create_act.py:62-79)."""
    fn = ACTIVATIONS[act]
    if (prefix + '.net.0.proj.weight') in sd:
        a, g = torch.tensor_split(F.linear(x, sd[prefix + '.net.0.proj.weight'], sd.get(prefix + '.net.0.proj.bias')), 2, dim=-1)
        h = a * fn(g)
    else:
        h = fn(F.linear(x, sd[prefix + '.net.0.0.weight'], sd.get(prefix + '.net.0.0.bias')))
    return F.linear(h, sd[prefix + '.net.2.weight'], sd.get(prefix + '.net.2.bias'))


def partition_attention(x, sd, prefix, partition_size, window: bool, dim_head: int,
                        skip_first_norm: bool = False, act: str = 'gelu'):
    """PartitionAttentionCl.forward, maxvit.py:252-270.  x: [B,H,W,C] channels-last.  Options read off the state dict: no ``ls*.gamma`` =
    no LayerScale (ls_init_value 0), ``self_attn.mha.*`` = torch MHA layout, missing biases, gated MLP; ``act`` = `mlp_activation`."""
    hw = x.shape[1:3]
    n1 = x if skip_first_norm else layer_norm(x, sd, prefix + '.norm1')
    part = window_partition(n1, partition_size) if window else grid_partition(n1, partition_size)
    part = self_attention(part, sd, prefix + '.self_attn', dim_head)
    a = window_reverse(part, partition_size, hw) if window else grid_reverse(part, partition_size, hw)
    x = x + a * sd.get(prefix + '.ls1.gamma', 1.0)
    x = x + mlp(layer_norm(x, sd, prefix + '.norm2'), sd, prefix + '.mlp', act) * sd.get(prefix + '.ls2.gamma', 1.0)
    return x


def conv_downsample(x, sd, prefix, stride):
    """ConvDownsampling_Cf2Cl.forward, maxvit.py:160-178: conv (no bias; overlapping k=2s-1, pad k//2, or -- ``overlap=False`` -- k=s, pad 0:
    read off the kernel's parity) on NCHW -> NHWC -> LayerNorm(eps 1e-5; without weight / bias when ``norm_affine=False``)."""
    w = sd[prefix + '.conv.weight']
    k = w.shape[-1]
    y = F.conv2d(x, w, None, stride=stride, padding=k // 2 if k % 2 else 0).permute(0, 2, 3, 1)
    if (prefix + '.norm.weight') not in sd:
        return F.layer_norm(y, (y.shape[-1],), None, None, 1e-5)
    return layer_norm(y, sd, prefix + '.norm')


def conv_lstm(x, hc, sd, prefix):
    """DWSConvLSTM2d.forward, models/layers/rnn.py:37-70.  NCHW.  ``dws_conv`` (rnn.py:20-30,50-55) is read off the state dict: a
    ``conv3x3_dws.weight`` of C filters is the depthwise conv on h alone (``dws_conv_only_hidden``), of 2C filters the one on cat(x, h)."""
    if hc is None:
        hc = (torch.zeros_like(x), torch.zeros_like(x))
    h0, c0 = hc
    C = x.shape[1]
    wd = sd.get(prefix + '.conv3x3_dws.weight')
    if wd is not None and wd.shape[0] == C:
        h0 = F.conv2d(h0, wd, sd[prefix + '.conv3x3_dws.bias'], padding=wd.shape[-1] // 2, groups=C)
    xh = torch.cat((x, h0), dim=1)
    if wd is not None and wd.shape[0] == 2 * C:
        xh = F.conv2d(xh, wd, sd[prefix + '.conv3x3_dws.bias'], padding=wd.shape[-1] // 2, groups=2 * C)
    mix = F.conv2d(xh, sd[prefix + '.conv1x1.weight'], sd[prefix + '.conv1x1.bias'])
    gates, cell_in = torch.tensor_split(mix, [3 * C], dim=1)
    f, i, o = torch.tensor_split(torch.sigmoid(gates), 3, dim=1)
    g = torch.tanh(cell_in)
    c = f * c0 + i * g
    h = o * torch.tanh(c)
    return h, c


def stage_forward(x, hc, sd, prefix, stride, partition_size, dim_head, num_blocks=1, token_mask=None, act='gelu'):
    """RNNDetectorStage.forward, models/detection/recurrent_backbone/maxvit_rnn.py:182-201 (``token_mask`` [B,H,W] bool: :190-192)."""
    x = conv_downsample(x, sd, prefix + '.downsample_cf2cl', stride)
    if token_mask is not None:
        x = torch.where(token_mask[..., None], sd[prefix + '.mask_token'].reshape(1, 1, 1, -1).expand_as(x), x)
    for b in range(num_blocks):
        x = partition_attention(x, sd, f'{prefix}.att_blocks.{b}.att_window', partition_size, True, dim_head,
                                skip_first_norm=(b == 0), act=act)
        x = partition_attention(x, sd, f'{prefix}.att_blocks.{b}.att_grid', partition_size, False, dim_head, act=act)
    x = x.permute(0, 3, 1, 2).contiguous()
    h, c = conv_lstm(x, hc, sd, prefix + '.lstm')
    return h, (h, c)


def backbone_forward(x, prev_states, sd, cfg, prefix='backbone'):
    """RNNDetector.forward, maxvit_rnn.py:97-115.  cfg: dict(partition_size, dim_head, num_blocks,
    patch_size).  Returns ({1..4: feat NCHW}, [(h,c)]*4)."""
    if prev_states is None:
        prev_states = [None] * 4
    out: Dict[int, torch.Tensor] = {}
    states: List[Tuple[torch.Tensor, torch.Tensor]] = []
    nb = cfg.get('num_blocks', (1, 1, 1, 1))
    for s in range(4):
        stride = cfg.get('patch_size', 4) if s == 0 else 2
        x, st = stage_forward(x, prev_states[s], sd, f'{prefix}.stages.{s}', stride,
                              tuple(cfg['partition_size']), cfg['dim_head'], nb[s], act=cfg.get('mlp_activation', 'gelu'))
        states.append(st)
        out[s + 1] = x
    return out, states


def pad_ev_repr(ev, hw):
    """InputPadderFromShape.pad_tensor_ev_repr, utils/padding.py:32-58: zero-pad bottom/right."""
    H, W = ev.shape[-2:]
    return F.pad(ev, [0, hw[1] - W, 0, hw[0] - H])
