"""YOLOX conv blocks on channels-last maps, HIP-backed (mirror of the reference's
models/detection/yolox/models/network_blocks.py:29-142; same module tree / state-dict keys).
BaseConv = implicit-GEMM conv + BatchNorm2d + SiLU: in eval the BN is folded into the conv epilogue (one
kernel); in training the conv epilogue accumulates per-channel statistics and a second kernel applies
BN + SiLU.  Maps travel as contiguous [B,H,W,C] tensors between blocks."""
import torch
import torch.nn as nn

from leod_amd import functions as Fn


def get_activation(name="silu", inplace=True):
    if name == "silu":
        return nn.SiLU(inplace=inplace)
    raise NotImplementedError(f'activation {name}: the HIP path implements SiLU (the only one the configs use)')


class BaseConv(nn.Module):
    def __init__(self, in_channels, out_channels, ksize, stride, groups=1, bias=False, act="silu"):
        super().__init__()
        depthwise = groups > 1 and groups == in_channels == out_channels         # DWConv.dconv: one k x k filter per channel (k_dwconv.hip)
        if (groups != 1 and not depthwise) or bias or act != 'silu' or (ksize not in (1, 3) and not depthwise) or ksize % 2 == 0:
            raise NotImplementedError('HIP BaseConv: dense 1x1 / 3x3 or depthwise k x k conv, no bias, SiLU')
        if depthwise and in_channels % 4:
            raise NotImplementedError('HIP depthwise conv: channels must be a multiple of 4')
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=ksize, stride=stride, padding=(ksize - 1) // 2,
                              groups=groups, bias=False)
        self.bn = nn.BatchNorm2d(out_channels)
        self.act = get_activation(act, inplace=True)
        self.stride = stride
        self.bn_calls_pending = 0            # training forwards not yet added to bn.num_batches_tracked (Fn.flush_bn_counters)
        self.register_state_dict_pre_hook(lambda module, prefix, keep_vars: Fn.flush_bn_counters(module))

    def forward_nhwc(self, x):
        return Fn.BaseConvFn.apply(self, x, self.conv.weight, self.bn.weight, self.bn.bias, self.stride, self.training)

    def forward(self, x):
        """NCHW-logical in/out (reference signature)."""
        return Fn.as_nchw(self.forward_nhwc(Fn.to_nhwc(x)))


class DWConv(nn.Module):
    """Depthwise k x k BaseConv followed by a pointwise 1 x 1 BaseConv (network_blocks.py:57-76; keys ``dconv.*`` / ``pconv.*``)."""

    def __init__(self, in_channels, out_channels, ksize, stride=1, act="silu"):
        super().__init__()
        self.dconv = BaseConv(in_channels, in_channels, ksize=ksize, stride=stride, groups=in_channels, act=act)
        self.pconv = BaseConv(in_channels, out_channels, ksize=1, stride=1, groups=1, act=act)

    def forward_nhwc(self, x):
        return self.pconv.forward_nhwc(self.dconv.forward_nhwc(x))

    def forward(self, x):
        return Fn.as_nchw(self.forward_nhwc(Fn.to_nhwc(x)))


class Bottleneck(nn.Module):
    def __init__(self, in_channels, out_channels, shortcut=True, expansion=0.5, depthwise=False, act="silu"):
        super().__init__()
        hidden = int(out_channels * expansion)
        Conv = DWConv if depthwise else BaseConv
        self.conv1 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv2 = Conv(hidden, out_channels, 3, stride=1, act=act)
        self.use_add = shortcut and in_channels == out_channels

    def forward_nhwc(self, x):
        y = self.conv2.forward_nhwc(self.conv1.forward_nhwc(x))
        return y + x if self.use_add else y

    def forward(self, x):
        """NCHW-logical in/out (reference signature, network_blocks.py:111-115)."""
        return Fn.as_nchw(self.forward_nhwc(Fn.to_nhwc(x)))


class CSPLayer(nn.Module):
    def __init__(self, in_channels, out_channels, n=1, shortcut=True, expansion=0.5, depthwise=False, act="silu"):
        super().__init__()
        hidden = int(out_channels * expansion)
        self.conv1 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv2 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv3 = BaseConv(2 * hidden, out_channels, 1, stride=1, act=act)
        self.m = nn.Sequential(*[Bottleneck(hidden, hidden, shortcut, 1.0, depthwise, act=act) for _ in range(n)])

    def forward_nhwc(self, x):
        x1, x2 = Fn.base_conv_group([self.conv1, self.conv2], [x, x])     # same input, independent: one statistics exchange
        for b in self.m:
            x1 = b.forward_nhwc(x1)
        return self.conv3.forward_nhwc(Fn.cat2_nhwc(x1, x2))

    def forward(self, x):
        return Fn.as_nchw(self.forward_nhwc(Fn.to_nhwc(x)))
