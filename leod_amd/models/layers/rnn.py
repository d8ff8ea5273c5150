"""ConvLSTM cell of the RVT stages, HIP-backed (mirror of the reference's models/layers/rnn.py:7-70;
same class name, ctor arguments and ``conv1x1`` / ``conv3x3_dws`` state-dict keys).  The whole cell -- cat(x, h),
1x1 conv 2C->4C, gate non-linearities, state update -- is ONE kernel (leod_convlstm_fwd).  ``dws_conv=True`` (off in the RVT configs)
puts the depthwise k x k convolution of rnn.py:26-30 in front of it (leod_dwconv_nhwc_*): the recurrence then mixes pixels, so a sequence
walks the cell timestep by timestep instead of through the one-launch sequence kernels."""
from typing import Optional, Tuple

import torch as th
import torch.nn as nn

from leod_amd import functions as Fn


class DWSConvLSTM2d(nn.Module):
    def __init__(self, dim: int, dws_conv: bool = True, dws_conv_only_hidden: bool = True,
                 dws_conv_kernel_size: int = 3, cell_update_dropout: float = 0.):
        super().__init__()
        if cell_update_dropout > 0:
            raise NotImplementedError('HIP ConvLSTM: no cell-update dropout (drop_cell_update is 0 in every config)')
        self.dim = dim
        self.dws_conv = bool(dws_conv)
        dws_dim = dim if dws_conv_only_hidden else 2 * dim
        self.conv3x3_dws = nn.Conv2d(dws_dim, dws_dim, kernel_size=dws_conv_kernel_size, padding=dws_conv_kernel_size // 2,
                                     groups=dws_dim) if dws_conv else nn.Identity()
        if dws_conv and (dim % 4 or dws_conv_kernel_size % 2 == 0):
            raise NotImplementedError('HIP depthwise conv: channels a multiple of 4, odd kernel size')
        self.conv1x1 = nn.Conv2d(in_channels=dim * 2, out_channels=dim * 4, kernel_size=1)
        self.conv_only_hidden = dws_conv_only_hidden
        self.cell_update_dropout = nn.Dropout(p=0.)

    def forward(self, x: th.Tensor, h_and_c_previous: Optional[Tuple[th.Tensor, th.Tensor]] = None) \
            -> Tuple[th.Tensor, th.Tensor]:
        """x, h, c: [N,C,H,W] (logical NCHW; channels-last memory is used as is, anything else is copied once)."""
        xr = Fn.to_nhwc(x)
        if h_and_c_previous is None:
            if th.is_grad_enabled():
                h0 = c0 = th.zeros_like(xr)
            else:
                h0 = c0 = None
        else:
            h0, c0 = (Fn.to_nhwc(t) for t in h_and_c_previous)
        if self.dws_conv:
            if h0 is None:
                h0 = c0 = th.zeros_like(xr)
            xr, h0 = self._dws(xr, h0)
        h, c = Fn.ConvLSTMFn.apply(self, xr, h0, c0, self.conv1x1.weight, self.conv1x1.bias)
        return Fn.as_nchw(h), Fn.as_nchw(c)

    def _dws(self, x: th.Tensor, h: th.Tensor):
        """conv3x3_dws on h (``dws_conv_only_hidden``) or on cat(x, h) (rnn.py:50-55), NHWC in / out; zero states still see the bias."""
        conv, C = self.conv3x3_dws, self.dim
        if self.conv_only_hidden:
            return x, Fn.DepthwiseConvFn.apply(conv, h, conv.weight, conv.bias, 0, C)
        return (Fn.DepthwiseConvFn.apply(conv, x, conv.weight, conv.bias, 0, C),
                Fn.DepthwiseConvFn.apply(conv, h, conv.weight, conv.bias, C, 2 * C))

    def forward_sequence(self, x: th.Tensor, T: int, h_and_c_previous: Optional[Tuple[th.Tensor, th.Tensor]] = None):
        """Time-batched entry point: x [T*B,C,H,W] (logical NCHW, channels-last memory) holds the inputs of all T
        timesteps of a sequence batch.  Returns h of all timesteps [T*B,C,H,W] and the final (h, c) state --
        the same values as T calls of ``forward`` chained through the state."""
        xr = Fn.to_nhwc(x)
        TB, H, W, C = xr.shape
        assert TB % T == 0
        if h_and_c_previous is None:
            h0 = c0 = None
        else:
            h0, c0 = (Fn.to_nhwc(t) for t in h_and_c_previous)
        if self.dws_conv:                          # the depthwise conv couples neighbouring pixels across timesteps: walk the cell over t
            xs = xr.view(T, TB // T, H, W, C)
            h, c = (th.zeros_like(xs[0]), th.zeros_like(xs[0])) if h0 is None else (h0, c0)
            hs = []
            for t in range(T):
                xt, ht = self._dws(xs[t], h)
                h, c = Fn.ConvLSTMFn.apply(self, xt, ht, c, self.conv1x1.weight, self.conv1x1.bias)
                hs.append(h)
            h_seq = th.stack(hs)
            return Fn.as_nchw(h_seq.reshape(TB, H, W, C)), (Fn.as_nchw(h), Fn.as_nchw(c))
        h_seq, c_last = Fn.ConvLSTMSeqFn.apply(self, xr.view(T, TB // T, H, W, C), h0, c0, self.conv1x1.weight, self.conv1x1.bias)
        return Fn.as_nchw(h_seq.reshape(TB, H, W, C)), (Fn.as_nchw(h_seq[T - 1]), Fn.as_nchw(c_last))
