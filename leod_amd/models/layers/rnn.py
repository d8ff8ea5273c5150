"""ConvLSTM cell of the RVT stages, HIP-backed (mirror of the reference's models/layers/rnn.py:7-70;
same class name, ctor arguments and ``conv1x1`` state-dict keys).  The whole cell -- cat(x, h),
1x1 conv 2C->4C, gate non-linearities, state update -- is ONE kernel (leod_convlstm_fwd)."""
from typing import Optional, Tuple

import torch as th
import torch.nn as nn

from leod_amd import functions as Fn


class DWSConvLSTM2d(nn.Module):
    def __init__(self, dim: int, dws_conv: bool = True, dws_conv_only_hidden: bool = True,
                 dws_conv_kernel_size: int = 3, cell_update_dropout: float = 0.):
        super().__init__()
        if dws_conv or cell_update_dropout > 0:
            raise NotImplementedError('HIP ConvLSTM implements the RVT config: dws_conv=False, no cell dropout')
        self.dim = dim
        self.conv3x3_dws = nn.Identity()
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
        h, c = Fn.ConvLSTMFn.apply(self, xr, h0, c0, self.conv1x1.weight, self.conv1x1.bias)
        return Fn.as_nchw(h), Fn.as_nchw(c)

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
        h_seq, c_last = Fn.ConvLSTMSeqFn.apply(self, xr.view(T, TB // T, H, W, C), h0, c0, self.conv1x1.weight, self.conv1x1.bias)
        return Fn.as_nchw(h_seq.reshape(TB, H, W, C)), (Fn.as_nchw(h_seq[T - 1]), Fn.as_nchw(c_last))
