"""Event representations built on the device (reference: data/utils/representations.py).

``StackedHistogram`` (representations.py:41-123) and ``MixedDensityEventStack`` (representations.py:132-221) keep the reference's
constructor arguments, ``construct(x, y, pol, time)``, ``get_shape`` and the dtype accessors; ``construct`` takes int64 device tensors and
runs ``leod_voxelize_u8`` / ``leod_mixed_density_i8`` (csrc/k_misc.hip).  There is no CPU path."""
from typing import Optional, Tuple

import numpy as np
import torch

from ... import ops


class RepresentationBase:
    def construct(self, x: torch.Tensor, y: torch.Tensor, pol: torch.Tensor, time: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def get_shape(self) -> Tuple[int, int, int]:
        raise NotImplementedError

    @staticmethod
    def get_numpy_dtype() -> np.dtype:
        raise NotImplementedError

    @staticmethod
    def get_torch_dtype() -> torch.dtype:
        raise NotImplementedError

    @property
    def dtype(self) -> torch.dtype:
        return self.get_torch_dtype()

    @staticmethod
    def _events(x, y, pol, time):
        n = x.numel()
        if not (y.numel() == pol.numel() == time.numel() == n):
            raise ValueError('x, y, pol, time must have the same number of events')
        for a in (x, y, pol, time):
            if a.is_floating_point() or a.dtype is torch.bool:
                raise TypeError('event coordinates, polarities and times are integer tensors')
        return tuple(a.to(torch.int64).contiguous() for a in (x, y, pol, time))


class StackedHistogram(RepresentationBase):
    """uint8 [2*bins, H, W]: event counts per polarity and linear time bin (representations.py:41-123)."""

    def __init__(self, bins: int, height: int, width: int, count_cutoff: Optional[int] = None, fastmode: bool = True):
        assert bins >= 1 and height >= 1 and width >= 1
        self.bins, self.height, self.width = bins, height, width
        self.count_cutoff = 255 if count_cutoff is None else min(int(count_cutoff), 255)
        assert self.count_cutoff >= 1
        self.fastmode = fastmode
        self.channels = 2

    @staticmethod
    def get_numpy_dtype() -> np.dtype:
        return np.dtype('uint8')

    @staticmethod
    def get_torch_dtype() -> torch.dtype:
        return torch.uint8

    def get_shape(self) -> Tuple[int, int, int]:
        return 2 * self.bins, self.height, self.width

    def merge_channel_and_bins(self, representation: torch.Tensor) -> torch.Tensor:
        assert representation.dim() == 4
        return representation.reshape(-1, self.height, self.width)

    def construct(self, x, y, pol, time):
        x, y, pol, time = self._events(x, y, pol, time)
        return ops.voxelize_u8(x, y, pol, time, self.bins, self.height, self.width, count_cutoff=self.count_cutoff,
                               fastmode=self.fastmode)


class MixedDensityEventStack(RepresentationBase):
    """int8 [bins, H, W]: polarity sums over logarithmically growing time windows ending at the last event (representations.py:132-221).
    ``allow_compilation`` is accepted and ignored (it selects torch.compile for the channel running sum in the reference)."""

    def __init__(self, bins: int, height: int, width: int, count_cutoff: Optional[int] = None, allow_compilation: bool = False):
        assert bins >= 1 and height >= 1 and width >= 1
        self.bins, self.height, self.width = bins, height, width
        if count_cutoff is not None:
            assert isinstance(count_cutoff, int) and 0 <= count_cutoff <= 2 ** 7 - 1
        self.count_cutoff = count_cutoff

    @staticmethod
    def get_numpy_dtype() -> np.dtype:
        return np.dtype('int8')

    @staticmethod
    def get_torch_dtype() -> torch.dtype:
        return torch.int8

    def get_shape(self) -> Tuple[int, int, int]:
        return self.bins, self.height, self.width

    def construct(self, x, y, pol, time):
        x, y, pol, time = self._events(x, y, pol, time)
        return ops.mixed_density_i8(x, y, pol, time, self.bins, self.height, self.width, count_cutoff=self.count_cutoff)
