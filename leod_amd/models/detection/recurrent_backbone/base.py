"""Interface every recurrent backbone offers to the detector that sits on top of it
(reference: models/detection/recurrent_backbone/base.py): which channel widths and which
strides the requested stages (1-based) produce, so that PAFPN / head can be sized from it."""
import abc
from typing import Sequence, Tuple

from torch import nn


class BaseDetector(nn.Module, metaclass=abc.ABCMeta):
    """``stages`` are 1-based stage numbers, e.g. (2, 3, 4) for the three maps the PAFPN consumes."""

    @abc.abstractmethod
    def get_stage_dims(self, stages: Sequence[int]) -> Tuple[int, ...]:
        """Channel count of each requested stage."""

    @abc.abstractmethod
    def get_strides(self, stages: Sequence[int]) -> Tuple[int, ...]:
        """Down-sampling factor of each requested stage relative to the input frame."""
