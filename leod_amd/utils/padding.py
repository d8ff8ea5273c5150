"""Input padding helper with the reference's interface (utils/padding.py:7-71).  On the HIP path the
bottom/right zero padding of the event tensor is folded into the stem-conv bounds predicate, so
``pad_tensor_ev_repr`` is only needed by callers that want the padded tensor itself."""
from typing import Tuple

import torch as th
import torch.nn.functional as F


class InputPadderFromShape:
    def __init__(self, desired_hw: Tuple[int, int], mode: str = 'constant', value: int = 0, type: str = 'corner'):
        assert isinstance(desired_hw, tuple) and len(desired_hw) == 2
        assert desired_hw[0] % 4 == 0 and desired_hw[1] % 4 == 0
        assert type == 'corner'
        self.desired_hw, self.mode, self.value = desired_hw, mode, value

    def _pad(self, t: th.Tensor, hw):
        ht, wd = t.shape[-2:]
        assert ht <= hw[0] and wd <= hw[1]
        return F.pad(t, [0, hw[1] - wd, 0, hw[0] - ht], mode=self.mode, value=self.value if self.mode == 'constant' else None)

    def pad_tensor_ev_repr(self, ev_repr: th.Tensor) -> th.Tensor:
        return self._pad(ev_repr, self.desired_hw)

    def pad_token_mask(self, token_mask: th.Tensor):
        return self._pad(token_mask, tuple(x // 4 for x in self.desired_hw))
