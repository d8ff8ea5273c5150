"""Keys of the loader batch dictionaries and of the step outputs, with the member names (and auto() numbering) of the
reference's data/utils/types.py so that batches produced by its loaders index this package's modules unchanged."""
from enum import Enum
from typing import Dict, List, Optional, Tuple

import torch as th

# what a loader sample / batch may carry; numbering follows declaration order (1, 2, ...) like ``auto()``
DataType = Enum('DataType', [
    'PATH',                     # recording directory
    'EV_IDX',                   # index of the event representation inside the recording
    'EV_REPR', 'FLOW', 'IMAGE',
    'OBJLABELS', 'OBJLABELS_SEQ',
    'SKIPPED_OBJLABELS_SEQ',    # ground truth withheld from semi-supervised training
    'IS_PADDED_MASK', 'IS_FIRST_SAMPLE', 'IS_LAST_SAMPLE',
    'IS_REVERSED',              # the recording is delivered back to front (time-flip view)
    'TOKEN_MASK',
    'PRED_MASK', 'GT_MASK', 'PRED_PROBS',   # teacher-side bookkeeping of soft-label training
    'AUGM_STATE',
])

DatasetType = Enum('DatasetType', ['GEN1', 'GEN4'])
DatasetMode = Enum('DatasetMode', ['TRAIN', 'VALIDATION', 'TESTING'])
ObjDetOutput = Enum('ObjDetOutput', ['LABELS_PROPH', 'PRED_PROPH', 'EV_REPR', 'SKIP_VIZ'])


class DatasetSamplingMode(str, Enum):
    """String-valued (the config holds 'random' / 'stream' / 'mixed' and compares against these members)."""
    RANDOM = 'random'
    STREAM = 'stream'
    MIXED = 'mixed'

    def __str__(self) -> str:
        return self.value


LstmState = Optional[Tuple[th.Tensor, th.Tensor]]     # (h, c) of one stage, None before the first frame
LstmStates = List[LstmState]
FeatureMap = th.Tensor
BackboneFeatures = Dict[int, th.Tensor]               # 1-based stage number -> [B, C, H, W]
