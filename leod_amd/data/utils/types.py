"""Batch dictionary keys / enums of the reference loaders (data/utils/types.py) that the hot path reads."""
from enum import Enum, auto
from typing import Dict, List, Optional, Tuple

import torch as th


class DataType(Enum):
    PATH = auto()
    EV_IDX = auto()
    EV_REPR = auto()
    FLOW = auto()
    IMAGE = auto()
    OBJLABELS = auto()
    OBJLABELS_SEQ = auto()
    SKIPPED_OBJLABELS_SEQ = auto()
    IS_PADDED_MASK = auto()
    IS_FIRST_SAMPLE = auto()
    IS_LAST_SAMPLE = auto()
    IS_REVERSED = auto()
    TOKEN_MASK = auto()
    PRED_MASK = auto()
    GT_MASK = auto()
    PRED_PROBS = auto()
    AUGM_STATE = auto()


class DatasetSamplingMode(str, Enum):
    RANDOM = 'random'
    STREAM = 'stream'
    MIXED = 'mixed'

    def __str__(self):
        return self.value


class ObjDetOutput(Enum):
    LABELS_PROPH = auto()
    PRED_PROPH = auto()
    EV_REPR = auto()
    SKIP_VIZ = auto()


LstmState = Optional[Tuple[th.Tensor, th.Tensor]]
LstmStates = List[LstmState]
FeatureMap = th.Tensor
BackboneFeatures = Dict[int, th.Tensor]
