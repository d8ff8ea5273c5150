"""Buffering evaluator of the validation / test loops (reference: utils/evaluation/prophesee/evaluator.py:8-110).

The module calls ``add_labels`` / ``add_predictions`` once per step with one record array per labelled frame and
``evaluate_buffer`` at the end of the epoch; KPIs are computed for all classes together and per class."""
import warnings
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .evaluation import evaluate_list

LABELMAP = {'gen1': ('car', 'ped'), 'gen4': ('ped', 'cyc', 'car')}


def get_labelmap(dst_name: str = None, num_cls: int = None) -> Tuple[str]:
    """Short class names by dataset name or by class count (2 -> Gen1, 3 -> Gen4 / 1 Mpx)."""
    assert dst_name is None or num_cls is None
    if dst_name is not None:
        return LABELMAP[dst_name.lower()]
    if num_cls is None:
        raise NotImplementedError('Either dst_name or num_cls must be input')
    assert num_cls in (2, 3), f'Invalid number of classes: {num_cls}'
    return LABELMAP['gen1' if num_cls == 2 else 'gen4']


def _select_class(records: Sequence[np.ndarray], cls_id: int) -> List[np.ndarray]:
    return [r[r['class_id'] == cls_id] for r in records]


class PropheseeEvaluator:
    LABELS = 'lables'              # (sic) buffer keys as spelled in the reference
    PREDICTIONS = 'predictions'

    def __init__(self, dataset: str, downsample_by_2: bool):
        assert dataset in LABELMAP
        self.dataset, self.downsample_by_2 = dataset, downsample_by_2
        self.label_map = get_labelmap(dataset)
        self.reset_buffer()

    # ---- buffer ------------------------------------------------------------------------------------
    def reset_buffer(self) -> None:
        self._buffer: Dict[str, List[np.ndarray]] = {self.LABELS: [], self.PREDICTIONS: []}
        self._buffer_empty = True

    def _extend(self, key: str, records: List[np.ndarray]) -> None:
        assert isinstance(records, list) and all(isinstance(r, np.ndarray) for r in records)
        self._buffer[key] += records
        self._buffer_empty = False

    def add_labels(self, labels: List[np.ndarray]) -> None:
        self._extend(self.LABELS, labels)

    def add_predictions(self, predictions: List[np.ndarray]) -> None:
        self._extend(self.PREDICTIONS, predictions)

    def has_data(self) -> bool:
        return not self._buffer_empty

    # ---- KPIs ----------------------------------------------------------------------------------------
    def evaluate_buffer(self, img_height: int, img_width: int, ret_pr_curve: bool = False) -> Optional[Dict[str, Any]]:
        """``{'AP': .., 'AP_50': .., ...}`` over all classes plus ``'<key>_<class>'`` for each class (:70-110)."""
        if not self.has_data():
            warnings.warn('Attempt to use prophesee evaluation buffer, but it is empty', UserWarning, stacklevel=2)
            return None
        labels, predictions = self._buffer[self.LABELS], self._buffer[self.PREDICTIONS]
        assert len(labels) == len(predictions)

        def kpis(preds, gts):
            return evaluate_list(result_boxes_list=preds, gt_boxes_list=gts, height=img_height, width=img_width,
                                 camera=self.dataset, apply_bbox_filters=True, downsampled_by_2=self.downsample_by_2)

        metrics = kpis(predictions, labels)
        for cls_id, name in enumerate(self.label_map):
            per_class = kpis(_select_class(predictions, cls_id), _select_class(labels, cls_id))
            metrics.update((f'{k}_{name}', v) for k, v in per_class.items())
        if ret_pr_curve:
            return metrics
        return {k: v for k, v in metrics.items() if 'PR' not in k}
