"""Buffering evaluator of the validation / test loops (reference: utils/evaluation/prophesee/evaluator.py:8-110)."""
from typing import Any, Dict, List, Optional, Tuple
from warnings import warn

import numpy as np

from .evaluation import evaluate_list

LABELMAP = {'gen1': ('car', 'ped'), 'gen4': ('ped', 'cyc', 'car')}


def get_labelmap(dst_name: str = None, num_cls: int = None) -> Tuple[str]:
    assert dst_name is None or num_cls is None
    if dst_name is not None:
        return LABELMAP[dst_name.lower()]
    if num_cls is not None:
        assert num_cls in (2, 3), f'Invalid number of classes: {num_cls}'
        return LABELMAP['gen1' if num_cls == 2 else 'gen4']
    raise NotImplementedError('Either dst_name or num_cls must be input')


class PropheseeEvaluator:
    LABELS = 'lables'              # (sic) key names kept from the reference
    PREDICTIONS = 'predictions'

    def __init__(self, dataset: str, downsample_by_2: bool):
        assert dataset in LABELMAP
        self.dataset = dataset
        self.label_map = get_labelmap(dataset)
        self.downsample_by_2 = downsample_by_2
        self._reset_buffer()

    def _reset_buffer(self):
        self._buffer_empty = True
        self._buffer = {self.LABELS: [], self.PREDICTIONS: []}

    def _add_to_buffer(self, key: str, value: List[np.ndarray]):
        assert isinstance(value, list) and all(isinstance(v, np.ndarray) for v in value)
        self._buffer_empty = False
        self._buffer[key].extend(value)

    def _get_from_buffer(self, key: str) -> List[np.ndarray]:
        assert not self._buffer_empty
        return self._buffer[key]

    def add_predictions(self, predictions: List[np.ndarray]):
        self._add_to_buffer(self.PREDICTIONS, predictions)

    def add_labels(self, labels: List[np.ndarray]):
        self._add_to_buffer(self.LABELS, labels)

    def reset_buffer(self) -> None:
        self._reset_buffer()

    def has_data(self) -> bool:
        return not self._buffer_empty

    def evaluate_buffer(self, img_height: int, img_width: int, ret_pr_curve: bool = False) -> Optional[Dict[str, Any]]:
        """Overall KPIs plus ``<key>_<class>`` for every class (:70-110)."""
        if self._buffer_empty:
            warn('Attempt to use prophesee evaluation buffer, but it is empty', UserWarning, stacklevel=2)
            return None
        labels, predictions = self._get_from_buffer(self.LABELS), self._get_from_buffer(self.PREDICTIONS)
        assert len(labels) == len(predictions)
        kw = dict(height=img_height, width=img_width, apply_bbox_filters=True, downsampled_by_2=self.downsample_by_2,
                  camera=self.dataset)
        metrics = evaluate_list(result_boxes_list=predictions, gt_boxes_list=labels, **kw)
        for cls_id, cls_name in enumerate(self.label_map):
            per_cls = evaluate_list(result_boxes_list=[p[p['class_id'] == cls_id] for p in predictions],
                                    gt_boxes_list=[l[l['class_id'] == cls_id] for l in labels], **kw)
            metrics.update({f'{k}_{cls_name}': v for k, v in per_cls.items()})
        if not ret_pr_curve:
            metrics = {k: v for k, v in metrics.items() if 'PR' not in k}
        return metrics
