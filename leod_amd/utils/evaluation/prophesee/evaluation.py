"""Entry point of the Prophesee evaluation protocol (reference: utils/evaluation/prophesee/evaluation.py:5-42)."""
from .io.box_filtering import filter_boxes
from .metrics.coco_eval import evaluate_detection

CLASSES = {'gen1': ('car', 'pedestrian'), 'gen4': ('pedestrian', 'two-wheeler', 'car')}


def evaluate_list(result_boxes_list, gt_boxes_list, height: int, width: int, camera: str = 'gen1',
                  apply_bbox_filters: bool = True, downsampled_by_2: bool = False, return_aps: bool = True):
    """Labels AND detections go through the paper's filters (min diagonal 30 / side 10 on Gen1, 60 / 20 on 1 Mpx,
    halved for the downsampled frames; nothing before 0.5 s), then the COCO KPIs."""
    assert camera in CLASSES
    if apply_bbox_filters:
        min_box_diag, min_box_side = (60, 20) if camera == 'gen4' else (30, 10)
        if downsampled_by_2:
            assert min_box_diag % 2 == 0 and min_box_side % 2 == 0
            min_box_diag, min_box_side = min_box_diag // 2, min_box_side // 2
        half_sec_us = int(5e5)
        gt_boxes_list = [filter_boxes(b, half_sec_us, min_box_diag, min_box_side) for b in gt_boxes_list]
        result_boxes_list = [filter_boxes(b, half_sec_us, min_box_diag, min_box_side) for b in result_boxes_list]
    return evaluate_detection(gt_boxes_list, result_boxes_list, height=height, width=width, classes=CLASSES[camera],
                              return_aps=return_aps)
