from .boxes import postprocess, bboxes_iou, postprocess_padded  # noqa: F401
