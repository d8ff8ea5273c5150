from enum import Enum


class InterpolationMode(Enum):
    NEAREST = 'nearest'
    BILINEAR = 'bilinear'


from . import functional  # noqa
