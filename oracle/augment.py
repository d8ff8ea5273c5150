"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement (numpy) of the tensor side of the reference's spatial augmentation
(data/utils/augmentor.py:229-249 zoom-out, :311-331 zoom-in, :396-401 flip): ``torch.flip``, slicing and
``interpolate(mode='nearest-exact')``, whose index rule is ATen's
``min(int(floorf((dst + 0.5f) * (float(in) / out))), in - 1)`` (aten/src/ATen/native/UpSample.h,
nearest_neighbor_exact_compute_source_index).  Pinned by tests/golden/g14_augment.npz, recorded from the reference.
"""
import numpy as np

F32 = np.float32


def nearest_exact_index(out_size: int, in_size: int) -> np.ndarray:
    scale = F32(in_size) / F32(out_size)
    dst = np.arange(out_size, dtype=F32)
    return np.minimum(np.floor((dst + F32(0.5)) * scale).astype(np.int64), in_size - 1)


def flip_lr(x: np.ndarray) -> np.ndarray:
    return x[..., ::-1].copy()


def zoom_in(x: np.ndarray, x0y0, factor: float) -> np.ndarray:
    """augmentor.py:311-331: crop the window of size int(H/f) x int(W/f) at (x0, y0), resize it to H x W."""
    H, W = x.shape[-2:]
    wh, ww = int(H / factor), int(W / factor)
    x0, y0 = x0y0
    win = x[..., y0:y0 + wh, x0:x0 + ww]
    iy, ix = nearest_exact_index(H, win.shape[-2]), nearest_exact_index(W, win.shape[-1])
    return win[..., iy[:, None], ix[None, :]].copy()


def zoom_out(x: np.ndarray, x0y0, factor: float) -> np.ndarray:
    """augmentor.py:229-249: resize to int(H/f) x int(W/f), paste at (x0, y0) on a zero canvas."""
    H, W = x.shape[-2:]
    wh, ww = int(H / factor), int(W / factor)
    iy, ix = nearest_exact_index(wh, H), nearest_exact_index(ww, W)
    win = x[..., iy[:, None], ix[None, :]]
    out = np.zeros_like(x)
    x0, y0 = x0y0
    out[..., y0:y0 + wh, x0:x0 + ww] = win
    return out


def apply(x: np.ndarray, hflip: bool, mode: int, x0: int, y0: int, factor: float) -> np.ndarray:
    """Order of RandomSpatialAugmentorGenX.__call__ (augmentor.py:455-476): flip, then zoom-in (1) or zoom-out (2)."""
    if hflip:
        x = flip_lr(x)
    if mode == 1:
        x = zoom_in(x, (x0, y0), factor)
    elif mode == 2:
        x = zoom_out(x, (x0, y0), factor)
    return x
