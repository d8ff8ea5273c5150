#!/usr/bin/env python
"""Write the raw ``.npy`` twin of a recording's HDF5 frame file (dataset 'data' [N,20,H,W] uint8, blosc-zstd chunks) once, so
that the loaders can memory-map / pread it (leod_amd.data.utils.misc.RawFrames).  Needs h5py (+ hdf5plugin); run it wherever the
dataset was pre-processed.   usage: h5_to_npy.py <dataset root | split dir | recording dir> ..."""
import glob
import os
import sys

import numpy as np


def convert(h5_fn: str, block: int = 256) -> str:
    import h5py
    try:
        import hdf5plugin  # noqa: F401
    except ImportError:
        pass
    out = os.path.splitext(h5_fn)[0] + '.npy'
    if os.path.exists(out):
        return out
    with h5py.File(h5_fn, 'r') as f:
        d = f['data']
        assert d.dtype == np.uint8 and d.ndim == 4, (h5_fn, d.dtype, d.shape)
        mm = np.lib.format.open_memmap(out + '.tmp', mode='w+', dtype=np.uint8, shape=d.shape)
        for lo in range(0, d.shape[0], block):
            mm[lo:lo + block] = d[lo:lo + block]
        mm.flush()
        del mm
    os.replace(out + '.tmp', out)
    return out


if __name__ == '__main__':
    for root in sys.argv[1:]:
        for fn in sorted(glob.glob(os.path.join(root, '**', 'event_representations*.h5'), recursive=True)):
            print(convert(os.path.realpath(fn)))
