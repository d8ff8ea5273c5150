"""On-disk layout of a (pseudo-)labelled recording and the frame store behind it (the interface of the reference's
data/utils/misc.py:11-99):

    <split>/<recording>/
        event_representations_v2/stacked_histogram_dt=50_nbins=10/
            event_representations.h5 | event_representations_ds2_nearest.h5    'data' [N,20,H,W] uint8 (blosc-zstd HDF5)
            event_representations.npy | event_representations_ds2_nearest.npy   the same array as a raw .npy (memory-mapped)
            objframe_idx_2_repr_idx.npy                                         int64 [n_labelled_frames]
        labels_v2/labels.npz                                                    labels (BBOX_DTYPE), objframe_idx_2_label_idx

The reference reads the frames through h5py + hdf5plugin.  Neither is in the MI355X image, and a compressed HDF5 chunk
store cannot feed >= 1.5 GB/s of uint8 voxels per GPU from Python anyway, so the frame store has two backends behind one
interface: ``RawFrames`` (np.load(mmap_mode='r') of the .npy twin: page-cache reads straight into pinned batches) and
``H5Frames`` (import-guarded h5py).  ``tools/h5_to_npy.py`` writes the twin once per recording."""
import os
from typing import Any,  List, Optional, Tuple

import numpy as np

EV_REPR_NAME = 'stacked_histogram_dt=50_nbins=10'


def get_labels_npz_fn(seq_dir: str) -> str:
    return os.path.join(seq_dir, 'labels_v2', 'labels.npz')


def read_npz_labels(label_fn: str) -> Tuple[np.ndarray, np.ndarray]:
    if 'labels_v2' not in label_fn:
        label_fn = get_labels_npz_fn(label_fn)
    with np.load(label_fn) as z:
        return z['labels'], z['objframe_idx_2_label_idx']


def get_ev_dir(seq_dir: str) -> str:
    return os.path.join(seq_dir, 'event_representations_v2', EV_REPR_NAME)


def get_objframe_idx_2_repr_idx_fn(ev_dir: str) -> str:
    if 'event_representations_v2' not in ev_dir:
        ev_dir = get_ev_dir(ev_dir)
    return os.path.join(ev_dir, 'objframe_idx_2_repr_idx.npy')


def read_objframe_idx_2_repr_idx(npy_fn: str) -> np.ndarray:
    if 'event_representations_v2' not in npy_fn:
        npy_fn = get_objframe_idx_2_repr_idx_fn(npy_fn)
    return np.load(npy_fn)


def get_ev_h5_fn(ev_dir: str, dst_name: Optional[str] = None) -> str:
    if 'event_representations_v2' not in ev_dir:
        ev_dir = get_ev_dir(ev_dir)
    if dst_name is None:
        dst_name = 'gen1' if 'gen1' in ev_dir else 'gen4'
    name = 'event_representations.h5' if dst_name == 'gen1' else 'event_representations_ds2_nearest.h5'
    return os.path.join(ev_dir, name)


def get_ev_raw_fn(ev_dir: str, dst_name: Optional[str] = None) -> str:
    """The raw twin of the HDF5 frame file (same stem, ``.npy``)."""
    return os.path.splitext(get_ev_h5_fn(ev_dir, dst_name))[0] + '.npy'


def resolve_link(fn: str) -> str:
    while os.path.islink(fn):
        fn = os.readlink(fn)
    return fn


class RawFrames:
    """[N,20,H,W] uint8 frames of one recording in a .npy file.  Reads into a caller buffer go through ``pread`` (the kernel
    copies from the page cache straight into the destination, e.g. a slot of a pinned batch: no per-page mapping faults on the
    source side, GIL released); ``data`` is the memory-mapped array for everything else."""

    def __init__(self, fn: str):
        self.fn = fn
        self.data = np.load(fn, mmap_mode='r')
        assert self.data.dtype == np.uint8 and self.data.ndim == 4 and self.data.flags['C_CONTIGUOUS'], (fn, self.data.dtype, self.data.shape)
        self._offset = int(self.data.offset)
        self._frame_bytes = int(np.prod(self.data.shape[1:]))
        self._fd = os.open(fn, os.O_RDONLY)

    @property
    def shape(self):
        return self.data.shape

    def __len__(self):
        return self.data.shape[0]

    def read(self, start: int, end: int, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Frames [start, end) -> ``out`` [n,C,H,W] (each frame contiguous, e.g. ``batch[:, b]``) or a new array."""
        assert 0 <= start < end <= len(self)
        if out is None:
            out = np.empty((end - start,) + tuple(self.data.shape[1:]), dtype=np.uint8)
        assert out.shape == (end - start,) + tuple(self.data.shape[1:]) and out.dtype == np.uint8
        if out.flags['C_CONTIGUOUS']:
            self._pread(memoryview(out.reshape(-1)), self._offset + start * self._frame_bytes)
        elif out[0].flags['C_CONTIGUOUS']:
            for k in range(end - start):
                self._pread(memoryview(out[k].reshape(-1)), self._offset + (start + k) * self._frame_bytes)
        else:
            np.copyto(out, self.data[start:end])
        return out

    def _pread(self, mv, offset: int) -> None:
        done, total = 0, len(mv)
        while done < total:
            n = os.preadv(self._fd, [mv[done:]], offset + done)
            if n <= 0:
                raise IOError(f'{self.fn}: short read at byte {offset + done}')
            done += n

    def close(self):
        self.data = None
        if self._fd is not None:
            os.close(self._fd)
            self._fd = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class H5Frames:
    """Same interface over the reference's HDF5 files (dataset 'data', sequence_base.py:184-193): through h5py (+ hdf5plugin for the blosc
    chunks) where it is installed, else through the package's own reader of that container (``h5lite``: classic HDF5 layout, blosc-zstd)."""

    def __init__(self, fn: str):
        self.fn = fn
        try:
            import h5py
        except ImportError:
            h5py = None
        if h5py is not None:
            try:                                                  # pragma: no cover
                import hdf5plugin  # noqa: F401
            except ImportError:
                pass
            self.h5f = h5py.File(fn, 'r')
        else:
            from . import h5lite
            self.h5f = h5lite.H5File(fn)
        self.data = self.h5f['data']

    @property
    def shape(self):
        return tuple(self.data.shape)

    def __len__(self):
        return self.data.shape[0]

    def read(self, start: int, end: int, out: Optional[np.ndarray] = None) -> np.ndarray:
        if out is None:
            return self.data[start:end]
        self.data.read_direct(out, np.s_[start:end])
        return out

    def close(self):
        self.h5f.close()


class FrameStoreCache:
    """At most ``capacity`` open frame stores per process, shared by every sequence object over the same file.

    A recording is wrapped by one ``SequenceForIter`` per labelled run plus one random-access sequence, and full Gen1 mixed-mode
    training builds several thousand of them; each used to keep its own fd + mmap for life, past the usual 1024 RLIMIT_NOFILE
    (the reference opens the HDF5 file per read, sequence_base.py:184-193).  Eviction only drops the cache's reference: a
    reader that is still inside ``read`` holds its own, and the store closes itself when the last one goes."""

    def __init__(self, capacity: int = 128):
        import collections
        import threading
        self.capacity = capacity
        self._stores: 'collections.OrderedDict[str, Any]' = collections.OrderedDict()
        self._lock = threading.Lock()

    def get(self, fn: str):
        with self._lock:
            st = self._stores.get(fn)
            if st is not None:
                self._stores.move_to_end(fn)
                return st
            st = RawFrames(fn) if fn.endswith('.npy') else H5Frames(fn)
            self._stores[fn] = st
            while len(self._stores) > self.capacity:
                self._stores.popitem(last=False)
            return st

    def __len__(self):
        return len(self._stores)

    def clear(self):
        with self._lock:
            self._stores.clear()


FRAME_STORES = FrameStoreCache()


def read_frame_header(fn: str):
    """(number of frames, (C, H, W)) of a frame file without keeping it open."""
    if fn.endswith('.npy'):
        with open(fn, 'rb') as f:
            major, _ = np.lib.format.read_magic(f)
            shape, fortran, dtype = (np.lib.format.read_array_header_1_0 if major == 1 else np.lib.format.read_array_header_2_0)(f)
        assert dtype == np.uint8 and len(shape) == 4 and not fortran, (fn, dtype, shape)
        return int(shape[0]), tuple(int(x) for x in shape[1:])
    st = H5Frames(fn)
    try:
        return len(st), tuple(st.shape[1:])
    finally:
        st.close()


def open_ev_repr(seq_or_ev_dir: str, dst_name: Optional[str] = None):
    """Frame store of a recording: the raw .npy twin when it exists, else the HDF5 file."""
    raw = resolve_link(get_ev_raw_fn(seq_or_ev_dir, dst_name))
    if os.path.exists(raw):
        return RawFrames(raw)
    h5 = resolve_link(get_ev_h5_fn(seq_or_ev_dir, dst_name))
    if os.path.exists(h5):
        return H5Frames(h5)
    raise FileNotFoundError(f'no event representation under {seq_or_ev_dir} (looked for {raw} and {h5})')


def ev_repr_files(seq_or_ev_dir: str, dst_name: Optional[str] = None) -> List[str]:
    """The frame files that exist for a recording (link targets resolved), raw twin first."""
    out = []
    for fn in (get_ev_raw_fn(seq_or_ev_dir, dst_name), get_ev_h5_fn(seq_or_ev_dir, dst_name)):
        if os.path.lexists(fn):
            out.append(fn)
    return out


def read_ev_repr(fn: str) -> np.ndarray:
    """All frames of a recording as one array [N, 20, H, W] (misc.py:83-88 of the reference reads the HDF5 file; here the raw twin is preferred)."""
    if 'event_representations_v2' not in fn:
        fn = get_ev_h5_fn(fn)
    raw = resolve_link(os.path.splitext(fn)[0] + '.npy')
    if os.path.exists(raw):
        return np.load(raw)
    store = H5Frames(resolve_link(fn))
    try:
        return store.data[:]
    finally:
        store.close()


def read_labels_as_list(seq_dir: str, dst_cfg, L: int, start_idx: int = 0) -> List[Any]:
    """The full-frequency labels of frames start_idx .. start_idx + L - 1 of a recording: ``ObjectLabels`` where a frame is labelled, None
    elsewhere (misc.py:28-46; what vis_pred.py pairs predictions with)."""
    from leod_amd.data.genx_utils.labels import ObjectLabelFactory
    labels, objframe_idx_2_label_idx = read_npz_labels(seq_dir)
    hw = tuple(dst_cfg.ev_repr_hw)
    ds2 = bool(dst_cfg.downsample_by_factor_2)
    if ds2:
        hw = tuple(s * 2 for s in hw)
    factory = ObjectLabelFactory.from_structured_array(labels, objframe_idx_2_label_idx, hw, 2 if ds2 else None)
    out: List[Any] = [None] * L
    for objframe_idx, repr_idx in enumerate(read_objframe_idx_2_repr_idx(seq_dir)):
        if start_idx <= repr_idx < start_idx + L:
            out[int(repr_idx) - start_idx] = factory[objframe_idx]
    return out
