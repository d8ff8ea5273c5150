"""A small read-only HDF5 reader for the one container the reference's datasets use: ``event_representations.h5`` with a
dataset ``data`` [N, 20, H, W] uint8, chunked one frame per chunk and compressed with the blosc filter (HDF5 filter id 32001, inner codec
zstd, byte shuffle) -- what ``h5py.File(fn, 'r')['data'][a:b]`` reads at /root/reference/data/genx_utils/sequence_base.py:184-193 (the
filter options are the reference's ``utils/preprocessing.py:4-15``).  h5py / hdf5plugin are not part of the MI355X image, so ``misc.H5Frames``
falls back to this module; with h5py installed it is not used.

Scope (everything else raises ``NotImplementedError`` with the feature named): the classic file format h5py writes by default --
superblock version 0 / 1, version-1 object headers (+ continuation blocks), old-style groups (symbol table: v1 B-tree + local heap),
dataspace v1 / v2, fixed-point and IEEE float datatypes, data layout v3 (compact, contiguous, chunked with a v1 B-tree chunk index),
filter pipeline v1 / v2 with shuffle (2), deflate (1) and blosc (32001: zstd, zlib, lz4 through pyarrow, or stored blocks).
Written from the published HDF5 File Format Specification (version 3.0) and the c-blosc 1.x frame format; no libhdf5-written file was
available in the build container, so parity with libhdf5 is pinned only through files built by ``tests/h5_writer.py`` from the same
specification -- stated in DESIGN.md."""
import ctypes
import ctypes.util
import mmap
import struct
import zlib
from typing import Dict, List, Optional, Tuple

import numpy as np

_SIG = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xFFFFFFFFFFFFFFFF


# ---- codecs ------------------------------------------------------------------------------------------------------------------------
_ZSTD = None


def _zstd():
    global _ZSTD
    if _ZSTD is None:
        name = ctypes.util.find_library('zstd') or 'libzstd.so.1'
        try:
            lib = ctypes.CDLL(name)
            lib.ZSTD_decompress.restype = ctypes.c_size_t
            lib.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
            lib.ZSTD_isError.restype = ctypes.c_uint
            lib.ZSTD_isError.argtypes = [ctypes.c_size_t]
            _ZSTD = lib
        except OSError:
            _ZSTD = False
    return _ZSTD


def zstd_decompress(src: bytes, n_out: int) -> bytes:
    lib = _zstd()
    if lib:
        dst = ctypes.create_string_buffer(n_out)
        got = lib.ZSTD_decompress(dst, n_out, src, len(src))
        if lib.ZSTD_isError(got) or got != n_out:
            raise IOError(f'zstd: corrupt block ({got} of {n_out} bytes)')
        return dst.raw
    import pyarrow as pa                                    # the image's pyarrow carries a zstd codec
    return pa.Codec('zstd').decompress(src, decompressed_size=n_out).to_pybytes()


def _lz4_block(src: bytes, n_out: int) -> bytes:
    import pyarrow as pa
    return pa.Codec('lz4_raw').decompress(src, decompressed_size=n_out).to_pybytes()


def _unshuffle(buf: bytes, typesize: int) -> bytes:
    """inverse of the byte shuffle (HDF5 filter 2 / blosc flag 0x1): byte j of element i sits at j * n + i"""
    n = len(buf) // typesize
    if typesize <= 1 or n == 0:
        return buf
    a = np.frombuffer(buf, np.uint8, n * typesize).reshape(typesize, n).T
    return a.tobytes() + buf[n * typesize:]


def blosc_decompress(src: bytes) -> bytes:
    """One c-blosc 1.x frame: 16-byte header (version, versionlz, flags, typesize, nbytes, blocksize, cbytes), then either the raw bytes
    (flag 0x2 'memcpyed') or a table of block starts followed by the blocks; a block is 1 (or ``typesize``, when split) streams of
    int32 compressed size + payload, a stream whose size equals its uncompressed size is stored."""
    if len(src) < 16:
        raise IOError('blosc: truncated frame')
    _ver, _verlz, flags, typesize = src[0], src[1], src[2], src[3]
    nbytes, blocksize, cbytes = struct.unpack_from('<III', src, 4)
    if flags & 0x2:
        return bytes(src[16:16 + nbytes])
    if flags & 0x4:
        raise NotImplementedError('blosc bit-shuffle')
    codec = flags >> 5
    dontsplit = bool(flags & 0x10)
    nblocks = (nbytes + blocksize - 1) // blocksize if blocksize else 0
    starts = struct.unpack_from(f'<{nblocks}i', src, 16)
    out = bytearray()
    for b in range(nblocks):
        bsize = min(blocksize, nbytes - b * blocksize)
        leftover = bsize != blocksize
        nsplit = typesize if (not dontsplit and 1 < typesize <= 16 and bsize // typesize >= 128 and not leftover) else 1
        neb = bsize // nsplit
        pos = starts[b]
        block = bytearray()
        for _ in range(nsplit):
            (csz,) = struct.unpack_from('<i', src, pos)
            pos += 4
            piece = src[pos:pos + csz]
            pos += csz
            if csz == neb:
                block += piece
            elif codec == 4:
                block += zstd_decompress(bytes(piece), neb)
            elif codec == 3:
                block += zlib.decompress(bytes(piece))
            elif codec == 1:
                block += _lz4_block(bytes(piece), neb)
            else:
                raise NotImplementedError(f'blosc inner codec {codec} (blosclz / snappy)')
        out += _unshuffle(bytes(block), typesize) if flags & 0x1 else block
    if len(out) != nbytes:
        raise IOError(f'blosc: {len(out)} bytes out of {nbytes}')
    return bytes(out)


# ---- file structure ------------------------------------------------------------------------------------------------------------------
class H5Dataset:
    def __init__(self, f: 'H5File', name: str, msgs: List[Tuple[int, bytes]]):
        self.file, self.name = f, name
        self.shape: Tuple[int, ...] = ()
        self.dtype = None
        self.filters: List[Tuple[int, Tuple[int, ...]]] = []
        self.layout = None
        self.fill = None
        for t, d in msgs:
            if t == 0x0001:
                self._dataspace(d)
            elif t == 0x0003:
                self._datatype(d)
            elif t == 0x0008:
                self._layout(d)
            elif t == 0x000B:
                self._pipeline(d)
        if self.dtype is None or self.layout is None:
            raise IOError(f'{f.fn}: object {name!r} is not a dataset')
        self._index: Optional[Dict[Tuple[int, ...], Tuple[int, int, int]]] = None

    # -- header messages --
    def _dataspace(self, d: bytes):
        ver, rank, flags = d[0], d[1], d[2]
        off = 8 if ver == 1 else 4
        if ver not in (1, 2):
            raise NotImplementedError(f'dataspace message version {ver}')
        self.shape = tuple(struct.unpack_from(f'<{rank}Q', d, off))

    def _datatype(self, d: bytes):
        cls, bits0 = d[0] & 0x0F, d[1]
        (size,) = struct.unpack_from('<I', d, 4)
        order = '>' if bits0 & 1 else '<'
        if cls == 0:
            self.dtype = np.dtype(f'{order}{"i" if bits0 & 0x08 else "u"}{size}')
        elif cls == 1:
            self.dtype = np.dtype(f'{order}f{size}')
        else:
            raise NotImplementedError(f'datatype class {cls} (only fixed-point and floating-point)')

    def _layout(self, d: bytes):
        ver, cls = d[0], d[1]
        if ver != 3:
            raise NotImplementedError(f'data layout message version {ver} (written with libver="latest"?)')
        if cls == 0:
            (n,) = struct.unpack_from('<H', d, 2)
            self.layout = ('compact', bytes(d[4:4 + n]))
        elif cls == 1:
            addr, size = struct.unpack_from('<QQ', d, 2)
            self.layout = ('contiguous', addr, size)
        elif cls == 2:
            ndim = d[2]
            (btree,) = struct.unpack_from('<Q', d, 3)
            dims = struct.unpack_from(f'<{ndim}I', d, 11)
            self.layout = ('chunked', btree, tuple(dims[:-1]), dims[-1])
        else:
            raise NotImplementedError(f'layout class {cls}')

    def _pipeline(self, d: bytes):
        ver, n = d[0], d[1]
        off = 8 if ver == 1 else 2
        for _ in range(n):
            fid, = struct.unpack_from('<H', d, off)
            off += 2
            if ver == 1 or fid >= 256:
                (nlen,) = struct.unpack_from('<H', d, off)
                off += 2
            else:
                nlen = 0
            _flags, ncd = struct.unpack_from('<HH', d, off)
            off += 4
            off += (nlen + 7) // 8 * 8 if ver == 1 else nlen
            cd = struct.unpack_from(f'<{ncd}I', d, off)
            off += 4 * ncd
            if ver == 1 and ncd % 2:
                off += 4
            self.filters.append((fid, tuple(cd)))

    # -- data --
    @property
    def chunks(self) -> Optional[Tuple[int, ...]]:
        return self.layout[2] if self.layout[0] == 'chunked' else None

    def __len__(self):
        return self.shape[0]

    def _decode(self, raw: bytes, mask: int) -> bytes:
        for i in reversed(range(len(self.filters))):
            if mask & (1 << i):
                continue
            fid, cd = self.filters[i]
            if fid == 32001:
                raw = blosc_decompress(raw)
            elif fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                raw = _unshuffle(raw, cd[0] if cd else self.dtype.itemsize)
            else:
                raise NotImplementedError(f'HDF5 filter {fid}')
        return raw

    def _walk(self, addr: int, rank: int, out: dict):
        m = self.file.mm
        if m[addr:addr + 4] != b'TREE':
            raise IOError(f'{self.file.fn}: no B-tree node at {addr}')
        ntype, level, used = m[addr + 4], m[addr + 5], struct.unpack_from('<H', m, addr + 6)[0]
        if ntype != 1:
            raise IOError('chunk index is not a raw-data B-tree')
        ksz = 8 + 8 * (rank + 1)
        pos = addr + 24
        for _ in range(used):
            size, mask = struct.unpack_from('<II', m, pos)
            offs = struct.unpack_from(f'<{rank}Q', m, pos + 8)
            (child,) = struct.unpack_from('<Q', m, pos + ksz)
            if level == 0:
                out[offs] = (child, size, mask)
            else:
                self._walk(child, rank, out)
            pos += ksz + 8

    def _chunk_index(self):
        if self._index is None:
            idx: Dict[Tuple[int, ...], Tuple[int, int, int]] = {}
            if self.layout[1] != UNDEF:
                self._walk(self.layout[1], len(self.shape), idx)
            self._index = idx
        return self._index

    def _read_rows(self, start: int, end: int, out: np.ndarray):
        """rows [start, end) of the first axis into ``out`` (C-contiguous, this dataset's dtype)"""
        kind = self.layout[0]
        row = int(np.prod(self.shape[1:], dtype=np.int64)) * self.dtype.itemsize
        flat = out.reshape(-1).view(np.uint8)
        if kind == 'contiguous' or kind == 'compact':
            src = self.file.mm[self.layout[1] + start * row: self.layout[1] + end * row] if kind == 'contiguous' \
                else self.layout[1][start * row:end * row]
            if kind == 'contiguous' and self.layout[1] == UNDEF:
                flat[:] = 0
            else:
                flat[:] = np.frombuffer(src, np.uint8)
            return
        cshape = self.layout[2]
        idx = self._chunk_index()
        if any(c != s for c, s in zip(cshape[1:], self.shape[1:])):
            raise NotImplementedError('chunks that split the trailing axes (the event representations are chunked frame by frame)')
        c0 = cshape[0]
        csize = int(np.prod(cshape, dtype=np.int64)) * self.dtype.itemsize
        zeros = (0,) * (len(self.shape) - 1)
        for k in range(start // c0, (end - 1) // c0 + 1):
            lo, hi = max(start, k * c0), min(end, (k + 1) * c0)
            ent = idx.get((k * c0,) + zeros)
            dst = flat[(lo - start) * row:(hi - start) * row]
            if ent is None:
                dst[:] = 0                                   # unallocated chunk: the fill value (0)
                continue
            addr, size, mask = ent
            raw = self._decode(bytes(self.file.mm[addr:addr + size]), mask) if self.filters else self.file.mm[addr:addr + size]
            if len(raw) < csize:
                raise IOError(f'{self.file.fn}: chunk at {addr} decodes to {len(raw)} bytes, expected {csize}')
            dst[:] = np.frombuffer(raw, np.uint8, csize)[(lo - k * c0) * row:(hi - k * c0) * row]

    def _norm(self, sel) -> Tuple[int, int]:
        if isinstance(sel, tuple):
            if len(sel) != 1 and any(s != slice(None) for s in sel[1:]):
                raise NotImplementedError('only selections along the first axis')
            sel = sel[0]
        if isinstance(sel, (int, np.integer)):
            i = int(sel) + (self.shape[0] if sel < 0 else 0)
            return i, i + 1
        if sel is Ellipsis:
            return 0, self.shape[0]
        a, b, step = sel.indices(self.shape[0])
        if step != 1:
            raise NotImplementedError('strided selections')
        return a, max(a, b)

    def __getitem__(self, sel) -> np.ndarray:
        a, b = self._norm(sel)
        out = np.empty((b - a,) + tuple(self.shape[1:]), self.dtype)
        if b > a:
            self._read_rows(a, b, out)
        return out[0] if isinstance(sel, (int, np.integer)) else out

    def read_direct(self, dest: np.ndarray, source_sel=None) -> None:
        a, b = self._norm(slice(None) if source_sel is None else source_sel)
        if dest.shape != (b - a,) + tuple(self.shape[1:]) or dest.dtype != self.dtype:
            raise ValueError('read_direct: destination shape / dtype mismatch')
        if dest.flags['C_CONTIGUOUS']:
            if b > a:
                self._read_rows(a, b, dest)
        else:
            np.copyto(dest, self[a:b])


class H5File:
    """``H5File(fn)['data']`` -> ``H5Dataset``; a context manager like ``h5py.File(fn, 'r')``."""

    def __init__(self, fn: str, mode: str = 'r'):
        if mode != 'r':
            raise ValueError('h5lite is read-only')
        self.fn = str(fn)
        self._f = open(self.fn, 'rb')
        try:
            self.mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        except ValueError as e:
            self._f.close()
            raise IOError(f'{fn}: empty file, not HDF5') from e
        try:
            self._root = self._superblock()
        except Exception:
            self.close()
            raise

    def _superblock(self) -> Dict[str, int]:
        m = self.mm
        base = 0
        while m[base:base + 8] != _SIG:                       # the superblock may sit at 0, 512, 1024, ...
            base = 512 if base == 0 else base * 2
            if base + 8 > len(m):
                raise IOError(f'{self.fn}: not an HDF5 file')
        ver = m[base + 8]
        if ver > 1:
            raise NotImplementedError(f'{self.fn}: superblock version {ver} (file written with libver="latest"); h5lite reads the '
                                      'classic format h5py writes by default -- install h5py for this file')
        if m[base + 13] != 8 or m[base + 14] != 8:
            raise NotImplementedError('offset / length sizes other than 8 bytes')
        pos = base + 24 + (4 if ver == 1 else 0)
        self.base = struct.unpack_from('<Q', m, pos)[0]
        entry = pos + 32                                      # root group symbol table entry
        _name_off, ohdr, cache = struct.unpack_from('<QQI', m, entry)
        if cache == 1:
            btree, heap = struct.unpack_from('<QQ', m, entry + 24)
        else:
            btree = heap = None
            for t, d in self._messages(ohdr):
                if t == 0x0011:
                    btree, heap = struct.unpack_from('<QQ', d, 0)
            if btree is None:
                raise NotImplementedError('root group without a symbol table (new-style group)')
        return self._group(btree, heap)

    def _messages(self, addr: int) -> List[Tuple[int, bytes]]:
        m = self.mm
        if m[addr:addr + 4] == b'OHDR':
            raise NotImplementedError('version-2 object headers (libver="latest")')
        if m[addr] != 1:
            raise IOError(f'{self.fn}: no object header at {addr}')
        nmsg, = struct.unpack_from('<H', m, addr + 2)
        (hsize,) = struct.unpack_from('<I', m, addr + 8)
        blocks = [(addr + 16, hsize)]
        out = []
        while blocks and len(out) < nmsg:
            pos, size = blocks.pop(0)
            end = pos + size
            while pos + 8 <= end and len(out) < nmsg:
                t, sz, _fl = struct.unpack_from('<HHB', m, pos)
                data = bytes(m[pos + 8:pos + 8 + sz])
                pos += 8 + sz
                if t == 0x0010:
                    off, ln = struct.unpack_from('<QQ', data, 0)
                    blocks.append((off + self.base, ln))
                out.append((t, data))
        return out

    def _group(self, btree: int, heap: int) -> Dict[str, int]:
        m = self.mm
        if m[heap:heap + 4] != b'HEAP':
            raise IOError('no local heap')
        (seg,) = struct.unpack_from('<Q', m, heap + 24)
        names: Dict[str, int] = {}

        def node(addr):
            if m[addr:addr + 4] != b'TREE' or m[addr + 4] != 0:
                raise IOError('bad group B-tree node')
            level, used = m[addr + 5], struct.unpack_from('<H', m, addr + 6)[0]
            pos = addr + 24 + 8
            for _ in range(used):
                (child,) = struct.unpack_from('<Q', m, pos)
                pos += 16
                if level:
                    node(child)
                    continue
                if m[child:child + 4] != b'SNOD':
                    raise IOError('bad symbol table node')
                (nsym,) = struct.unpack_from('<H', m, child + 6)
                for i in range(nsym):
                    noff, ohdr = struct.unpack_from('<QQ', m, child + 8 + 40 * i)
                    s = seg + noff
                    names[bytes(m[s:m.find(b'\x00', s)]).decode()] = ohdr
        node(btree)
        return names

    def keys(self):
        return list(self._root)

    def __contains__(self, name):
        return name in self._root

    def __getitem__(self, name: str) -> H5Dataset:
        if name not in self._root:
            raise KeyError(name)
        return H5Dataset(self, name, self._messages(self._root[name]))

    def close(self):
        if getattr(self, 'mm', None) is not None:
            self.mm.close()
            self.mm = None
        if getattr(self, '_f', None) is not None:
            self._f.close()
            self._f = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
