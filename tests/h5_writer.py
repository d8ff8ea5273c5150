"""Test infrastructure: writes the classic HDF5 layout h5py produces by default for the reference's event-representation files -- superblock
version 0, an old-style root group (symbol table: v1 B-tree, local heap, one symbol-table node), one dataset with a version-1 object
header (dataspace v1 with maximum dimensions, fixed-point datatype, fill value, data layout v3, filter pipeline v1), and for chunked data a v1
B-tree chunk index of as many levels as the chunk count needs -- from the HDF5 File Format Specification 3.0, with the chunks compressed
as c-blosc 1.x frames (zstd through libzstd, the reference's ``_blosc_opts``: utils/preprocessing.py:4-15), deflate, or stored.
h5py is not in the image; this writer exists so that ``leod_amd/data/utils/h5lite.py`` (and ``misc.H5Frames`` on top of it) reads real
bytes of that format in the tests.  It is NOT libhdf5: agreement of the two in-repo implementations of the specification is what the tests
show."""
import ctypes
import ctypes.util
import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


def _zstd_compress(raw: bytes, level: int = 1) -> bytes:
    lib = ctypes.CDLL(ctypes.util.find_library('zstd') or 'libzstd.so.1')
    lib.ZSTD_compressBound.restype = ctypes.c_size_t
    lib.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
    lib.ZSTD_compress.restype = ctypes.c_size_t
    lib.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    cap = lib.ZSTD_compressBound(len(raw))
    dst = ctypes.create_string_buffer(cap)
    n = lib.ZSTD_compress(dst, cap, raw, len(raw), level)
    return dst.raw[:n]


def blosc_frame(raw: bytes, typesize: int = 1, codec: str = 'zstd', shuffle: bool = True, blocksize: int = 0, split: bool = False) -> bytes:
    """A c-blosc 1.x frame of ``raw``: header, block-start table, blocks of [int32 size][payload] streams (stored when incompressible)."""
    nbytes = len(raw)
    blocksize = blocksize or min(nbytes, 32 * 1024) or 1
    code = {'blosclz': 0, 'lz4': 1, 'snappy': 2, 'zlib': 3, 'zstd': 4}[codec]
    do_shuffle = shuffle and typesize > 1
    flags = (code << 5) | (0x1 if do_shuffle else 0) | (0 if split else 0x10)
    nblocks = (nbytes + blocksize - 1) // blocksize
    body, starts = bytearray(), []
    pos0 = 16 + 4 * nblocks
    for b in range(nblocks):
        blk = raw[b * blocksize:(b + 1) * blocksize]
        if do_shuffle:
            n = len(blk) // typesize
            blk = np.frombuffer(blk, np.uint8, n * typesize).reshape(n, typesize).T.tobytes() + blk[n * typesize:]
        leftover = len(blk) != blocksize
        nsplit = typesize if (split and 1 < typesize <= 16 and len(blk) // typesize >= 128 and not leftover) else 1
        neb = len(blk) // nsplit
        starts.append(pos0 + len(body))
        for s in range(nsplit):
            piece = blk[s * neb:(s + 1) * neb]
            comp = _zstd_compress(piece) if codec == 'zstd' else zlib.compress(piece, 1) if codec == 'zlib' else piece
            if len(comp) >= len(piece):
                comp = piece
            body += struct.pack('<i', len(comp)) + comp
    frame = struct.pack('<BBBBIII', 2, 1, flags, typesize, nbytes, blocksize, pos0 + len(body)) + struct.pack(f'<{nblocks}i', *starts) + bytes(body)
    return frame


def _pad8(b: bytes) -> bytes:
    return b + b'\x00' * (-len(b) % 8)


def _msg(t: int, data: bytes) -> bytes:
    data = _pad8(data)
    return struct.pack('<HHBBBB', t, len(data), 0, 0, 0, 0) + data


def write_h5(fn: str, arr: np.ndarray, name: str = 'data', chunks=None, compression: str = 'blosc', istore_k: int = 32, missing=()):
    """``arr`` as dataset ``name`` of a new file.  ``chunks`` None: contiguous; a tuple (rows per chunk, *trailing shape): chunked with
    ``compression`` in {'blosc' (zstd), 'blosc-zlib', 'gzip', 'shuffle-gzip', None}; ``missing``: chunk numbers left unallocated (they read
    back as the fill value 0); ``istore_k``: half the fan-out of the chunk B-tree (h5py's default 32)."""
    arr = np.ascontiguousarray(arr)
    rank = arr.ndim
    buf = bytearray(b'\x00' * 96)                                   # superblock (56 + 40 bytes) is filled in at the end

    def alloc(b: bytes) -> int:
        addr = len(buf)
        buf.extend(_pad8(b))
        return addr

    # ---- raw data ----
    filters = []
    if chunks is None:
        data_addr = alloc(arr.tobytes())
        layout = struct.pack('<BBQQ', 3, 1, data_addr, arr.nbytes)
    else:
        assert tuple(chunks[1:]) == arr.shape[1:]
        c0 = chunks[0]
        entries = []
        for k in range((arr.shape[0] + c0 - 1) // c0):
            if k in missing:
                continue
            blk = np.zeros(chunks, arr.dtype)
            part = arr[k * c0:(k + 1) * c0]
            blk[:len(part)] = part
            raw = blk.tobytes()
            if compression in ('blosc', 'blosc-zlib'):
                raw = blosc_frame(raw, arr.dtype.itemsize, 'zstd' if compression == 'blosc' else 'zlib')
            elif compression == 'shuffle-gzip':
                n = len(raw) // arr.dtype.itemsize
                raw = zlib.compress(np.frombuffer(raw, np.uint8).reshape(n, arr.dtype.itemsize).T.tobytes(), 4)
            elif compression == 'gzip':
                raw = zlib.compress(raw, 4)
            entries.append(((k * c0,) + (0,) * rank, len(raw), alloc(raw)))           # key offsets have rank + 1 entries
        ksz = 8 + 8 * (rank + 1)

        def node(level, items, last_key):
            body = b''
            for offs, size, child in items:
                body += struct.pack('<II', size, 0) + struct.pack(f'<{rank + 1}Q', *offs) + struct.pack('<Q', child)
            body += struct.pack('<II', 0, 0) + struct.pack(f'<{rank + 1}Q', *last_key)
            body += b'\x00' * ((2 * istore_k) * (ksz + 8) + ksz - len(body))       # nodes are allocated at full size
            return alloc(b'TREE' + struct.pack('<BBHQQ', 1, level, len(items), UNDEF, UNDEF) + body)

        end_key = (((arr.shape[0] + c0 - 1) // c0) * c0,) + (0,) * rank
        level, items = 0, entries
        btree = UNDEF
        while items:
            groups = [items[i:i + 2 * istore_k] for i in range(0, len(items), 2 * istore_k)]
            nodes = []
            for gi, g in enumerate(groups):
                lk = groups[gi + 1][0][0] if gi + 1 < len(groups) else end_key
                nodes.append((g[0][0], g[0][1], node(level, g, lk)))
            if len(nodes) == 1:
                btree = nodes[0][2]
                break
            level, items = level + 1, nodes
        layout = struct.pack('<BBB', 3, 2, rank + 1) + struct.pack('<Q', btree) + struct.pack(f'<{rank + 1}I', *chunks, arr.dtype.itemsize)
        if compression in ('blosc', 'blosc-zlib'):
            nm = _pad8(b'blosc\x00')
            cd = (2, 2, arr.dtype.itemsize, int(np.prod(chunks)) * arr.dtype.itemsize, 1, 1, 5 if compression == 'blosc' else 4)
            filters.append(struct.pack('<HHHH', 32001, len(nm), 1, len(cd)) + nm + struct.pack(f'<{len(cd)}I', *cd) + (b'\x00' * 4 if len(cd) % 2 else b''))
        elif compression == 'shuffle-gzip':
            filters.append(struct.pack('<HHHH', 2, 0, 1, 1) + struct.pack('<I', arr.dtype.itemsize) + b'\x00' * 4)
            filters.append(struct.pack('<HHHH', 1, 0, 1, 1) + struct.pack('<I', 4) + b'\x00' * 4)
        elif compression == 'gzip':
            filters.append(struct.pack('<HHHH', 1, 0, 1, 1) + struct.pack('<I', 4) + b'\x00' * 4)

    # ---- dataset object header (version 1) ----
    kind = arr.dtype.kind
    if kind in 'ui':
        dt = struct.pack('<BBBBI', 0x10 | 0, 0x08 if kind == 'i' else 0, 0, 0, arr.dtype.itemsize) + struct.pack('<HH', 0, 8 * arr.dtype.itemsize)
    else:
        assert kind == 'f' and arr.dtype.itemsize in (4, 8)
        sz = arr.dtype.itemsize
        props = struct.pack('<HHBBBBI', 0, 8 * sz, 23 if sz == 4 else 52, 8 if sz == 4 else 11, 0, 23 if sz == 4 else 52, 127 if sz == 4 else 1023)
        dt = struct.pack('<BBBBI', 0x10 | 1, 0x20, 8 * sz - 1, 0, sz) + props
    maxdims = (UNDEF,) + arr.shape[1:] if chunks is not None else arr.shape
    msgs = [_msg(0x0001, struct.pack('<BBBBI', 1, rank, 1, 0, 0) + struct.pack(f'<{rank}Q', *arr.shape) + struct.pack(f'<{rank}Q', *maxdims)),
            _msg(0x0003, dt),
            _msg(0x0005, struct.pack('<BBBB', 2, 2, 2, 0)),        # fill value v2: allocate late, write at allocation, undefined -> default 0
            _msg(0x0008, layout)]
    if filters:
        msgs.append(_msg(0x000B, struct.pack('<BBHI', 1, len(filters), 0, 0) + b''.join(filters)))
    # a second header block reached through a continuation message, as libhdf5 produces once attributes are added
    tail = _msg(0x0012, struct.pack('<BBBBI', 1, 0, 0, 0, 0x5F000000)) + _msg(0x0000, b'\x00' * 8)
    tail_addr = alloc(tail)
    msgs.append(_msg(0x0010, struct.pack('<QQ', tail_addr, len(tail))))
    body = b''.join(msgs)
    ohdr = alloc(struct.pack('<BBHII', 1, 0, len(msgs) + 2, 1, len(body)) + b'\x00' * 4 + body)

    # ---- root group: local heap, symbol-table node, B-tree, object header ----
    heap_data = _pad8(b'\x00' * 8 + name.encode() + b'\x00')
    heap_data += b'\x00' * (88 - len(heap_data)) if len(heap_data) < 88 else b''
    seg = alloc(heap_data)
    heap = alloc(b'HEAP' + struct.pack('<BBBBQQQ', 0, 0, 0, 0, len(heap_data), UNDEF, seg))
    snod = alloc(b'SNOD' + struct.pack('<BBH', 1, 0, 1) + struct.pack('<QQII', 8, ohdr, 0, 0) + b'\x00' * 16 + b'\x00' * (40 * 7))
    gtree = alloc(b'TREE' + struct.pack('<BBHQQ', 0, 0, 1, UNDEF, UNDEF) + struct.pack('<QQQ', 0, snod, 8) + b'\x00' * (16 * 31))
    root = alloc(struct.pack('<BBHII', 1, 0, 1, 1, 24) + b'\x00' * 4 + _msg(0x0011, struct.pack('<QQ', gtree, heap)))
    sb = b'\x89HDF\r\n\x1a\n' + struct.pack('<BBBBBBBB', 0, 0, 0, 0, 0, 8, 8, 0) + struct.pack('<HHI', 4, 16, 0)
    sb += struct.pack('<QQQQ', 0, UNDEF, len(buf), UNDEF)
    sb += struct.pack('<QQII', 0, root, 1, 0) + struct.pack('<QQ', gtree, heap)
    assert len(sb) == 96
    buf[:96] = sb
    with open(fn, 'wb') as f:
        f.write(bytes(buf))
    return fn
