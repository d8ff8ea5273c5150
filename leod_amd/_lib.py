"""ctypes binding of libleod_hip.so (C ABI: include/leod_hip.h).

This is the stub a maintainer of the reference would add: the reference (100 % Python on ATen) has
no FFI, so the boundary is defined here -- plain C symbols taking raw device pointers, sizes and a
HIP stream.  Prototypes are parsed from the header so the two cannot drift.

The product path has NO fallback: if the shared library is missing or a symbol fails to resolve,
``lib()`` raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` (hipcc,
--offload-arch=gfx950; works without a GPU).
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libleod_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'leod_hip.h')

class LeodHipError(RuntimeError):
    pass


_CTYPE = {'int': ctypes.c_int, 'long': ctypes.c_long, 'float': ctypes.c_float, 'double': ctypes.c_double,
          'leod_stream_t': ctypes.c_void_p}


class DevPtr:
    """A device address that remembers the element type of the tensor it came from (``ops._p``): the typed pointer parameters below
    check it, so that a swapped argument (an int32 index buffer where float rows are expected, a float64 statistics block for a
    float one) is refused at the boundary instead of being read as garbage on the device."""
    __slots__ = ('addr', 'kind')

    def __init__(self, addr: int, kind: str):
        self.addr, self.kind = addr, kind


# element kinds a declared C pointer type accepts.  ``float*`` parameters also carry the 16-bit rows of precision mode bf16 (the
# header documents which, e.g. dy_bf16 / qkv_bf16 flags); ``void*`` takes anything.
_ACCEPTS = {'float': ('f32', 'bf16', 'f16'), 'double': ('f64',), 'int': ('i32',), 'long': ('i64',),
            'unsigned char': ('u8', 'bool'), 'signed char': ('i8',), 'void': None, 'char': None}


def _pointer_type(elem: str):
    accepts = _ACCEPTS.get(elem)

    class _Ptr(ctypes.c_void_p):
        @classmethod
        def from_param(cls, v):
            if isinstance(v, DevPtr):
                if accepts is not None and v.kind not in accepts:
                    raise LeodHipError(f'pointer to {v.kind} elements passed where the C ABI declares {elem}*')
                return ctypes.c_void_p(v.addr)
            return ctypes.c_void_p.from_param(v)                 # None, raw addresses (int), ctypes arrays / pointers
    _Ptr.__name__ = 'Ptr_' + elem.replace(' ', '_')
    return _Ptr


_PTR = {}


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every ``leod_*`` prototype in the header."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    protos = {}
    for m in re.finditer(r'\b(int|long|const char\s*\*)\s+(leod_\w+)\s*\(([^)]*)\)\s*;', src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a:
                    # declared element type: `const float* x`, `unsigned char* mask`, `void* const* tensors` (pointer tables: void)
                    elem = re.sub(r'\bconst\b', ' ', a.split('*')[0]).strip()
                    elem = elem if a.count('*') == 1 and elem in _ACCEPTS else 'void'
                    if elem not in _PTR:
                        _PTR[elem] = _pointer_type(elem)
                    argtypes.append(_PTR[elem])
                else:
                    ty = a.replace('const', '').split()[0]
                    argtypes.append(_CTYPE[ty])
        restype = ctypes.c_char_p if '*' in ret else _CTYPE[ret.strip()]
        protos[name] = (restype, argtypes)
    return protos


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise LeodHipError(
                f'{LIB_PATH} not found: the LEOD MI355X path needs its HIP kernel library and has no fallback. '
                f'Build it with: python -c "import __graft_entry__ as g; g.build()"')
        dll = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in parse_header().items():
            try:
                fn = getattr(dll, name)
            except AttributeError as e:
                raise LeodHipError(f'libleod_hip.so does not export {name} declared in include/leod_hip.h') from e
            fn.restype = restype
            fn.argtypes = argtypes
        _LIB = dll
    return _LIB


_ERR = {-1: 'bad argument', -2: 'kernel launch failed', -3: 'unsupported shape'}


def check(rc, what):
    if rc != 0:
        raise LeodHipError(f'{what} failed: {_ERR.get(rc, rc)}')
