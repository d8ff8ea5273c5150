"""CPU-side checks of the C-ABI boundary: the shared library loads without a GPU and exports every
symbol that include/leod_hip.h declares (no compute calls here)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


@pytest.fixture(scope='module')
def built():
    import __graft_entry__ as g
    g.build()
    from leod_amd import _lib
    return _lib


def test_header_prototypes_parse(built):
    protos = built.parse_header()
    assert len(protos) >= 28
    for must in ('leod_ln_linear_fwd', 'leod_partition_attn_fwd', 'leod_partition_attn_bwd', 'leod_convlstm_fwd',
                 'leod_stem_conv_fwd', 'leod_conv_nhwc_fwd', 'leod_simota_assign', 'leod_yolox_loss',
                 'leod_postprocess_nms', 'leod_pseudo_filter', 'leod_adamw_clip_step', 'leod_voxelize_u8', 'leod_mixed_density_i8'):
        assert must in protos


def test_library_exports_every_declared_symbol(built):
    dll = built.lib()
    out = subprocess.check_output(['nm', '-D', '--defined-only', built.LIB_PATH], text=True)
    exported = set(re.findall(r'\bT (leod_\w+)', out))
    declared = set(built.parse_header())
    assert declared <= exported, declared - exported
    assert exported <= declared, f'exported but undeclared: {exported - declared}'
    assert dll.leod_version().decode().startswith('leod_hip')


def test_ops_refuse_cpu_tensors(built):
    import torch
    from leod_amd import ops
    from leod_amd._lib import LeodHipError
    with pytest.raises(LeodHipError):
        ops.layernorm_fwd(torch.zeros(4, 16), torch.ones(16), torch.zeros(16))


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'leod_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), f'{f} imports the oracle'


def test_typed_pointer_parameters_refuse_a_swapped_tensor(built):
    """the binding used to map every pointer to c_void_p, so a swapped argument was undetectable at the boundary.
    Pointer parameters now carry their declared element type (``_lib._pointer_type``) and ``ops._p`` hands over (address, dtype)."""
    import ctypes
    protos = built.parse_header()
    names = [a.__name__ for a in protos['leod_simota_assign'][1]]
    assert names[:6] == ['Ptr_float', 'Ptr_float', 'Ptr_float', 'Ptr_unsigned_char', 'Ptr_unsigned_char', 'Ptr_int']
    assert [a.__name__ for a in protos['leod_bn_silu_fwd'][1]][1] == 'Ptr_double'
    f32 = protos['leod_layernorm_fwd'][1][0]
    assert isinstance(f32.from_param(built.DevPtr(64, 'f32')), ctypes.c_void_p)
    assert isinstance(f32.from_param(built.DevPtr(64, 'bf16')), ctypes.c_void_p)      # 16-bit rows of precision mode bf16 travel as float*
    for wrong in ('i32', 'f64', 'u8', 'i64'):
        with pytest.raises(built.LeodHipError):
            f32.from_param(built.DevPtr(64, wrong))
    assert f32.from_param(None) is None or f32.from_param(None) is not None            # NULL stays legal (optional arguments)
