"""Thin tensor-level wrappers over the C ABI (include/leod_hip.h).

PyTorch is used here only as plumbing: device memory (``torch.empty``), the current HIP stream and
shape bookkeeping.  All arithmetic happens in libleod_hip.so; there is no fallback path -- every
function raises if its input is not a contiguous fp32 CUDA(HIP) tensor or the library is missing.

Convention: activations are channels-last.  2-D ``[M, C]`` "rows" and 4-D ``[B, H, W, C]`` maps are the
same memory.
"""
from typing import Optional, Sequence, Tuple

import ctypes

import torch

from ._lib import lib, check, LeodHipError, DevPtr

F32 = torch.float32


_LIB = None


def _l():
    """cached CDLL handle (attribute lookups on a CDLL are cached by ctypes after the first use)"""
    global _LIB
    if _LIB is None:
        _LIB = lib()
    return _LIB


try:
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:                                        # pragma: no cover
    _raw_stream = None


PRECISIONS = {'f32': 0, 'fp32': 0, 'float32': 0, 32: 0, '32': 0, '32-true': 0,
              'bf16': 1, 'bfloat16': 1, 'bf16-mixed': 1,
              '16f': 2, 'f16': 2, 'fp16': 2, 'float16': 2, 16: 2, '16': 2, '16-mixed': 2}
PRECISION_NAMES = {0: 'f32', 1: 'bf16', 2: '16f'}


def set_precision(mode) -> int:
    """'f32' (default): fp32 end to end, bit-tight against the fp32 oracle.
    '16f' (the reference's ``precision=16`` = fp16 autocast, train.py:236-243): the FORWARD contractions take fp16 operands (fp16 MFMA, fp32
    accumulation) and the 16-bit activations the forward pass leaves in HBM (qkv, attention output, MLP hidden) are fp16; the gradient
    contractions take bf16 operands (fp32's exponent range, so no loss scaler); statistics, softmax, residual stream, LSTM state,
    SimOTA cost, losses and optimiser in fp32.
    'bf16' (Lightning's ``bf16-mixed``): bf16 operands in both directions.
    Process-wide; returns the previous mode (0 / 1 / 2)."""
    prev = int(_l().leod_get_precision())
    check(_l().leod_set_precision(PRECISIONS[mode] if not isinstance(mode, bool) and mode in PRECISIONS else int(mode)), 'set_precision')
    return prev


def precision_from_config(training_cfg=None) -> str:
    """The mode a run asks for: ``LEOD_PRECISION`` (f32 | bf16 | 16f) if set, else the reference's ``training.precision`` key
    (config/general.yaml: 16 -> fp16 mixed precision = mode 16f; 'bf16' / 'bf16-mixed' -> bf16; 32 -> fp32)."""
    import os
    env = os.environ.get('LEOD_PRECISION')
    if env:
        return PRECISION_NAMES[PRECISIONS[env]]
    prec = None if training_cfg is None else training_cfg.get('precision', 32)
    return PRECISION_NAMES[PRECISIONS.get(str(prec).lower(), 0)]


def get_precision() -> str:
    return PRECISION_NAMES[int(_l().leod_get_precision())]


def is_16bit() -> bool:
    """One of the two mixed-precision modes ('bf16', '16f'): the 16-bit tensor layouts and kernel families are in use."""
    return int(_l().leod_get_precision()) != 0


def act16_dtype():
    """torch dtype of the 16-bit activation rows the forward pass stores (qkv, attention output): fp16 in mode 16f, else bf16.  Gradient
    rows (dqkv, du, dO, LSTM gate gradients) are bf16 in both modes; the MLP hidden pre-activation and the LSTM gates fp16 in both."""
    return torch.float16 if int(_l().leod_get_precision()) == 2 else torch.bfloat16


def _is16(t) -> bool:
    return t.dtype is torch.bfloat16 or t.dtype is torch.float16


def _stream():
    """raw hipStream_t of torch's current stream (the stream every kernel of this library is enqueued on)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


_KIND = {torch.float32: 'f32', torch.bfloat16: 'bf16', torch.float16: 'f16', torch.float64: 'f64', torch.int32: 'i32', torch.int64: 'i64',
         torch.uint8: 'u8', torch.bool: 'bool', torch.int8: 'i8'}


def _p(t: Optional[torch.Tensor]):
    """device address + element kind: the typed pointer parameters of the binding (``_lib.DevPtr``) refuse a mismatching tensor"""
    return None if t is None else DevPtr(t.data_ptr(), _KIND.get(t.dtype, 'void'))


def _ck(t: Optional[torch.Tensor], dtype=F32, name='tensor'):
    """Loud input validation: device tensor, expected dtype, contiguous.  No fallback of any kind."""
    if t is None:
        return
    if t.dtype is not dtype or not t.is_cuda or not t.is_contiguous():
        if not t.is_cuda:
            raise LeodHipError(f'{name}: the LEOD HIP path needs device tensors (got {t.device}); there is no CPU fallback')
        if t.dtype != dtype:
            raise LeodHipError(f'{name}: expected {dtype}, got {t.dtype}')
        raise LeodHipError(f'{name}: expected a contiguous tensor, got strides {t.stride()}')


def _ck_dtype_dev(t: torch.Tensor, dtype=F32, name='tensor'):
    """``_ck`` without the contiguity requirement (for operands a kernel reads with an explicit row stride)."""
    if not t.is_cuda:
        raise LeodHipError(f'{name}: the LEOD HIP path needs device tensors (got {t.device}); there is no CPU fallback')
    if t.dtype != dtype:
        raise LeodHipError(f'{name}: expected {dtype}, got {t.dtype}')


def _empty(shape, like: torch.Tensor, dtype=F32):
    return torch.empty(shape, dtype=dtype, device=like.device)


# ---------------------------------------------------------------------------------------------------
# backbone pieces
# ---------------------------------------------------------------------------------------------------
def ln_linear_fwd(x, ln_w, ln_b, W, bias, want_act=False, want_stats=False, eps=1e-5, out_bf16=False):
    """x [.., K] -> out [.., N] = LN(x) W^T + b ; optional gelu(out), LN stats [M,2].
    With want_act in precision mode bf16 and a shape the row-streaming kernel covers (RVT stages 1-2) the result is
    (u as torch.float16, None, stats): pass that u to linear_lsres_fwd / linear_dgrad(aux_u=) / linear_wgrad(x=), which apply GELU."""
    for t, n in ((x, 'x'), (ln_w, 'ln_w'), (ln_b, 'ln_b'), (W, 'W'), (bias, 'bias')):
        _ck(t, name=n)
    K = x.shape[-1]
    N = W.shape[0]
    M = x.numel() // K
    ev = _probe('linear_gemm', 0, 0) if _PROBE is not None else None
    try:
        return _ln_linear_fwd(x, ln_w, ln_b, W, bias, want_act, want_stats, eps, out_bf16, K, N, M, ev)
    finally:
        if ev is not None:
            ev.record()


def _ln_linear_fwd(x, ln_w, ln_b, W, bias, want_act, want_stats, eps, out_bf16, K, N, M, ev):
    def tally(out_bytes_per_el, n_out=1):
        # algorithmic bytes: x once, W once, the output(s) at their stored width; 2 M N K flops
        if ev is not None:
            _PROBE.amend('linear_gemm', 4.0 * M * K + 4.0 * N * K + out_bytes_per_el * n_out * M * N, 2.0 * M * N * K,
                         2.0 * M * K + 4.0 * N * K + 2.0 * n_out * M * N)
    if out_bf16 and not want_act:
        # the qkv rows in precision mode bf16 (consumed only by bf16 MFMAs): stored as bf16 where the kernels allow (rc -3: they do not)
        o16 = torch.empty(x.shape[:-1] + (N,), dtype=act16_dtype(), device=x.device)
        stats = _empty((M, 2), x) if ln_w is not None else None
        rc = _l().leod_ln_linear_bf16_fwd(_p(x), _p(ln_w), _p(ln_b), eps, _p(W), _p(bias), _p(o16), _p(stats), M, N, K, _stream())
        if rc != -3:
            check(rc, 'ln_linear_bf16_fwd')
            tally(2.0)
            return o16, None, stats
    if want_act and want_stats and ln_w is not None and is_16bit():
        # precision mode bf16: the hidden pre-activation is stored once, as fp16 (the reference's autocast dtype); consumers apply GELU on load
        u16 = torch.empty(x.shape[:-1] + (N,), dtype=torch.float16, device=x.device)
        stats = _empty((M, 2), x)
        rc = _l().leod_ln_linear_gelu16_fwd(_p(x), _p(ln_w), _p(ln_b), eps, _p(W), _p(bias), _p(u16), _p(stats), M, N, K, _stream())
        if rc != -3:
            check(rc, 'ln_linear_gelu16_fwd')
            tally(2.0)
            return u16, None, stats
    out = _empty(x.shape[:-1] + (N,), x)
    act = _empty(out.shape, x) if want_act else None
    # always hand the kernel a statistics buffer: the LDS-staged GEMM reads precomputed (mean, rstd) from it
    stats = _empty((M, 2), x) if ln_w is not None else None
    check(_l().leod_ln_linear_fwd(_p(x), K, _p(ln_w), _p(ln_b), eps, _p(W), _p(bias), _p(out), _p(act), _p(stats),
                                   M, N, K, _stream()), 'ln_linear_fwd')
    tally(4.0, 2 if want_act else 1)
    return out, act, stats


def linear_lsres_fwd(a, W, bias, gamma, res, want_t=True, a_gelu=None):
    """out = res + gamma * (a W^T + b); also returns t = a W^T + b when want_t.
    a_gelu: ``a`` is the fp16 PRE-activation of the MLP hidden (the kernel applies GELU on load).  None: decided by the dtype as before
    mode 16f existed -- torch.float16 means pre-activation; callers that hand over plain fp16 rows (the attention output of mode 16f)
    say a_gelu=False."""
    for t, n in ((W, 'W'), (bias, 'bias'), (gamma, 'gamma'), (res, 'res')):
        _ck(t, name=n)
    K = a.shape[-1]
    N = W.shape[0]
    M = a.numel() // K
    # algorithmic bytes: a once (stored width), W, residual in, output out
    ev = _probe('linear_gemm', a.element_size() * M * K + 4.0 * N * K + (12.0 if want_t else 8.0) * M * N, 2.0 * M * N * K)
    try:
        return _linear_lsres_fwd(a, W, bias, gamma, res, want_t, K, N, M, a_gelu)
    finally:
        if ev is not None:
            ev.record()


def _linear_lsres_fwd(a, W, bias, gamma, res, want_t, K, N, M, a_gelu=None):
    if a.dtype is torch.float16 and a_gelu is not False:     # a = fp16 pre-activation: out = res + gamma * (gelu(a) W^T + b)
        _ck(a, torch.float16, 'a')
        if want_t:
            raise LeodHipError('linear_lsres_fwd: the fp16 pre-activation path does not return t')
        out = _empty(res.shape, res)
        check(_l().leod_linear_lsres_gelu16_fwd(_p(a), _p(W), _p(bias), _p(gamma), _p(res), _p(out), M, N, K, _stream()),
              'linear_lsres_gelu16_fwd')
        return out, None
    if _is16(a):                                             # a = 16-bit rows (the attention output: bf16, or fp16 in mode 16f)
        _ck(a, act16_dtype(), 'a')
        if want_t:
            raise LeodHipError('linear_lsres_fwd: the 16-bit-row path does not return t')
        out = _empty(res.shape, res)
        check(_l().leod_linear_lsres_bf16_fwd(_p(a), _p(W), _p(bias), _p(gamma), _p(res), _p(out), M, N, K, _stream()),
              'linear_lsres_bf16_fwd')
        return out, None
    _ck(a, name='a')
    out = _empty(res.shape, a)
    tout = _empty(res.shape, a) if want_t else None
    check(_l().leod_linear_lsres_fwd(_p(a), _p(W), _p(bias), _p(gamma), _p(res), _p(out), _p(tout), M, N, K,
                                      _stream()), 'linear_lsres_fwd')
    return out, tout


def mlp_fwd_fused(y, ln_w, ln_b, W1, b1, W2, b2, gamma, want_saved=False, eps=1e-5):
    """z = y + gamma * (gelu(LN(y) W1^T + b1) W2^T + b2) in ONE launch (csrc/k_mlp.hip) -> (z, u16 | None, stats | None), or None where
    the fused kernel does not cover the shape / precision mode (the caller then runs ln_linear_fwd + linear_lsres_fwd)."""
    for t, n in ((y, 'y'), (ln_w, 'ln_w'), (ln_b, 'ln_b'), (W1, 'W1'), (b1, 'b1'), (W2, 'W2'), (b2, 'b2'), (gamma, 'gamma')):
        _ck(t, name=n)
    K = y.shape[-1]
    H = W1.shape[0]
    M = y.numel() // K
    out = _empty(y.shape, y)
    u16 = torch.empty(y.shape[:-1] + (H,), dtype=torch.float16, device=y.device) if want_saved else None
    stats = _empty((M, 2), y) if want_saved else None
    ev = _probe('linear_gemm', 8.0 * M * K + 4.0 * 2 * H * K + (2.0 * M * H if want_saved else 0.0), 4.0 * M * H * K)
    rc = _l().leod_mlp_fwd_fused(_p(y), _p(ln_w), _p(ln_b), eps, _p(W1), _p(b1), _p(W2), _p(b2), _p(gamma), _p(out), _p(u16), _p(stats),
                                 M, H, K, _stream())
    if ev is not None:
        ev.record()
    if rc == -3:
        return None
    check(rc, 'mlp_fwd_fused')
    return out, u16, stats


def mlp_bwd_dgrad_fused(dz, y, stats, ln_w, ln_b, W1, b1, W2, gamma, dgamma, dbeta, want_du=True):
    """The activation-path backward of the fused MLP in ONE launch (csrc/k_mlp.hip) -> (dy, du bf16 | None), or None where it does not
    apply (the caller then runs linear_dgrad(aux_u=) + linear_dgrad_ln_bwd).  dgamma / dbeta accumulate norm2's gradients."""
    for t, n in ((dz, 'dz'), (y, 'y'), (stats, 'stats'), (ln_w, 'ln_w'), (ln_b, 'ln_b'), (W1, 'W1'), (b1, 'b1'), (W2, 'W2'),
                 (gamma, 'gamma'), (dgamma, 'dgamma'), (dbeta, 'dbeta')):
        _ck(t, name=n)
    K = y.shape[-1]
    H = W1.shape[0]
    M = y.numel() // K
    if not BF16_GRADS:
        return None
    dy = _empty(y.shape, y)
    du = torch.empty(y.shape[:-1] + (H,), dtype=torch.bfloat16, device=y.device) if want_du else None
    ev = _probe('linear_gemm', 12.0 * M * K + 4.0 * 2 * H * K + (2.0 * M * H if want_du else 0.0), 8.0 * M * H * K)
    rc = _l().leod_mlp_bwd_dgrad_fused(_p(dz), _p(y), _p(stats), _p(ln_w), _p(ln_b), _p(W1), _p(b1), _p(W2), _p(gamma), _p(dy), _p(du),
                                       _p(dgamma), _p(dbeta), M, H, K, _stream())
    if ev is not None:
        ev.record()
    if rc == -3:
        return None
    check(rc, 'mlp_bwd_dgrad_fused')
    return dy, du


def partition_attn_fwd(qkv, heads, part, window, want_lse=False, out_bf16=False):
    """qkv [B,H,W,3C] -> out [B,H,W,C] (+ lse [B,H,W,heads]); out_bf16 (attn_block_o16_ok): out as bf16 rows."""
    q16 = _is16(qkv)
    _ck(qkv, act16_dtype() if q16 else F32, 'qkv')
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    out = torch.empty((B, H, W, C), dtype=act16_dtype() if out_bf16 else F32, device=qkv.device)
    lse = torch.empty((B, H, W, heads), dtype=F32, device=qkv.device) if want_lse else None
    check(_l().leod_partition_attn_fwd(_p(qkv), _p(out), _p(lse), B, H, W, C, heads, part[0], part[1],
                                        1 if window else 0, (1 if q16 else 0) | (2 if out_bf16 else 0), _stream()), 'partition_attn_fwd')
    return out, lse


def attn_block_o16_ok(B, H, W, C, heads, part) -> bool:
    """The attention output O and its gradient dO of this block geometry may live in HBM as bf16 rows (every consumer has a 16-bit path)."""
    return BF16_GRADS and bool(_l().leod_attn_block_o16_ok(B, H, W, C, heads, part[0], part[1]))


def partition_attn_16bit_ok(B, H, W, C, heads, part) -> bool:
    """qkv may be handed to partition_attn_fwd / _bwd as torch.bfloat16 (and dqkv comes back as bfloat16) for this geometry."""
    return BF16_GRADS and bool(_l().leod_partition_attn_16bit_ok(B, H, W, C, heads, part[0], part[1]))


def partition_attn_bwd(qkv, dout, lse, heads, part, window):
    q16 = _is16(qkv)
    _ck(qkv, act16_dtype() if q16 else F32, 'qkv')
    do16 = dout.dtype is torch.bfloat16                       # bf16 dO rows (attn_block_o16_ok)
    _ck(dout, torch.bfloat16 if do16 else F32, 'dout')
    _ck(lse, name='lse')
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    dqkv = torch.empty(qkv.shape, dtype=torch.bfloat16 if q16 else F32, device=qkv.device)     # 16-bit qkv <=> bf16 dqkv (partition_attn_16bit_ok)
    dsum = torch.empty(lse.shape, dtype=F32, device=qkv.device)
    check(_l().leod_partition_attn_bwd(_p(qkv), _p(dout), _p(lse), _p(dsum), _p(dqkv), B, H, W, C, heads, part[0],
                                        part[1], 1 if window else 0, (1 if q16 else 0) | (2 if do16 else 0), 1 if q16 else 0, _stream()), 'partition_attn_bwd')
    return dqkv


def convlstm_fwd(x, h_prev, c_prev, W, bias, want_gates=False, h_out=None, c_out=None, gates_out=None):
    """x/h_prev/c_prev [..,C] channels-last rows; W [4C,2C]; -> h, c, gates[M,4,C] (optionally into given buffers)."""
    for t, n in ((x, 'x'), (h_prev, 'h_prev'), (c_prev, 'c_prev'), (W, 'W'), (bias, 'bias'), (h_out, 'h_out'),
                 (c_out, 'c_out'), (gates_out, 'gates_out')):
        _ck(t, name=n)
    C = x.shape[-1]
    M = x.numel() // C
    h = _empty(x.shape, x) if h_out is None else h_out
    c = _empty(x.shape, x) if c_out is None else c_out
    gates = gates_out if gates_out is not None else (_empty((M, 4, C), x) if want_gates else None)
    check(_l().leod_convlstm_fwd(_p(x), _p(h_prev), _p(c_prev), _p(W), _p(bias), _p(h), _p(c), _p(gates), M, C,
                                  _stream()), 'convlstm_fwd')
    return h, c, gates


def convlstm_gates_bwd(dh, dc_next, gates, c_prev, c_t, want_dc_prev=True, dh2=None, dgates_out=None):
    for t, n in ((dh, 'dh'), (dc_next, 'dc_next'), (gates, 'gates'), (c_prev, 'c_prev'), (c_t, 'c_t'), (dh2, 'dh2'),
                 (dgates_out, 'dgates_out')):
        _ck(t, name=n)
    M, _, C = gates.shape
    dgates = _empty((M, 4 * C), gates) if dgates_out is None else dgates_out
    dc_prev = _empty(c_t.shape, gates) if want_dc_prev else None
    check(_l().leod_convlstm_gates_bwd(_p(dh), _p(dh2), _p(dc_next), _p(gates), _p(c_prev), _p(c_t), _p(dgates), _p(dc_prev),
                                        M, C, _stream()), 'convlstm_gates_bwd')
    return dgates, dc_prev


def convlstm_seq_mode(C: int) -> int:
    """0: no sequence kernel for this channel count / precision mode; 1: fused [x | h] contraction (xin = x_seq);
    2: the caller supplies the time-batched projection gx = x W_x^T + b (leod_convlstm_seq_mode)."""
    return int(_l().leod_convlstm_seq_mode(int(C)))


def convlstm_seq_pack(W, C: int):
    """Mode 3 only: the fragment-ordered bf16 copy of W_h the sequence kernels stream (valid until the weights change) | None."""
    n = int(_l().leod_convlstm_seq_pack_bytes(int(C)))
    if n == 0:
        return None
    _ck(W, name='W')
    # a persistent buffer of the pack cache (never a temporary of the calling step: the launch plans run weight packs ahead of the captured
    # order, which is only sound for buffers no other kernel of the step ever writes), re-packed once per optimiser step
    buf, valid = PackCache.get(W, ('lstm', int(C)), (n + 3) // 4)
    wp = buf.view(torch.uint8)[:n]
    if not valid:
        check(_l().leod_convlstm_seq_pack(_p(W), _p(wp), int(C), _stream()), 'convlstm_seq_pack')
    return wp


def convlstm_gates16_ok(C: int) -> bool:
    """Precision mode bf16: the sequence kernels keep the gates as fp16 (their own layout) and emit bf16 gate-gradient rows."""
    return bool(_l().leod_convlstm_seq_gates16_ok(int(C)))


def convlstm_gates16_buffer(T: int, M: int, C: int, device):
    """fp16 gate storage for convlstm_seq_fwd / _bwd (opaque layout: only those two calls read it)."""
    return torch.empty((T, (M + 15) // 16 * 16, 4, C), dtype=torch.float16, device=device)


def convlstm_seq_fwd(xin, is_projection, hbuf, cbuf, W, bias, gates_out, zero_state, wpack=None):
    """The whole recurrence in one launch: xin [T,M,C] (or gx [T,M,4C]), hbuf / cbuf [T+1,M,C] (slot 0 = incoming state,
    slots 1.. written), W [4C,2C], gates_out [T,M,4,C] fp32 | convlstm_gates16_buffer(...) | None."""
    g16 = gates_out is not None and gates_out.dtype is torch.float16
    for t, n in ((xin, 'xin'), (hbuf, 'hbuf'), (cbuf, 'cbuf'), (W, 'W'), (bias, 'bias')):
        _ck(t, name=n)
    _ck(gates_out, torch.float16 if g16 else F32, 'gates_out')
    T = hbuf.shape[0] - 1
    C = hbuf.shape[-1]
    M = hbuf[0].numel() // C
    check(_l().leod_convlstm_seq_fwd(_p(xin), 1 if is_projection else 0, _p(hbuf), _p(cbuf), _p(W), _p(bias), _p(gates_out), _p(wpack),
                                      M, C, T, 1 if zero_state else 0, 1 if g16 else 0, _stream()), 'convlstm_seq_fwd')


def convlstm_seq_bwd(dh_seq, dc_last, gates, cbuf, W, dgates_out, dh0=None, dc0=None, zero_state=False, wpack=None) -> bool:
    """Backward through time in one launch -> dgates_out [T,M,4C] (+ dh0, dc0).  False: the weight slice does not fit the registers
    for this C / precision mode (the caller then runs the per-timestep kernels on the same saved tensors)."""
    g16 = gates.dtype is torch.float16
    for t, n in ((dh_seq, 'dh_seq'), (dc_last, 'dc_last'), (cbuf, 'cbuf'), (W, 'W'), (dh0, 'dh0'), (dc0, 'dc0')):
        _ck(t, name=n)
    _ck(gates, torch.float16 if g16 else F32, 'gates')
    _ck(dgates_out, torch.bfloat16 if g16 else F32, 'dgates_out')
    T = cbuf.shape[0] - 1
    C = cbuf.shape[-1]
    M = cbuf[0].numel() // C
    rc = _l().leod_convlstm_seq_bwd(_p(dh_seq), _p(dc_last), _p(gates), _p(cbuf), _p(W), _p(dgates_out), _p(dh0), _p(dc0), _p(wpack),
                                    M, C, T, 1 if zero_state else 0, 1 if g16 else 0, _stream())
    if rc == -3:
        return False
    check(rc, 'convlstm_seq_bwd')
    return True


def linear_dgrad(dy, W, kscale=None, aux_u=None, colsum=None, out=None, accumulate=False, split=0, out2=None, dres=None, out_bf16=False):
    """dx = (dy * kscale) @ W  with W [N,K] (+ dres: the other gradient source of a residual branch); see leod_linear_dgrad."""
    dy16 = dy.dtype is torch.bfloat16                         # bf16 gradient rows (du / dqkv of precision mode bf16)
    _ck(dy, torch.bfloat16 if dy16 else F32, 'dy')
    for t, n in ((W, 'W'), (kscale, 'kscale'), (colsum, 'colsum'), (out, 'out'), (out2, 'out2'), (dres, 'dres')):
        _ck(t, name=n)
    N, K = W.shape[0], W.shape[1] if W.dim() == 2 else W.numel() // W.shape[0]
    M = dy.numel() // N
    aux_b = 0.0 if aux_u is None else aux_u.element_size() * M * K
    out_b = (2.0 if (out_bf16 or (aux_u is not None and aux_u.dtype is torch.float16 and BF16_GRADS)) else 4.0) * M * K
    ev = _probe('linear_gemm', dy.element_size() * M * N + 4.0 * N * K + aux_b + out_b + (4.0 * M * K if (accumulate or dres is not None) else 0.0),
                2.0 * M * N * K)
    try:
        return _linear_dgrad(dy, W, kscale, aux_u, colsum, out, accumulate, split, out2, dres, dy16, N, K, M, out_bf16)
    finally:
        if ev is not None:
            ev.record()


def _linear_dgrad(dy, W, kscale, aux_u, colsum, out, accumulate, split, out2, dres, dy16, N, K, M, out_bf16=False):
    if aux_u is not None and aux_u.dtype is torch.float16:    # through GELU on the fp16 pre-activation (stages 1-2, bf16 mode)
        _ck(aux_u, torch.float16, 'aux_u')
        if split or colsum is not None or accumulate or out is not None or dres is not None:
            raise LeodHipError('linear_dgrad: unsupported option with an fp16 pre-activation')
        if dy16:
            raise LeodHipError('linear_dgrad: bf16 dy does not combine with an fp16 pre-activation')
        # the gradient of the hidden goes out as bf16: its two consumers (dgrad of fc1, fc1 weight gradient) feed bf16 MFMAs
        out = torch.empty(dy.shape[:-1] + (K,), dtype=torch.bfloat16 if BF16_GRADS else F32, device=dy.device)
        check(_l().leod_linear_dgrad_gelu16(_p(dy), _p(kscale), _p(W), _p(aux_u), _p(out), M, N, K, 1 if BF16_GRADS else 0, _stream()),
              'linear_dgrad_gelu16')
        return out
    _ck(aux_u, name='aux_u')
    if split:
        if out is None:
            out = torch.empty(dy.shape[:-1] + (split,), dtype=F32, device=dy.device)
        if out2 is None:
            out2 = torch.empty(dy.shape[:-1] + (K - split,), dtype=F32, device=dy.device)
        ld1, ld2 = split, K - split
    else:
        if out is None:
            out = torch.empty(dy.shape[:-1] + (K,), dtype=torch.bfloat16 if out_bf16 else F32, device=dy.device)
        ld1, ld2 = K, 0
    check(_l().leod_linear_dgrad(_p(dy), N, _p(kscale), _p(W), _p(out), ld1, _p(out2), ld2, split, _p(aux_u),
                                  _p(colsum), 1 if accumulate else 0, _p(dres), M, N, K, (1 if dy16 else 0) | (2 if out_bf16 else 0), _stream()),
          'linear_dgrad')
    return (out, out2) if split else out


_WORKSPACES = {}     # raw stream -> uint8 tensor registered as that stream's weight-gradient workspace (kept alive here)


def _wgrad_workspace(device):
    """The partial-tile scratch of the wide weight-gradient kernel (wgrad_bf16.hpp), one per stream, from torch's allocator -- which also
    serves a stream under graph capture, where the library could not allocate for itself."""
    s = _stream()
    if s not in _WORKSPACES:
        ws = torch.empty(int(_l().leod_workspace_bytes()), dtype=torch.uint8, device=device)
        check(_l().leod_set_workspace(ws.data_ptr(), ws.numel(), s), 'set_workspace')
        _WORKSPACES[s] = ws


def linear_wgrad(dy, x, dW, dbias=None, stats=None, ln_w=None, ln_b=None, x2=None, x_gelu=None):
    """dW += dy^T X ; dbias += colsum(dy).  X = x | LN(x) | [x|x2] | gelu(x) (x the fp16 pre-activation; x_gelu as in linear_lsres_fwd)."""
    dy16 = dy.dtype is torch.bfloat16
    _ck(dy, torch.bfloat16 if dy16 else F32, 'dy')
    for t, n in ((dW, 'dW'), (dbias, 'dbias'), (stats, 'stats'), (ln_w, 'ln_w'), (ln_b, 'ln_b'), (x2, 'x2')):
        _ck(t, name=n)
    N = dW.shape[0]
    K = dW.numel() // N
    M = dy.numel() // N
    K1 = x.shape[-1]
    if M >= 8192 and _stream() not in _WORKSPACES and is_16bit():
        _wgrad_workspace(dW.device)
    if x.dtype is torch.float16 and x_gelu is not False:     # X = gelu(x): x is the fp16 pre-activation of the MLP hidden
        _ck(x, torch.float16, 'x')
        if stats is not None or x2 is not None or dy16:
            raise LeodHipError('linear_wgrad: LayerNorm / concat / bf16-dy options do not combine with an fp16 pre-activation')
        ev = _probe('linear_wgrad', 4.0 * (M * N + N * K) + 2.0 * M * K, 2.0 * M * N * K, rows=M, nbytes16=2.0 * M * (N + K) + 4.0 * N * K)
        check(_l().leod_linear_wgrad_gelu16(_p(dy), N, _p(x), _p(dW), _p(dbias), M, N, K, _stream()), 'linear_wgrad_gelu16')
        if ev is not None:
            ev.record()
        return
    x16 = _is16(x)                                            # 16-bit rows (the attention output: bf16, or fp16 in mode 16f)
    xh = x.dtype is torch.float16
    _ck(x, x.dtype if x16 else F32, 'x')
    # algorithmic work of one launch: reads dy, X once, read-modify-writes dW once; 2*M*N*K flops
    ev = _probe('linear_wgrad', (2.0 if dy16 else 4.0) * M * N + (2.0 if x16 else 4.0) * M * K + 4.0 * N * K, 2.0 * M * N * K, rows=M,
                nbytes16=2.0 * M * (N + K) + 4.0 * N * K)
    check(_l().leod_linear_wgrad(_p(dy), N, _p(x), K1, _p(stats), _p(ln_w), _p(ln_b), _p(x2),
                                  (x2.shape[-1] if x2 is not None else 0), K1, _p(dW), _p(dbias), M, N, K,
                                  (1 if dy16 else 0) | (4 if xh else (2 if x16 else 0)), _stream()), 'linear_wgrad')
    if ev is not None:
        ev.record()


def _x_fmt(x, stats, gelu):
    if gelu:
        return 2
    if x.dtype is torch.bfloat16:
        return 3
    if x.dtype is torch.float16:
        return 4
    return 1 if stats is not None else 0


def linear_wgrad_group(problems) -> bool:
    """n <= 4 Linear weight gradients of one row count in ONE preparation / contraction / reduce launch (``leod_linear_wgrad_group``).
    problems: dicts with dy, x, dW, dbias and optionally stats + ln_w + ln_b (X = LayerNorm(x)) or gelu=True (x the fp16 pre-activation).
    False: not coverable -- nothing was launched (the caller runs them singly)."""
    n = len(problems)
    if not (1 <= n <= 4) or not is_16bit():
        return False
    M = None
    Ns, Ks, dyf, xf = [], [], [], []
    nb = fl = nb16 = 0.0
    for p in problems:
        dy, x, dW = p['dy'], p['x'], p['dW']
        N = dW.shape[0]
        K = dW.numel() // N
        m = dy.numel() // N
        if M is None:
            M = m
        if m != M or x.numel() != m * K or x.shape[-1] != K or p.get('x2') is not None:
            return False
        dy16 = dy.dtype is torch.bfloat16
        _ck(dy, torch.bfloat16 if dy16 else F32, 'dy')
        _ck(x, x.dtype, 'x')
        fmt = _x_fmt(x, p.get('stats'), p.get('gelu'))
        if fmt == 2 and x.dtype is not torch.float16:
            return False
        for t_ in (dW, p.get('dbias'), p.get('stats'), p.get('ln_w'), p.get('ln_b')):
            _ck(t_, name='linear_wgrad_group')
        Ns.append(N); Ks.append(K); dyf.append(1 if dy16 else 0); xf.append(fmt)
        nb += (2.0 if dy16 else 4.0) * M * N + (2.0 if fmt >= 2 else 4.0) * M * K + 4.0 * N * K
        fl += 2.0 * M * N * K
        nb16 += 2.0 * M * (N + K) + 4.0 * N * K
    if M >= 8192 and _stream() not in _WORKSPACES:
        _wgrad_workspace(problems[0]['dW'].device)
    ev = _probe('linear_wgrad', nb, fl, rows=M, nbytes16=nb16)
    rc = _l().leod_linear_wgrad_group(n, _ptr_array([p['dy'] for p in problems]), _int_array(dyf), _ptr_array([p['x'] for p in problems]), _int_array(xf),
                                      _ptr_array_opt([p.get('stats') for p in problems]), _ptr_array_opt([p.get('ln_w') for p in problems]),
                                      _ptr_array_opt([p.get('ln_b') for p in problems]), _ptr_array([p['dW'] for p in problems]),
                                      _ptr_array_opt([p.get('dbias') for p in problems]), M, _int_array(Ns), _int_array(Ks), _stream())
    if rc == -3:
        if ev is not None:
            _PROBE.cancel('linear_wgrad')
        return False
    check(rc, 'linear_wgrad_group')
    if ev is not None:
        ev.record()
    return True


def attn_block_wgrads(dz, h, h_gelu, du, y, st2, n2w, n2b, dy, o, dqkv, x, st1, n1w, n1b, fc2, fc1, proj, qkv) -> bool:
    """The four weight gradients of an attention block (fc2 and proj behind LayerScale, fc1 and qkv behind LayerNorm) in one grouped launch.
    fc2 / proj = (W, b, gamma, dW, db, dgamma); fc1 / qkv = (dW, db).  False: not coverable, nothing was launched."""
    W2, b2, g2, dW2, db2, dg2 = fc2
    Wp, bp, g1, dWp, dbp, dg1 = proj
    (N2, K2), (Np, Kp) = W2.shape, Wp.shape
    if not is_16bit() or dz.numel() // N2 < 8192:
        return False
    scratch = StatArena.zeros((N2 * K2 + N2 + Np * Kp + Np,), dz.device, torch.float32)
    G2, s2 = scratch[:N2 * K2].view(N2, K2), scratch[N2 * K2:N2 * K2 + N2]
    off = N2 * K2 + N2
    Gp, sp = scratch[off:off + Np * Kp].view(Np, Kp), scratch[off + Np * Kp:]
    probs = [dict(dy=dz, x=h, dW=G2, dbias=s2, gelu=bool(h_gelu)),
             dict(dy=du, x=y, dW=fc1[0], dbias=fc1[1], stats=st2, ln_w=n2w, ln_b=n2b),
             dict(dy=dy, x=o, dW=Gp, dbias=sp),
             dict(dy=dqkv, x=x, dW=qkv[0], dbias=qkv[1], stats=st1 if n1w is not None else None, ln_w=n1w, ln_b=n1b)]
    if not linear_wgrad_group(probs):
        return False
    check(_l().leod_layerscale_finalize(_p(W2), _p(b2), _p(g2), _p(G2), _p(s2), _p(dW2), _p(db2), _p(dg2), N2, K2, _stream()), 'layerscale_finalize')
    check(_l().leod_layerscale_finalize(_p(Wp), _p(bp), _p(g1), _p(Gp), _p(sp), _p(dWp), _p(dbp), _p(dg1), Np, Kp, _stream()), 'layerscale_finalize')
    return True


def layernorm_fwd(x, w, b, want_stats=False, eps=1e-5):
    for t, n in ((x, 'x'), (w, 'w'), (b, 'b')):
        _ck(t, name=n)
    C = x.shape[-1]
    M = x.numel() // C
    y = _empty(x.shape, x)
    stats = _empty((M, 2), x) if want_stats else None
    check(_l().leod_layernorm_fwd(_p(x), _p(w), _p(b), _p(y), _p(stats), M, C, eps, _stream()), 'layernorm_fwd')
    return y, stats


def layernorm_bwd(dn, x, stats, w, dres, dw, db, eps=1e-5):
    for t, n in ((dn, 'dn'), (x, 'x'), (stats, 'stats'), (w, 'w'), (dres, 'dres'), (dw, 'dw'), (db, 'db')):
        _ck(t, name=n)
    C = x.shape[-1]
    M = x.numel() // C
    dx = _empty(x.shape, x)
    check(_l().leod_layernorm_bwd(_p(dn), _p(x), _p(stats), _p(w), _p(dres), _p(dx), _p(dw), _p(db), M, C, eps,
                                   _stream()), 'layernorm_bwd')
    return dx


def linear_dgrad_ln_bwd(dy, W, x, stats, ln_w, dres, dw, db, eps=1e-5):
    """dx of  x -> LayerNorm -> Linear(W)  from dy: LN-backward(dy @ W) + dres; dw / db accumulate the LayerNorm weight / bias
    gradients.  One fused launch for the stage-1 shapes, leod_linear_dgrad + leod_layernorm_bwd otherwise."""
    dy16 = dy.dtype is torch.bfloat16
    _ck(dy, torch.bfloat16 if dy16 else F32, 'dy')
    for t, n in ((W, 'W'), (x, 'x'), (stats, 'stats'), (ln_w, 'ln_w'), (dres, 'dres'), (dw, 'dw'), (db, 'db')):
        _ck(t, name=n)
    N, K = W.shape
    M = dy.numel() // N
    dx = _empty(x.shape, x)
    rc = _l().leod_linear_dgrad_lnbwd(_p(dy), _p(W), _p(x), _p(stats), _p(ln_w), _p(dres), _p(dx), _p(dw), _p(db), M, N, K,
                                      1 if dy16 else 0, _stream())
    if rc == -3:                                            # shape outside the fused kernel's coverage
        return layernorm_bwd(linear_dgrad(dy, W), x, stats, ln_w, dres, dw, db, eps)
    check(rc, 'linear_dgrad_lnbwd')
    return dx


def layerscale_bwd(dz, t, gamma, dgamma):
    for tt, n in ((dz, 'dz'), (t, 't'), (gamma, 'gamma'), (dgamma, 'dgamma')):
        _ck(tt, name=n)
    C = dz.shape[-1]
    M = dz.numel() // C
    dt = _empty(dz.shape, dz)
    check(_l().leod_layerscale_bwd(_p(dz), _p(t), _p(gamma), _p(dt), _p(dgamma), M, C, _stream()), 'layerscale_bwd')
    return dt


def layerscale_linear_wgrad(dz, h, W, b, gamma, dW, db, dgamma, h_gelu=None):
    """Weight, bias and LayerScale gradients of z = res + gamma * (h W^T + b) from dz alone: the un-scaled G = dz^T h goes
    into a scratch buffer (one wgrad launch), ``leod_layerscale_finalize`` turns it into dW, db and dgamma -- the stored
    pre-scale tensor of the forward pass and the scaled copy of dz are not needed."""
    N, K = W.shape
    scratch = StatArena.zeros((N * K + N,), dz.device, torch.float32)         # one memset per step instead of 16 fill kernels
    G, s = scratch[:N * K].view(N, K), scratch[N * K:]
    linear_wgrad(dz, h, G, s, x_gelu=h_gelu)
    check(_l().leod_layerscale_finalize(_p(W), _p(b), _p(gamma), _p(G), _p(s), _p(dW), _p(db), _p(dgamma), N, K, _stream()),
          'layerscale_finalize')


# ---------------------------------------------------------------------------------------------------
# convolutions
# ---------------------------------------------------------------------------------------------------
def _out_hw(H, W, ks, stride, pad):
    return (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1


def stem_conv_fwd(x_nchw, w, padded_hw, stride, pad):
    """x [B,Cin,H,W] uint8|fp32 NCHW (unpadded) -> y [B,Ho,Wo,N] NHWC."""
    if x_nchw.dtype not in (torch.uint8, F32):
        raise LeodHipError(f'stem input must be uint8 or float32, got {x_nchw.dtype}')
    _ck(x_nchw, x_nchw.dtype, 'x')
    _ck(w, name='w')
    B, Cin, H, W = x_nchw.shape
    N, ks = w.shape[0], w.shape[-1]
    Ho, Wo = _out_hw(padded_hw[0], padded_hw[1], ks, stride, pad)
    y = torch.empty((B, Ho, Wo, N), dtype=F32, device=x_nchw.device)
    check(_l().leod_stem_conv_fwd(_p(x_nchw), 1 if x_nchw.dtype == torch.uint8 else 0, _p(w), _p(y), B, Cin, H, W,
                                   padded_hw[0], padded_hw[1], N, ks, stride, pad, _stream()), 'stem_conv_fwd')
    return y


def stem_conv_wgrad(dy, x_nchw, dw, padded_hw, stride, pad):
    _ck(dy, name='dy')
    _ck(x_nchw, x_nchw.dtype, 'x')
    _ck(dw, name='dw')
    B, Cin, H, W = x_nchw.shape
    N, ks = dw.shape[0], dw.shape[-1]
    check(_l().leod_stem_conv_wgrad(_p(dy), _p(x_nchw), 1 if x_nchw.dtype == torch.uint8 else 0, _p(dw), B, Cin, H, W,
                                     padded_hw[0], padded_hw[1], N, ks, stride, pad, _stream()), 'stem_conv_wgrad')


BF16_GRADS = True          # 16-bit modes: du (and dqkv) stored as bf16
STAT_REPLICAS = 32          # most copies of the BatchNorm (sum, sumsq) accumulators a conv epilogue spreads its atomics over


def stat_replicas(rows: int) -> int:
    """Replica count for a conv with ``rows`` output pixels: one per ~1280 rows (80 wave tiles), a power of two in [1, 32] -- the
    consumer (bn_silu_fwd) folds the replicas in every workgroup, so small layers should not carry 32 of them."""
    r = 1
    while r < STAT_REPLICAS and r * 1280 < rows:
        r *= 2
    return r


class PackCache:
    """Packed copies of the conv weights (K-contiguous / per-tap bf16, written by the conv calls themselves into ``wpack``), kept for
    as long as the weights they were made from are unchanged: the 20 3x3 convs of PAFPN + head each re-packed their weights on every
    forward AND dgrad call (40 pack launches per training step) although weights change once per optimiser step.

    An entry is valid for (weight tensor object, direction, geometry) while the tensor's address, its torch version counter, the
    library's precision mode and ``epoch`` are unchanged.  ``epoch`` is advanced by whatever rewrites parameters behind torch's back -- the fused AdamW
    kernel (``FlatParams.adamw_step``) -- and must be advanced (``PackCache.invalidate()``) by any other code that edits parameter
    memory through another alias (e.g. in-place ops on ``FlatParams.data``; ``FlatParams`` wraps that in ``FlatParams.touch()``)."""
    epoch = 0
    entries = {}
    next_token = 1

    @classmethod
    def get(cls, w, key, nfloats):
        # Identity of the WEIGHT TENSOR OBJECT, not of its address: a token stored on the tensor the first time it is seen.  Keyed by
        # data_ptr alone, the parameter of a NEW module that the allocator placed at the address of a dead module's parameter (same
        # geometry, same version counter, no optimiser step in between -- two evaluation models in one process) hit the dead
        # module's pack: stale weights in that conv (seen as a 25 % flake of the TTA test inside the full suite).
        tok = getattr(w, '_leod_pack_token', None)
        if tok is None:
            tok = cls.next_token
            cls.next_token += 1
            w._leod_pack_token = tok
        k = (tok,) + key
        ver = (w.data_ptr(), w._version, cls.epoch, w.numel(), get_precision())
        e = cls.entries.get(k)
        if e is not None and e[1] == ver and e[0].numel() >= nfloats:
            return e[0], 1
        buf = e[0] if (e is not None and e[0].numel() >= nfloats and e[0].device == w.device) else cls._alloc(nfloats, w.device)
        if len(cls.entries) > 4096:
            cls.entries.clear()
        cls.entries[k] = (buf, ver)
        return buf, 0

    _aux = {}

    @classmethod
    def _alloc(cls, nfloats, device):
        """A pack buffer lives as long as the cache and must never come out of a stream capture's private pool: a buffer from that pool may
        alias a temporary of the captured step, and the launch plans hoist weight packs out of their captured order (csrc/k_plan.hip).
        While the current stream is capturing, the allocation is made on a non-capturing stream (torch's allocator then serves it from the
        ordinary pool; the capture runs in relaxed mode)."""
        if device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
            st = cls._aux.get(device)
            if st is None:
                st = cls._aux[device] = torch.cuda.Stream(device=device)
            with torch.cuda.stream(st):
                return torch.empty(nfloats, dtype=F32, device=device)
        return torch.empty(nfloats, dtype=F32, device=device)

    @classmethod
    def ranges(cls):
        """[(address, bytes)] of every live pack buffer (what a launch plan may hoist a weight pack into)"""
        seen = {}
        for buf, _ in cls.entries.values():
            seen[buf.data_ptr()] = buf.numel() * 4
        return sorted(seen.items())

    @classmethod
    def invalidate(cls):
        cls.epoch += 1


def _is_depthwise(w, cin: int) -> bool:
    """a conv weight [C,1,ks,ks] over a C-channel map (C > 1) is a depthwise convolution (nn.Conv2d(C, C, ks, groups=C))"""
    return w.dim() == 4 and w.shape[1] == 1 and cin > 1 and w.shape[0] == cin


def conv_nhwc_fwd(x, w, bias=None, stride=1, colstats=None, bn=None, bn_eps=1e-5):
    """x [B,H,W,Cin] -> y [B,Ho,Wo,N]; pad = (ks-1)//2.  bn = (weight, bias, running_mean, running_var) -> eval BN+SiLU fused.
    colstats: zero-filled float64 [2,N] or [R,2,N] (R a power of two: the epilogue's atomics are spread over the R copies)."""
    _ck(x, name='x')
    _ck(w, name='w')
    _ck(bias, name='bias')
    _ck(colstats, torch.float64, 'colstats')
    B, H, W, Cin = x.shape
    N, ks = w.shape[0], w.shape[-1]
    pad = (ks - 1) // 2
    Ho, Wo = _out_hw(H, W, ks, stride, pad)
    y = _empty((B, Ho, Wo, N), x)
    bw = bb = brm = brv = None
    if bn is not None:
        bw, bb, brm, brv = bn
        for t in bn:
            _ck(t, name='bn')
    if _is_depthwise(w, Cin):                    # groups == channels: w [C,1,ks,ks] (DWConv.dconv, conv3x3_dws of the ConvLSTM)
        rep = colstats.shape[0] if colstats is not None and colstats.dim() == 3 else 1
        check(_l().leod_dwconv_nhwc_fwd(_p(x), _p(w), _p(bias), _p(y), _p(colstats), rep, _p(bw), _p(bb), _p(brm), _p(brv), bn_eps,
                                         B, H, W, Cin, ks, stride, pad, _stream()), 'dwconv_nhwc_fwd')
        return y
    wpack, valid = PackCache.get(w, ('fwd', B, H, W, stride, bn is not None, bias is not None), N * Cin * ks * ks) if ks > 1 else (None, 0)
    rep = colstats.shape[0] if colstats is not None and colstats.dim() == 3 else 1
    check(_l().leod_conv_nhwc_fwd(_p(x), _p(w), _p(bias), _p(y), _p(colstats), rep, _p(bw), _p(bb), _p(brm), _p(brv), bn_eps,
                                   B, H, W, Cin, N, ks, stride, pad, _p(wpack), valid, _stream()), 'conv_nhwc_fwd')
    return y


def conv_nhwc_dgrad(dy, w, x_shape, stride=1, out=None, accumulate=False):
    _ck(dy, name='dy')
    _ck(w, name='w')
    B, H, W, Cin = x_shape
    N, ks = w.shape[0], w.shape[-1]
    pad = (ks - 1) // 2
    if out is None:
        out = _empty(tuple(x_shape), dy)
        accumulate = False
    _ck(out, name='dx')
    if _is_depthwise(w, Cin):
        check(_l().leod_dwconv_nhwc_dgrad(_p(dy), _p(w), _p(out), 1 if accumulate else 0, B, H, W, Cin, ks, stride, pad, _stream()),
              'dwconv_nhwc_dgrad')
        return out
    wpack, valid = PackCache.get(w, ('dgrad', B, H, W, stride), N * Cin * ks * ks) if ks > 1 else (None, 0)
    check(_l().leod_conv_nhwc_dgrad(_p(dy), _p(w), _p(out), 1 if accumulate else 0, B, H, W, Cin, N, ks, stride, pad,
                                     _p(wpack), valid, _stream()), 'conv_nhwc_dgrad')
    return out


def _int_array(v):
    return (ctypes.c_int * len(v))(*[int(a) for a in v])


def conv3x3_group_ok(xs, ws) -> bool:
    """n <= 8 dense 3x3 convs of one (Cin, Cout) geometry on fp32 NHWC maps: candidates for ``conv3x3_group_fwd`` / ``_dgrad``"""
    w0 = ws[0]
    return (1 < len(xs) <= 8 and all(w.shape == w0.shape and w.dim() == 4 and w.shape[-1] == 3 and w.shape[-2] == 3 for w in ws)
            and all(x.dim() == 4 and x.shape[-1] == w0.shape[1] and x.dtype is F32 and x.is_cuda for x in xs) and not _is_depthwise(w0, xs[0].shape[-1]))


def conv3x3_group_fwd(xs, ws, colstats):
    """[x_k [B,H,W,Cin]], [w_k [N,Cin,3,3]], [colstats_k: zero-filled float64 [R,2,N]] -> [y_k] in ONE launch (stride 1, pad 1), or None
    where the direct kernel does not cover the geometry (run them singly)."""
    n = len(xs)
    for t in list(xs) + list(ws):
        _ck(t, name='conv3x3_group')
    for c in colstats:
        _ck(c, torch.float64, 'colstats')
    N, Cin = ws[0].shape[0], ws[0].shape[1]
    if not _l().leod_conv3x3_group_supported(n, _int_array([x.shape[1] for x in xs]), _int_array([x.shape[2] for x in xs]), Cin, N):
        return None                                           # asked before any pack buffer is handed out (a refused call must not leave one marked valid)
    ys = [_empty(tuple(x.shape[:3]) + (N,), x) for x in xs]
    packs = [PackCache.get(w, ('fwd', x.shape[0], x.shape[1], x.shape[2], 1, False, False), N * Cin * 9) for x, w in zip(xs, ws)]
    rc = _l().leod_conv3x3_group_fwd(n, _ptr_array(xs), _ptr_array(ws), _ptr_array(ys), _ptr_array(colstats),
                                     _int_array([c.shape[0] if c.dim() == 3 else 1 for c in colstats]), _ptr_array([p for p, _ in packs]),
                                     _int_array([v for _, v in packs]), _int_array([x.shape[0] for x in xs]), _int_array([x.shape[1] for x in xs]),
                                     _int_array([x.shape[2] for x in xs]), Cin, N, _stream())
    check(rc, 'conv3x3_group_fwd')
    return ys


def conv3x3_group_dgrad(dys, ws, x_shapes, outs, accumulate) -> bool:
    """outs[k] (+)= input gradient of conv(x_k, w_k) from dys[k] in ONE launch; False where not coverable (nothing was written)."""
    n = len(dys)
    for t in list(dys) + list(ws) + list(outs):
        _ck(t, name='conv3x3_group')
    N, Cin = ws[0].shape[0], ws[0].shape[1]
    if not _l().leod_conv3x3_group_supported(n, _int_array([sh[1] for sh in x_shapes]), _int_array([sh[2] for sh in x_shapes]), N, Cin):
        return False
    packs = [PackCache.get(w, ('dgrad', sh[0], sh[1], sh[2], 1), N * Cin * 9) for w, sh in zip(ws, x_shapes)]
    rc = _l().leod_conv3x3_group_dgrad(n, _ptr_array(dys), _ptr_array(ws), _ptr_array(outs), _int_array([1 if a else 0 for a in accumulate]),
                                       _ptr_array([p for p, _ in packs]), _int_array([v for _, v in packs]), _int_array([sh[0] for sh in x_shapes]),
                                       _int_array([sh[1] for sh in x_shapes]), _int_array([sh[2] for sh in x_shapes]), Cin, N, _stream())
    check(rc, 'conv3x3_group_dgrad')
    return True


def conv3x3_group_wgrad(dys, xs, dws) -> bool:
    """dws[k] += weight gradient of conv(x_k, w_k) from dys[k] (stride 1) in ONE launch (+ one reduce); False where not coverable."""
    n = len(dys)
    for t in list(dys) + list(xs) + list(dws):
        _ck(t, name='conv3x3_group')
    N, Cin = dws[0].shape[0], dws[0].shape[1]
    sizes = [int(_l().leod_conv3x3_group_wgrad_workspace_floats(x.shape[0], x.shape[1], x.shape[2], Cin, N)) for x in xs]
    if not all(sizes):
        return False
    ws = _empty((sum(sizes),), dys[0])
    parts, off = [], 0
    for sz in sizes:
        parts.append(ws[off:off + sz])
        off += sz
    rc = _l().leod_conv3x3_group_wgrad(n, _ptr_array(dys), _ptr_array(xs), _ptr_array(dws), _ptr_array(parts), _int_array([x.shape[0] for x in xs]),
                                       _int_array([x.shape[1] for x in xs]), _int_array([x.shape[2] for x in xs]), Cin, N, _stream())
    if rc == -3:
        return False
    check(rc, 'conv3x3_group_wgrad')
    return True


def conv_nhwc_wgrad(dy, x, dw, dbias=None, stride=1):
    for t, n in ((dy, 'dy'), (x, 'x'), (dw, 'dw'), (dbias, 'dbias')):
        _ck(t, name=n)
    B, H, W, Cin = x.shape
    N, ks = dw.shape[0], dw.shape[-1]
    pad = (ks - 1) // 2
    if _is_depthwise(dw, Cin):
        check(_l().leod_dwconv_nhwc_wgrad(_p(dy), _p(x), _p(dw), _p(dbias), B, H, W, Cin, ks, stride, pad, _stream()), 'dwconv_nhwc_wgrad')
        return
    nws = int(_l().leod_conv_nhwc_wgrad_workspace_floats(B, H, W, Cin, N, ks, stride, pad, 0 if dbias is None else 1))
    ws = _empty((nws,), dy) if nws else None
    check(_l().leod_conv_nhwc_wgrad(_p(dy), _p(x), _p(dw), _p(dbias), _p(ws), B, H, W, Cin, N, ks, stride, pad, _stream()),
          'conv_nhwc_wgrad')


def bn_silu_fwd(z, colstats, w, b, run_mean, run_var, count, eps=1e-5, momentum=0.1, count_dev=None):
    _ck(z, name='z')
    _ck(colstats, torch.float64, 'colstats')
    N = z.shape[-1]
    M = z.numel() // N
    y = _empty(z.shape, z)
    mean = _empty((N,), z)
    rstd = _empty((N,), z)
    rep = colstats.shape[0] if colstats.dim() == 3 else 1
    check(_l().leod_bn_silu_fwd(_p(z), _p(colstats), rep, _p(w), _p(b), _p(y), _p(mean), _p(rstd), _p(run_mean), _p(run_var),
                                 M, N, float(count), _p(count_dev), eps, momentum, _stream()), 'bn_silu_fwd')
    return y, mean, rstd


def _ptr_array_opt(ts):
    return (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])


def _f64_array(v):
    return (ctypes.c_double * len(v))(*[float(a) for a in v])


def _f32_array(v):
    return (ctypes.c_float * len(v))(*[float(a) for a in v])


def bn_silu_fwd_group(zs, colstats, ws, bs, run_means, run_vars, counts, eps, momenta, count_devs=None):
    """``bn_silu_fwd`` for n <= 8 layers of one channel count in ONE launch -> [(y, mean, rstd)]"""
    n = len(zs)
    N = zs[0].shape[-1]
    for z_, c_ in zip(zs, colstats):
        _ck(z_, name='z')
        _ck(c_, torch.float64, 'colstats')
        if z_.shape[-1] != N:
            raise LeodHipError('bn_silu_fwd_group: one channel count per launch')
    ys = [_empty(z_.shape, z_) for z_ in zs]
    means = [_empty((N,), z_) for z_ in zs]
    rstds = [_empty((N,), z_) for z_ in zs]
    cd = None if count_devs is None or all(c is None for c in count_devs) else _ptr_array_opt(count_devs)
    check(_l().leod_bn_silu_fwd_group(n, _ptr_array(zs), _ptr_array(colstats), _int_array([c.shape[0] if c.dim() == 3 else 1 for c in colstats]),
                                       _ptr_array(ws), _ptr_array(bs), _ptr_array(ys), _ptr_array(means), _ptr_array(rstds),
                                       _ptr_array_opt(run_means), _ptr_array_opt(run_vars), _int_array([z_.numel() // N for z_ in zs]), N,
                                       _f64_array(counts), cd, float(eps), _f32_array(momenta), _stream()), 'bn_silu_fwd_group')
    return list(zip(ys, means, rstds))


def bn_silu_bwd_reduce_group(dys, zs, means, rstds, ws, bs, outs):
    n, N = len(zs), zs[0].shape[-1]
    lds = []
    for dy, z_ in zip(dys, zs):
        ld = row_stride(dy, N)
        if not ld or z_.shape[-1] != N:
            raise LeodHipError('bn_silu_bwd_reduce_group: dy contiguous or a channel slice of a contiguous map; one channel count per launch')
        _ck_dtype_dev(dy, F32, 'dy')
        lds.append(ld)
    check(_l().leod_bn_silu_bwd_reduce_group(n, _ptr_array(dys), _ptr_array(zs), _ptr_array(means), _ptr_array(rstds), _ptr_array(ws), _ptr_array(bs),
                                              _ptr_array(outs), _int_array([o.shape[0] if o.dim() == 3 else 1 for o in outs]),
                                              _int_array([z_.numel() // N for z_ in zs]), N, _int_array(lds), _stream()), 'bn_silu_bwd_reduce_group')


def bn_silu_bwd_apply_group(dys, zs, means, rstds, ws, bs, sums, dws, dbs, counts, count_devs=None):
    n, N = len(zs), zs[0].shape[-1]
    lds = [row_stride(dy, N) for dy in dys]
    if not all(lds):
        raise LeodHipError('bn_silu_bwd_apply_group: dy must be contiguous or a channel slice of a contiguous map')
    dzs = [_empty(z_.shape, z_) for z_ in zs]
    cd = None if count_devs is None or all(c is None for c in count_devs) else _ptr_array_opt(count_devs)
    check(_l().leod_bn_silu_bwd_apply_group(n, _ptr_array(dys), _ptr_array(zs), _ptr_array(means), _ptr_array(rstds), _ptr_array(ws), _ptr_array(bs),
                                             _ptr_array(sums), _int_array([o.shape[0] if o.dim() == 3 else 1 for o in sums]), _ptr_array(dzs),
                                             _ptr_array(dws), _ptr_array(dbs), _int_array([z_.numel() // N for z_ in zs]), N, _f64_array(counts), cd,
                                             _int_array(lds), _stream()), 'bn_silu_bwd_apply_group')
    return dzs


class StatArena:
    """Zero-initialised scratch of ONE training step: the BatchNorm statistic accumulators ((sum, sumsq) per conv in the forward pass,
    (sum du, sum du*xhat) in the backward pass: 78 tiny float64 buffers per step) and the fp32 un-scaled weight-gradient scratch of
    the 16 LayerScale layers (``layerscale_linear_wgrad``).  The engine zeroes the used part of the arena with ONE memset at the start
    of a step and the ops take slices; outside an engine step (or when the arena is exhausted) the ops fall back to ``torch.zeros``."""
    buf: Optional[torch.Tensor] = None                # uint8
    off = 0
    high = 0                                          # bytes handed out since the arena was last zeroed in full
    active = False
    SIZE = 48 << 20                                   # bytes

    @classmethod
    def begin_step(cls, device):
        if cls.buf is None or cls.buf.device != torch.device(device):
            cls.buf = torch.zeros(cls.SIZE, dtype=torch.uint8, device=device)
        elif cls.high:
            cls.buf[:cls.high].zero_()                # everything beyond the high-water mark was never handed out: still zero
        cls.off, cls.high, cls.active = 0, 0, True

    @classmethod
    def end_step(cls):
        cls.active = False

    @classmethod
    def swap(cls, buf, off=0, high=0, active=False):
        """Install another arena buffer (a captured step keeps a private one: its replays write at the offsets handed out during capture,
        behind the back of this class's high-water bookkeeping); returns the state to hand back to ``swap`` afterwards."""
        old = (cls.buf, cls.off, cls.high, cls.active)
        cls.buf, cls.off, cls.high, cls.active = buf, off, high, active
        return old

    @classmethod
    def zeros(cls, shape, device, dtype=torch.float64):
        n = 1
        for d in shape:
            n *= d
        nbytes = n * (8 if dtype is torch.float64 else 4)
        n_al = (nbytes + 15) & ~15                    # keep 16-byte alignment
        if cls.active and cls.buf is not None and cls.buf.device == torch.device(device) and cls.off + n_al <= cls.SIZE:
            t = cls.buf[cls.off:cls.off + nbytes].view(dtype).view(shape)
            cls.off += n_al
            cls.high = max(cls.high, cls.off)
            return t
        return torch.zeros(shape, dtype=dtype, device=device)


def row_stride(t: torch.Tensor, N: int) -> int:
    """Row stride (elements) of ``t`` seen as [rows, N] when its rows are evenly spaced -- a contiguous tensor (N) or the channel slice
    ``wide[..., c0:c0 + N]`` of a contiguous wider map, which is what ``torch.cat``'s backward hands out; 0: anything else."""
    if t.is_contiguous():
        return N
    if t.dim() < 2 or t.shape[-1] != N or t.stride(-1) != 1:
        return 0
    ld = t.stride(-2)
    exp = ld
    for d in range(t.dim() - 2, -1, -1):
        if t.shape[d] != 1 and t.stride(d) != exp:
            return 0
        exp *= t.shape[d]
    return ld if (ld >= N and ld % 4 == 0 and t.storage_offset() % 4 == 0) else 0


def bn_bwd_replicas(rows: int) -> int:
    """Copies of the (sum du, sum du*xhat) block the workgroups of bn_silu_bwd_reduce spread their closing atomics over: one per
    2560 rows, a power of two in [1, 16]; bn_silu_bwd_apply folds them in every workgroup."""
    r = 1
    while r < 16 and r * 2560 < rows:
        r *= 2
    return r


def bn_silu_bwd_reduce(dy, z, mean, rstd, w, b, out=None):
    """out: zero-filled float64 [2,N] or [R,2,N] (R replicas, see ``bn_bwd_replicas``).
    dy: contiguous, or a channel slice of a wider contiguous map (see ``row_stride``) -- read in place with its row stride."""
    N = z.shape[-1]
    M = z.numel() // N
    ld = row_stride(dy, N)
    if not ld:
        raise LeodHipError('bn_silu_bwd_reduce: dy must be contiguous or a channel slice of a contiguous map')
    _ck_dtype_dev(dy, F32, 'dy')
    sums = StatArena.zeros((bn_bwd_replicas(M), 2, N), z.device) if out is None else out
    _ck(sums, torch.float64, 'sums')
    rep = sums.shape[0] if sums.dim() == 3 else 1
    check(_l().leod_bn_silu_bwd_reduce(_p(dy), _p(z), _p(mean), _p(rstd), _p(w), _p(b), _p(sums), rep, M, N, ld, _stream()),
          'bn_silu_bwd_reduce')
    return sums


def bn_silu_bwd_apply(dy, z, mean, rstd, w, b, sums, dw, db, count, count_dev=None):
    N = z.shape[-1]
    M = z.numel() // N
    ld = row_stride(dy, N)
    if not ld:
        raise LeodHipError('bn_silu_bwd_apply: dy must be contiguous or a channel slice of a contiguous map')
    _ck_dtype_dev(dy, F32, 'dy')
    dz = _empty(z.shape, z)
    rep = sums.shape[0] if sums.dim() == 3 else 1
    check(_l().leod_bn_silu_bwd_apply(_p(dy), _p(z), _p(mean), _p(rstd), _p(w), _p(b), _p(sums), rep, _p(dz), _p(dw), _p(db),
                                       M, N, float(count), _p(count_dev), ld, _stream()), 'bn_silu_bwd_apply')
    return dz


# ---------------------------------------------------------------------------------------------------
# head tail
# ---------------------------------------------------------------------------------------------------
def _iarr(v: Sequence[int]):
    return (ctypes.c_int * len(v))(*[int(a) for a in v])


def head_pred_fwd(cls_feat, reg_feat, cls_w, cls_b, reg_w, reg_b, obj_w, obj_b, out_train, out_infer, stride, a0):
    for t in (cls_feat, reg_feat, cls_w, cls_b, reg_w, reg_b, obj_w, obj_b, out_train, out_infer):
        _ck(t, name='head_pred')
    B, h, w, Hd = cls_feat.shape
    nc = cls_w.shape[0]
    ref = out_train if out_train is not None else out_infer
    A = ref.shape[1]
    check(_l().leod_head_pred_fwd(_p(cls_feat), _p(reg_feat), _p(cls_w), _p(cls_b), _p(reg_w), _p(reg_b), _p(obj_w),
                                   _p(obj_b), _p(out_train), _p(out_infer), B, h, w, Hd, nc, stride, a0, A, _stream()),
          'head_pred_fwd')


def head_pred_bwd(d_raw, cls_feat, reg_feat, cls_w, reg_w, obj_w, d_cls_w, d_cls_b, d_reg_w, d_reg_b, d_obj_w, d_obj_b, a0,
                  gscale=None):
    B, h, w, Hd = cls_feat.shape
    nc = cls_w.shape[0]
    A = d_raw.shape[1]
    dcf = _empty(cls_feat.shape, cls_feat)
    drf = _empty(reg_feat.shape, reg_feat)
    check(_l().leod_head_pred_bwd(_p(d_raw), _p(cls_feat), _p(reg_feat), _p(cls_w), _p(reg_w), _p(obj_w), _p(dcf), _p(drf),
                                   _p(d_cls_w), _p(d_cls_b), _p(d_reg_w), _p(d_reg_b), _p(d_obj_w), _p(d_obj_b), _p(gscale),
                                   B, h, w, Hd, nc, a0, A, _stream()), 'head_pred_bwd')
    return dcf, drf


def simota_assign(outputs, labels, hws, strides, ignore_label=1024.0):
    """outputs [B,A,5+nc] decoded+logits, labels [B,Nmax,7] -> dict of device tensors (no host sync)."""
    _ck(outputs, name='outputs')
    _ck(labels, name='labels')
    B, A, nch = outputs.shape
    Nmax = labels.shape[1]
    dev = outputs.device
    ws = torch.empty(_l().leod_simota_workspace_floats(B, Nmax, A), dtype=F32, device=dev)
    r = dict(fg_mask=torch.empty((B, A), dtype=torch.uint8, device=dev),
             ignore_mask=torch.empty((B, A), dtype=torch.uint8, device=dev),
             matched_row=torch.empty((B, A), dtype=torch.int32, device=dev),
             matched_valid_idx=torch.empty((B, A), dtype=torch.int32, device=dev),
             pred_iou=torch.empty((B, A), dtype=F32, device=dev),
             num_fg_img=torch.empty((B,), dtype=torch.int32, device=dev),
             totals=StatArena.zeros((3,), dev, torch.int32))
    check(_l().leod_simota_assign(_p(outputs), _p(labels), _p(ws), _p(r['fg_mask']), _p(r['ignore_mask']),
                                   _p(r['matched_row']), _p(r['matched_valid_idx']), _p(r['pred_iou']), _p(r['num_fg_img']),
                                   _p(r['totals']), B, Nmax, nch - 5, len(hws), _iarr([h for h, _ in hws]),
                                   _iarr([w for _, w in hws]), _iarr(strides), float(ignore_label), _stream()),
          'simota_assign')
    return r


def bg_topk_ignore(outputs, labels, assign, k, ignore_label=1024.0):
    """``ignore_bg_k`` of the YOLOX head (yolo_head.py:335-356): marks the top ``k`` fraction of each image's background objectness logits
    in ``assign['ignore_mask']`` (in place), unless the batch holds an ignore box."""
    B, A, nch = outputs.shape
    check(_l().leod_bg_topk_ignore(_p(outputs), _p(labels), _p(assign['fg_mask']), _p(assign['ignore_mask']), B, labels.shape[1], A,
                                    nch - 5, float(k), float(ignore_label), _stream()), 'bg_topk_ignore')


def yolox_loss(outputs, labels, assign, hws, strides, want_grad=True, focal=False, reg_weight=5.0, obj_weight=1.0,
               cls_weight=1.0, grad_scale=1.0, label_w=None):
    """label_w [B, Nmax] (``bbox_loss_weighting``): per-label weights of the IoU / class terms, normalised on the device to mean 1 over
    the batch's foreground anchors."""
    B, A, nch = outputs.shape
    dev = outputs.device
    sums = StatArena.zeros((3,), dev, torch.float64)
    losses = torch.empty((6,), dtype=F32, device=dev)
    d_raw = torch.empty_like(outputs) if want_grad else None
    if label_w is not None:
        _ck(label_w, name='label_w')
        if tuple(label_w.shape) != tuple(labels.shape[:2]):
            raise ValueError('yolox_loss: one weight per label row')
        wsum = StatArena.zeros((1,), dev, torch.float64)
        check(_l().leod_yolox_loss_weighted(_p(outputs), _p(labels), _p(assign['fg_mask']), _p(assign['ignore_mask']),
                                             _p(assign['matched_row']), _p(assign['pred_iou']), _p(assign['totals']), _p(label_w),
                                             _p(wsum), _p(sums), _p(losses), _p(d_raw), B, labels.shape[1], nch - 5, len(hws),
                                             _iarr([h for h, _ in hws]), _iarr([w for _, w in hws]), _iarr(strides),
                                             1 if focal else 0, reg_weight, obj_weight, cls_weight, grad_scale, _stream()),
              'yolox_loss_weighted')
        return losses, d_raw
    check(_l().leod_yolox_loss(_p(outputs), _p(labels), _p(assign['fg_mask']), _p(assign['ignore_mask']),
                                _p(assign['matched_row']), _p(assign['pred_iou']), _p(assign['totals']), _p(sums), _p(losses),
                                _p(d_raw), B, labels.shape[1], nch - 5, len(hws), _iarr([h for h, _ in hws]),
                                _iarr([w for _, w in hws]), _iarr(strides), 1 if focal else 0, reg_weight, obj_weight,
                                cls_weight, grad_scale, _stream()), 'yolox_loss')
    return losses, d_raw


def postprocess_nms(pred, num_classes, conf_thre, nms_thre, class_agnostic=False, max_det=None, vanilla_limit=20000):
    """pred [B,A,5+nc] (mutated in place to xyxy) or, with num_classes=0, [B,A,7] xyxy rows.
    -> det [B,max_det,7], cnt [B] (device tensors, no sync)."""
    _ck(pred, name='pred')
    B, A, _ = pred.shape
    max_det = A if max_det is None else max_det
    det = torch.empty((B, max_det, 7), dtype=F32, device=pred.device)
    cnt = torch.empty((B,), dtype=torch.int32, device=pred.device)
    nws = int(_l().leod_postprocess_nms_workspace_bytes(B, A))
    ws = torch.empty(nws, dtype=torch.uint8, device=pred.device) if nws else None
    check(_l().leod_postprocess_nms(_p(pred), _p(det), _p(cnt), _p(ws), B, A, num_classes, float(conf_thre), float(nms_thre),
                                     1 if class_agnostic else 0, max_det, vanilla_limit, _stream()), 'postprocess_nms')
    return det, cnt


def host_counts(cnt, what='postprocess_nms'):
    """Per-image counts on the host (one sync).  A negative count can only come from calling the C ABI without the NMS
    workspace (``ops.postprocess_nms`` always passes it)."""
    counts = cnt.tolist()
    if counts and min(counts) < 0:
        raise LeodHipError(f'{what}: image {counts.index(min(counts))} overflowed the LDS candidate arrays and no workspace was given')
    return counts


_THRESHOLDS = {}


def _threshold_tensor(thr, device):
    """Per-class thresholds as a device tensor, uploaded once per distinct value (a fresh pageable upload per call made every
    pseudo-label chunk wait for the device at this point)."""
    key = ((float(thr),) if isinstance(thr, (float, int)) else tuple(float(t) for t in thr), str(device))
    t = _THRESHOLDS.get(key)
    if t is None:
        if len(_THRESHOLDS) > 64:
            _THRESHOLDS.clear()
        t = _THRESHOLDS[key] = torch.tensor(key[0], dtype=F32, device=device)
    return t


def pseudo_filter(det, cnt, obj_thr, cls_thr, filter_boxes, frame_hw):
    _ck(det, name='det')
    _ck(cnt, torch.int32, 'cnt')
    B, max_det, _ = det.shape
    ot, ct = _threshold_tensor(obj_thr, det.device), _threshold_tensor(cls_thr, det.device)
    if ot.numel() != ct.numel():
        raise LeodHipError('obj_thresh and cls_thresh must both be floats or per-class lists of equal length')
    lab = torch.empty((B, max_det, 8), dtype=F32, device=det.device)
    lcnt = torch.empty((B,), dtype=torch.int32, device=det.device)
    check(_l().leod_pseudo_filter(_p(det), _p(cnt), _p(lab), _p(lcnt), B, max_det, _p(ot), _p(ct), ot.numel(),
                                   1 if filter_boxes else 0, float(frame_hw[1]), float(frame_hw[0]), _stream()),
          'pseudo_filter')
    return lab, lcnt


# ---------------------------------------------------------------------------------------------------
def adamw_clip_step(p, g, m, v, lr, step, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_value=0.0, grad_scale=1.0,
                    hp_dev=None):
    for t in (p, g, m, v):
        _ck(t, name='adamw buffer')
    check(_l().leod_adamw_clip_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, betas[0], betas[1], eps, weight_decay,
                                     int(step), float(clip_value), float(grad_scale), _p(hp_dev), _stream()), 'adamw_clip_step')


def rows_index_add(dst: torch.Tensor, src: torch.Tensor, idx: torch.Tensor) -> None:
    """dst[idx[j]] += src[j] on dim 0 (fp32, contiguous, UNIQUE int64 indices on the device): one launch, no atomics."""
    _ck(dst, name='dst')
    _ck(src, name='src')
    _ck(idx, torch.int64, 'idx')
    if dst.shape[1:] != src.shape[1:] or src.shape[0] != idx.numel():
        raise ValueError('rows_index_add: src rows must match idx and the row shape of dst')
    row = dst[0].numel()
    check(_l().leod_rows_index_add(_p(dst), _p(src), _p(idx), int(idx.numel()), row, int(dst.shape[0]), _stream()), 'rows_index_add')


def cat2_up_fwd(a: torch.Tensor, b: torch.Tensor, up: bool = False) -> torch.Tensor:
    """out [B,H,W,Ca+Cb] = cat(a (nearest x2 upsampled if ``up``), b) on the channel axis of NHWC maps, one launch."""
    _ck(a, name='a')
    _ck(b, name='b')
    B, H, W, Cb = b.shape
    Ca = a.shape[-1]
    if tuple(a.shape[:3]) != ((B, H // 2, W // 2) if up else (B, H, W)):
        raise ValueError(f'cat2_up: a {tuple(a.shape)} does not match b {tuple(b.shape)} (up={up})')
    out = _empty((B, H, W, Ca + Cb), b)
    check(_l().leod_cat2_up_fwd(_p(a), _p(b), _p(out), B, H, W, Ca, Cb, 1 if up else 0, _stream()), 'cat2_up_fwd')
    return out


def cat2_up_bwd(dout: torch.Tensor, Ca: int, up: bool = False):
    """-> (da [B,H>>up,W>>up,Ca], db [B,H,W,C-Ca]) of ``cat2_up_fwd``."""
    _ck(dout, name='dout')
    B, H, W, C = dout.shape
    Cb = C - Ca
    da = _empty((B, H // 2, W // 2, Ca) if up else (B, H, W, Ca), dout)
    db = _empty((B, H, W, Cb), dout)
    check(_l().leod_cat2_up_bwd(_p(dout), _p(da), _p(db), B, H, W, Ca, Cb, 1 if up else 0, _stream()), 'cat2_up_bwd')
    return da, db


def set_weight_shadow(base: torch.Tensor, shadow: Optional[torch.Tensor]) -> None:
    """Register ``shadow`` (bf16, same length) as the 16-bit copy of the flat fp32 parameter buffer ``base`` (None: withdraw it)."""
    _ck(base, name='weight buffer')
    if shadow is not None and (shadow.dtype != torch.bfloat16 or shadow.numel() != base.numel() or shadow.device != base.device):
        raise ValueError('weight shadow: bf16 tensor of the length and device of the parameter buffer')
    check(_l().leod_set_weight_shadow(_p(base), base.numel(), _p(shadow)), 'set_weight_shadow')


def set_weight_shadow_f16(base: torch.Tensor, shadow_f16: Optional[torch.Tensor]) -> None:
    """Attach ``shadow_f16`` (fp16, same length) to the registration of ``base``: the copy the forward GEMMs of precision mode 16f read."""
    _ck(base, name='weight buffer')
    if shadow_f16 is not None and (shadow_f16.dtype != torch.float16 or shadow_f16.numel() != base.numel() or shadow_f16.device != base.device):
        raise ValueError('weight shadow: fp16 tensor of the length and device of the parameter buffer')
    check(_l().leod_set_weight_shadow_f16(_p(base), _p(shadow_f16)), 'set_weight_shadow_f16')


def unset_weight_shadow_ptr(base_ptr: int) -> None:
    """Withdraw the registration of the buffer that lived at ``base_ptr`` (finaliser of its owner; the tensor may be gone)."""
    try:
        _l().leod_set_weight_shadow(DevPtr(base_ptr, 'f32'), 0, None)
    except Exception:                                      # noqa: BLE001 -- interpreter shutdown
        pass


def weight_shadow_refresh(force: bool = False) -> int:
    """Round the registered parameter buffers whose shadow is stale into their shadows (``force``: all of them, freshness untouched --
    for launches that are being recorded, see ``weight_shadow_pin``).  Returns the number of launches."""
    if not torch.cuda.is_available():
        return 0
    rc = _l().leod_weight_shadow_refresh(1 if force else 0, _stream())
    if rc < 0:
        check(rc, 'weight_shadow_refresh')
    return rc


def weight_shadow_invalidate() -> None:
    check(_l().leod_weight_shadow_invalidate(), 'weight_shadow_invalidate')


def weight_shadow_pin(on: bool) -> None:
    """While a step is being recorded behind a forced refresh, the GEMM launchers read the shadows whatever their freshness flags say."""
    check(_l().leod_weight_shadow_pin(1 if on else 0), 'weight_shadow_pin')


def _ptr_array(ts):
    import ctypes
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def _long_array(v):
    import ctypes
    return (ctypes.c_long * len(v))(*[int(x) for x in v])


def multi_ok(ts) -> bool:
    """Operands the multi-buffer plumbing kernels take: dense device tensors, 16-byte aligned, a multiple of 16 bytes per row."""
    def dense(t):      # rows (dim 0) back to back, each row dense: plain contiguous, or NCHW-shaped views of NHWC memory
        return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))
    return 0 < len(ts) <= 16 and all(t.is_cuda and dense(t) and t.data_ptr() % 16 == 0 and t.numel() > 0 and
                                     (t[0].numel() * t.element_size()) % 16 == 0 for t in ts)


ACTIVATIONS = {'gelu': 0, 'silu': 1, 'swish': 1, 'relu': 2, 'sigmoid': 3, 'tanh': 4, 'relu6': 5, 'leaky_relu': 6, 'elu': 7, 'hard_sigmoid': 8,
               'hardsigmoid': 8, 'hard_swish': 9, 'hardswish': 9, 'mish': 10, 'selu': 11, 'celu': 12, 'hard_mish': 13}


def act_glu_fwd(p, act: str, gated: bool):
    """h = a * act(g) with (a | g) the two halves of p's last dim (gated), or h = act(p): the MLP activation options of the attention block."""
    _ck(p, name='p')
    I = p.shape[-1] // 2 if gated else p.shape[-1]
    M = p.numel() // p.shape[-1]
    h = _empty(p.shape[:-1] + (I,), p)
    check(_l().leod_act_glu_fwd(_p(p), _p(h), M, I, ACTIVATIONS[act], 1 if gated else 0, _stream()), 'act_glu_fwd')
    return h


def act_glu_bwd(p, dh, act: str, gated: bool):
    _ck(p, name='p')
    _ck(dh, name='dh')
    I = dh.shape[-1]
    M = dh.numel() // I
    dp = _empty(p.shape, p)
    check(_l().leod_act_glu_bwd(_p(p), _p(dh), _p(dp), M, I, ACTIVATIONS[act], 1 if gated else 0, _stream()), 'act_glu_bwd')
    return dp


def token_mask_fwd_(x, mask, token):
    """x[mask] = token in place: x [.., C] contiguous rows, mask bool with one entry per row, token [C] (maxvit_rnn.py:190-192)."""
    _ck(x, name='x')
    _ck(token, name='token')
    C = x.shape[-1]
    if mask.dtype is not torch.bool or not mask.is_contiguous() or mask.device != x.device or mask.numel() * C != x.numel():
        raise LeodHipError('token_mask: one contiguous bool per row on the device of x')
    check(_l().leod_token_mask_fwd(_p(x), _p(mask), _p(token), mask.numel(), C, _stream()), 'token_mask_fwd')
    return x


def token_mask_bwd_(dx, mask, dtoken):
    """dtoken += sum of the masked rows of dx; those rows of dx are zeroed (in place)."""
    _ck(dx, name='dx')
    _ck(dtoken, name='dtoken')
    check(_l().leod_token_mask_bwd(_p(dx), _p(mask), _p(dtoken), mask.numel(), dx.shape[-1], _stream()), 'token_mask_bwd')
    return dx


def rows_masked_zero(tensors, mask):
    """t[b] = 0 where mask[b], for every t in ``tensors`` ([B, ...] each), in one launch (RNNStates.reset)."""
    if mask.dtype is not torch.bool or not mask.is_cuda or not mask.is_contiguous() or not multi_ok(tensors):
        raise LeodHipError('rows_masked_zero: contiguous device tensors with 16-byte rows and a bool device mask')
    B = mask.numel()
    if any(t.shape[0] != B for t in tensors):
        raise LeodHipError('rows_masked_zero: every tensor needs one row per mask entry')
    check(_l().leod_rows_masked_zero(_ptr_array(tensors), _long_array([t[0].numel() * t.element_size() for t in tensors]), len(tensors),
                                      _p(mask), B, _stream()), 'rows_masked_zero')


def copy_multi(dsts, srcs):
    """dst_k[:] = src_k[:] for all k in one launch (same dtype and element count per pair)."""
    if len(dsts) != len(srcs) or not multi_ok(dsts) or not multi_ok(srcs) or \
            any(d.dtype is not s_.dtype or d.shape != s_.shape or d.stride() != s_.stride() or (d.numel() * d.element_size()) % 16
                for d, s_ in zip(dsts, srcs)):
        raise LeodHipError('copy_multi: pairs of dense, 16-byte aligned device tensors of equal dtype, shape and strides')
    check(_l().leod_copy_multi(_ptr_array(dsts), _ptr_array(srcs), _long_array([d.numel() * d.element_size() for d in dsts]), len(dsts),
                                _stream()), 'copy_multi')


def onehot_scale(g: torch.Tensor, n: int, idx: int) -> torch.Tensor:
    """[0, .., g, .., 0] (n entries, g a device scalar at position idx) in one launch."""
    _ck(g.reshape(1), name='g')
    out = torch.empty((n,), dtype=F32, device=g.device)
    check(_l().leod_onehot_scale(_p(g), _p(out), int(n), int(idx), _stream()), 'onehot_scale')
    return out


def set_scalars4(dst, a, b, c, d):
    _ck(dst, name='dst')
    check(_l().leod_set_scalars4(_p(dst), float(a), float(b), float(c), float(d), _stream()), 'set_scalars4')


def stack_hflip_u8(frames) -> torch.Tensor:
    """T frame tensors [B, ...spatial..., W] (uint8 / bool / int8, same shape, contiguous) -> [T, 2B, ..., W]: the frames and, behind them on
    the batch axis, their horizontally flipped copies (torch.cat([torch.stack(frames), torch.stack(frames).flip(-1)], 1)) in one pass."""
    f0 = frames[0]
    if f0.element_size() != 1 or any(f.shape != f0.shape or f.dtype is not f0.dtype or not f.is_cuda or not f.is_contiguous() for f in frames):
        raise LeodHipError('stack_hflip_u8: contiguous one-byte device tensors of one shape expected')
    B, W = f0.shape[0], f0.shape[-1]
    out = torch.empty((len(frames), 2 * B) + tuple(f0.shape[1:]), dtype=f0.dtype, device=f0.device)
    check(_l().leod_stack_hflip_u8(_ptr_array(frames), len(frames), ctypes.c_void_p(out.data_ptr()), B, f0.numel() // (B * W), W, _stream()),
          'stack_hflip_u8')
    return out


def voxelize_u8(x, y, pol, t, bins, height, width, count_cutoff=None, fastmode=True):
    for a in (x, y, pol, t):
        _ck(a, torch.int64, 'events')
    dev = x.device
    ws = torch.empty((2 * bins * height * width,), dtype=torch.int32, device=dev)
    out = torch.empty((2 * bins, height, width), dtype=torch.uint8, device=dev)
    check(_l().leod_voxelize_u8(_p(x), _p(y), _p(pol), _p(t), x.numel(), _p(ws), _p(out), bins, height, width,
                                 0 if count_cutoff is None else int(count_cutoff), 1 if fastmode else 0, _stream()),
          'voxelize_u8')
    return out


def mixed_density_i8(x, y, pol, t, bins, height, width, count_cutoff=None):
    """MixedDensityEventStack.construct (data/utils/representations.py:132-221): int64 events sorted in time -> int8 [bins, H, W]."""
    for a in (x, y, pol, t):
        _ck(a, torch.int64, 'events')
    if count_cutoff is not None and not 0 <= int(count_cutoff) <= 127:
        raise ValueError('count_cutoff must lie in [0, 127]')
    dev = x.device
    ws = torch.empty((bins * height * width,), dtype=torch.int32, device=dev)
    out = torch.empty((bins, height, width), dtype=torch.int8, device=dev)
    check(_l().leod_mixed_density_i8(_p(x), _p(y), _p(pol), _p(t), x.numel(), _p(ws), _p(out), bins, height, width,
                                     -1 if count_cutoff is None else int(count_cutoff), _stream()), 'mixed_density_i8')
    return out



# ---------------------------------------------------------------------------------------------------
# launch plans: a captured hipGraph replayed as plain stream launches (csrc/k_plan.hip)
# ---------------------------------------------------------------------------------------------------
class LaunchPlan:
    """``LaunchPlan(torch.cuda.CUDAGraph(keep_graph=True) after capture)``: ``launch()`` enqueues every kernel / memset / copy of
    the captured work on torch's current stream (lane 0) and on the plan's own side streams (parallel branches of the capture), in one
    C loop.  The CUDAGraph object (it owns the hipGraph whose argument blocks the plan borrows, and the memory pool) is kept alive here."""

    def __init__(self, graph: 'torch.cuda.CUDAGraph', max_lanes: int = 8):
        self.graph = graph
        raw = graph.raw_cuda_graph()
        # weight-pack kernels may leave their captured position only when they write a persistent pack buffer (PackCache)
        rg = PackCache.ranges()
        check(_l().leod_plan_set_hoist_ranges(_long_array([a for a, _ in rg]), _long_array([b for _, b in rg]), len(rg)), 'plan_set_hoist_ranges')
        h = int(_l().leod_plan_create(ctypes.c_void_p(int(raw)), int(max_lanes)))
        if h <= 0:
            raise LeodHipError(f'leod_plan_create: {_l().leod_plan_last_error().decode()} (rc {h})')
        self.handle = h
        info = (ctypes.c_int * 9)()
        check(_l().leod_plan_info(h, info), 'plan_info')
        self.info = dict(zip(('kernels', 'memsets', 'memcpys', 'empty', 'lanes', 'events', 'waits', 'ops', 'collectives'), list(info)))

    def launch(self, join: bool = True):
        """``join=False``: the current stream does not wait for the plan's side lanes (call ``join()`` before the next launch of this plan
        and before anything reads what they wrote)."""
        if join:
            check(_l().leod_plan_launch(self.handle, _stream()), 'plan_launch')
        else:
            check(_l().leod_plan_launch_nojoin(self.handle, _stream()), 'plan_launch_nojoin')

    def join(self):
        check(_l().leod_plan_join(self.handle, _stream()), 'plan_join')

    def rebase_input(self, captured: torch.Tensor, new: torch.Tensor) -> int:
        """Re-point the plan's input-reading kernels (the stem convolution and its weight gradient) from the buffer they were captured with
        to ``new`` (same shape / dtype / layout).  -> number of kernels re-pointed (0: none here)."""
        n = int(_l().leod_plan_rebase_input(self.handle, ctypes.c_void_p(captured.data_ptr()), captured.numel() * captured.element_size(),
                                             ctypes.c_void_p(new.data_ptr())))
        if n < 0:
            raise LeodHipError(f'leod_plan_rebase_input: rc {n}')
        return n

    def dump(self, path: str):
        check(_l().leod_plan_dump(self.handle, path.encode()), 'plan_dump')

    def close(self):
        if getattr(self, 'handle', 0):
            _l().leod_plan_destroy(self.handle)
            self.handle = 0
        self.graph = None

    def __del__(self):
        try:
            self.close()
        except Exception:                                     # interpreter shutdown
            pass

# ---------------------------------------------------------------------------------------------------
# live kernel timing for bench.py's roofline object
# ---------------------------------------------------------------------------------------------------
_PROBE = None


class KernelProbe:
    """Brackets every launch of the probed kernel FAMILIES with HIP events on the launch stream (torch's current stream is
    the stream every leod_* call is enqueued on) and tallies their algorithmic bytes / flops:
    achieved GB/s = sum(bytes) / sum(event time).  Families: ``linear_wgrad`` (the weight-gradient GEMM of the Linear layers -- ``wgrad_wide_bf16_kernel`` and its reduce kernel in precision mode
    bf16, ``wgradw_kernel`` in f32 mode -- the largest
    single kernel family of the training step) and ``linear_gemm`` (forward / dgrad GEMMs of the Linear layers: row-streaming,
    LDS-staged and wide-tile kernels).  With ``families=True`` every C entry point is bracketed as well (``family_ms``): the
    benchmark line then carries its own per-family time table."""

    def __init__(self, targets=('linear_wgrad', 'linear_gemm'), kernel_names=None, families=True):
        global _PROBE, _LIB
        self.targets = tuple(targets)
        self.kernel_names = kernel_names or {'linear_wgrad': 'wgrad_wide_bf16_kernel + wgrad_wide_reduce_kernel (bf16 mode) / wgradw_kernel<.., XRows> (f32 mode)',
                                             'linear_gemm': 'rowstream* / gemm_lds_kernel / gemm_wide_bf16_kernel (Linear forward + dgrad)'}
        self.step = 0                      # index of the probe step being recorded (mark_step() closes one)
        self.events = {t: [] for t in self.targets}     # target -> [(step, e0, e1, bytes, flops, bytes16, rows)]
        self.fam_events = {}               # C entry point -> [(step, e0, e1)]
        self.calls = []                    # (C entry point, small integer arguments, start event, end event, step) in launch order
        self.real_lib = _l()
        if families:
            _LIB = _ProbedLib(self.real_lib, self)
        _PROBE = self

    def mark_step(self):
        """Closes one probe step: figures are medians over the steps recorded (a host stall inside one bracket -- first use of a code
        object, a page fault of the launch thread -- lands in one step's sum and the median drops it)."""
        self.step += 1

    @property
    def steps(self):
        return max(self.step, 1)

    def begin(self, target, nbytes, flops=0.0, rows=None, nbytes16=None):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.events[target].append((self.step, e0, e1, nbytes, flops, nbytes if nbytes16 is None else nbytes16, rows))
        return e1

    def cancel(self, target):
        """Drops the bracket ``begin`` opened last for ``target`` (the call it was opened for launched nothing)."""
        self.events[target].pop()

    def amend(self, target, nbytes, flops, nbytes16=None):
        """Sets the work of the bracket ``begin`` opened last for ``target`` (the byte count was not known when it was opened)."""
        st, e0, e1, _, _, _, rows = self.events[target][-1]
        self.events[target][-1] = (st, e0, e1, nbytes, flops, nbytes if nbytes16 is None else nbytes16, rows)

    def close(self):
        global _PROBE, _LIB
        _PROBE = None
        _LIB = self.real_lib
        torch.cuda.synchronize()

    @staticmethod
    def _median(v):
        v = sorted(v)
        n = len(v)
        return 0.0 if not n else (v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2]))

    def family_ms(self, steps=None, top=8):
        """{C entry point: ms per step}, largest first: per probe step the sum of the entry point's event brackets, then the MEDIAN over
        the recorded steps (single-stream probe steps)."""
        tot = {}
        for k, v in self.fam_events.items():
            per = [0.0] * self.steps
            for st, a, b in v:
                per[min(st, self.steps - 1)] += a.elapsed_time(b)
            tot[k] = self._median(per)
        return {k: round(v, 3) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:top]}

    def call_table(self, step=0):
        """[(C entry point, small integer arguments (shapes / flags), us)] of every bracketed launch of one probe step, in launch order."""
        return [(n, ints, round(1e3 * a.elapsed_time(b), 1)) for n, ints, a, b, st in self.calls if st == step]

    def finish(self, peak_gbs, peak_tflops=157.3, target=None):
        """Roofline object of one probed family.  The bound is chosen by the family's arithmetic intensity
        (sum flops / sum algorithmic bytes) against the ridge peak_tflops / peak_gbs; ``achieved`` is in the unit of that
        bound, the other roof is reported alongside.  Times: the family's summed event brackets of the MEDIAN probe step.
        ``achieved`` / ``frac`` are denominated in SURVEY 8(d)'s bytes (activation and gradient operands once at 2 bytes, parameter
        gradients at 4: ``algorithmic_bytes_16bit_per_launch``); the figure with every operand at its stored width is kept beside it."""
        if _PROBE is self:
            self.close()
        target = target or self.targets[0]
        ev = self.events[target]
        if not ev:
            return None
        per = [0.0] * self.steps
        for e in ev:
            per[min(e[0], self.steps - 1)] += e[1].elapsed_time(e[2])
        ms = self._median(per)                                   # one step's worth of launches
        pick = min(range(self.steps), key=lambda i: (abs(per[i] - ms), i))
        one = [e for e in ev if min(e[0], self.steps - 1) == pick]
        n = len(one)
        stored, flops, nbytes = sum(e[3] for e in one), sum(e[4] for e in one), sum(e[5] for e in one)
        gbs = nbytes / (ms * 1e-3) / 1e9
        tfl = flops / (ms * 1e-3) / 1e12
        ridge = peak_tflops * 1e12 / (peak_gbs * 1e9)
        intensity = flops / max(nbytes, 1.0)
        out = {'kernel': self.kernel_names.get(target, target), 'launches': n, 'probe_steps': self.steps,
               'family_ms_per_probe_step': [round(v, 3) for v in per], 'avg_us': round(1e3 * ms / n, 3),
               'algorithmic_bytes_16bit_per_launch': round(nbytes / n, 1), 'algorithmic_bytes_stored_width_per_launch': round(stored / n, 1),
               'algorithmic_flops_per_launch': round(flops / n, 1),
               'flop_per_byte': round(intensity, 2), 'ridge_flop_per_byte': round(ridge, 2),
               'hbm_achieved_GBs': round(gbs, 2), 'hbm_frac': round(gbs / peak_gbs, 5),
               'hbm_frac_stored_width': round(stored / (ms * 1e-3) / 1e9 / peak_gbs, 5),
               'mfma_achieved_TFLOPs': round(tfl, 2), 'mfma_frac': round(tfl / peak_tflops, 5), 'traffic': None}
        # the same family split by the row count of the launch (= the stage of the backbone): the long row ranges of stages 1-2 are
        # the HBM-bound launches, the short ones of stages 3-4 carry the same flops on 1/4 - 1/16 of the bytes
        rows = sorted({e[6] for e in one if e[6] is not None}, reverse=True)
        if rows:
            split = []
            for r in rows:
                evs = [e for e in one if e[6] == r]
                t = sum(e[1].elapsed_time(e[2]) for e in evs) * 1e-3
                by, fl = sum(e[5] for e in evs), sum(e[4] for e in evs)
                split.append({'rows': int(r), 'launches': len(evs), 'avg_us': round(1e6 * t / len(evs), 1), 'hbm_GBs': round(by / t / 1e9, 1),
                              'hbm_frac': round(by / t / 1e9 / peak_gbs, 4), 'mfma_TFLOPs': round(fl / t / 1e12, 1)})
            out['by_rows'] = split
        if intensity >= ridge:
            out.update(bound='mfma', achieved=round(tfl, 2), peak=peak_tflops, unit='TFLOP/s', frac=round(tfl / peak_tflops, 5))
        else:
            out.update(bound='hbm', achieved=round(gbs, 2), peak=peak_gbs, unit='GB/s', frac=round(gbs / peak_gbs, 5))
        return out


class _ProbedLib:
    """Stand-in for the CDLL while a KernelProbe with ``families=True`` is active: every leod_* launch is bracketed with events."""

    def __init__(self, lib, probe):
        self._lib, self._probe, self._cache = lib, probe, {}

    def __getattr__(self, name):
        fn = self._cache.get(name)
        if fn is None:
            real = getattr(self._lib, name)
            if not name.startswith('leod_') or name.endswith(('_ok', '_mode', '_bytes', '_floats', 'get_precision', 'set_precision')):
                fn = real
            else:
                probe = self._probe

                def fn(*a, _real=real, _name=name):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    rc = _real(*a)
                    e1.record()
                    probe.fam_events.setdefault(_name, []).append((probe.step, e0, e1))
                    probe.calls.append((_name, [x for x in a if isinstance(x, int) and not isinstance(x, bool) and 0 < x < (1 << 24)], e0, e1, probe.step))
                    return rc
            self._cache[name] = fn
        return fn


def _probe(name, nbytes, flops=0.0, rows=None, nbytes16=None):
    """``nbytes``: algorithmic bytes of the launch with every operand at its STORED width; ``nbytes16``: the SURVEY 8(d) figure -- every
    activation / gradient operand once at 2 bytes (the reference's autocast class), parameters and their gradients at 4."""
    if _PROBE is not None and name in _PROBE.targets:
        return _PROBE.begin(name, nbytes, flops, rows, nbytes16)
    return None
