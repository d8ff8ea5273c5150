"""Parity of every hand-written HIP kernel (through the C ABI / leod_amd.ops) against the CPU oracle
or a plain PyTorch fp32 CPU reference of the same op.  Needs a real MI355X: ``pytest -m gpu``.

Tolerances: integer / index / byte outputs bit-exact; fp32 kernels 2e-5 relative (+ abs floor) for
forward values, 2e-4 for gradients that are long reductions (weight grads accumulate over M rows
with fp32 atomics in nondeterministic order)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import backbone as ob  # noqa: E402
from oracle import head as oh  # noqa: E402
from oracle import postproc as op  # noqa: E402
from oracle.synth import synth_labels  # noqa: E402


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from leod_amd import ops as _ops
    return _ops


DEV = 'cuda'


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


# tests/test_bf16_gpu.py re-runs the contraction tests of this module in precision mode 'bf16' (operands rounded to bf16, fp32
# accumulation) against the same fp32 references, with the tolerance SURVEY 8(c) states for that mode: 3e-2 of the
# tensor's magnitude for the worst element, 0.75e-2 rms (a bf16 operand carries 2^-9 relative rounding error; the reference's own fp16 autocast is the comparison class)
MODE = {'bf16': False, 'fwd16f': False}
BF16_RTOL = 3e-2
# mode '16f' computes the FORWARD contractions on fp16 operands (11 significand bits against bf16's 8): forward tensors (``close(..., fwd=True)``)
# get their own, 8x tighter bound there -- worst element 4e-3 of the tensor's magnitude, 1e-3 rms -- so that a forward kernel that fell back
# to bf16 operands (2^-9 per operand: ~1e-2 worst element) fails at kernel level.  Gradient tensors keep the bf16 bound in both modes (the
# gradient contractions use bf16 operands in both).
F16_FWD_RTOL = 4e-3


def close(a, b, rtol=2e-5, atol=2e-6, what='', fwd=False):
    """fp32 parity: |a-b| <= atol + rtol*|b| with the absolute floor scaled by the tensor's magnitude
    (fp32 dot products of length K carry ~sqrt(K)*eps*max|term| of summation-order noise)."""
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().float().cpu().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    scale = float(np.abs(b).max()) if b.size else 1.0
    if MODE['bf16']:
        err = np.abs(a - b)
        tol, name = (F16_FWD_RTOL, '16f-mode forward') if (fwd and MODE['fwd16f']) else (BF16_RTOL, 'bf16-mode')
        rms, rms_b = np.sqrt((err ** 2).mean()), max(np.sqrt((b ** 2).mean()), 1e-30)
        assert err.max() <= tol * scale + atol, f'{what}: {name} max error {err.max():.3e} vs {tol} * {scale:.3e}'
        assert rms <= 0.25 * tol * rms_b + atol, f'{what}: {name} rms error {rms:.3e} vs {0.25 * tol} * {rms_b:.3e}'
        return
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol + 5e-6 * scale if atol > 0 else 0, err_msg=what)


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,N,K,ln,act', [(200, 144, 48, True, False), (333, 192, 48, True, True),
                                          (64, 1152, 384, True, False), (130, 96, 32, False, False),
                                          (77, 64, 16, True, True), (4096, 256, 64, True, True),
                                          # large M: LDS-staged GEMM, row-layout epilogue, row_stats prologue
                                          (20000, 144, 48, True, False), (33333, 192, 48, True, True),
                                          (17000, 1152, 384, True, False), (24000, 96, 96, False, True),
                                          (9000, 128, 64, True, True),
                                          # K = 48, N = 144 / 192: wave-autonomous row-streaming kernel (ragged last tile,
                                          # fewer tiles than persistent waves, with / without LayerNorm and GELU)
                                          (70001, 144, 48, True, False), (50000, 192, 48, True, True),
                                          (16390, 144, 48, False, True), (300000, 192, 48, False, False),
                                          # K = 96 in column slabs (stage 2: N = 288 -> 2 x 144, N = 384 -> 3 x 128)
                                          (40003, 288, 96, True, False), (30000, 384, 96, True, True), (20000, 384, 96, False, False),
                                          # K = 64 (RVT-base stage 1): N = 192 whole, N = 256 in two slabs
                                          (30001, 192, 64, True, False), (25000, 256, 64, True, True)])
def test_ln_linear_fwd(ops, M, N, K, ln, act):
    x, W, b = rnd((M, K), 1), rnd((N, K), 2, 0.2), rnd((N,), 3, 0.1)
    lw, lb = 1 + 0.2 * rnd((K,), 4), 0.1 * rnd((K,), 5)
    xn = F.layer_norm(x, (K,), lw, lb, 1e-5) if ln else x
    ref = F.linear(xn, W, b)
    out, a, st = ops.ln_linear_fwd(x.to(DEV), lw.to(DEV) if ln else None, lb.to(DEV) if ln else None, W.to(DEV),
                                   b.to(DEV), want_act=act, want_stats=True)
    if out.dtype is torch.float16:
        # precision mode bf16, LayerNorm -> fc1 -> GELU: the pre-activation comes back ONCE, as fp16 (consumers apply GELU on load)
        assert act and ln and a is None and MODE['bf16']
        close(out.float(), ref, what='linear (fp16 pre-activation)', fwd=True)
    else:
        close(out, ref, what='linear', fwd=True)
        if act:
            close(a, F.gelu(ref), what='gelu', fwd=True)
    if ln:
        close(st[:, 0], x.mean(1), atol=1e-6)
        close(st[:, 1], 1 / torch.sqrt(x.var(1, unbiased=False) + 1e-5), rtol=1e-5)


@pytest.mark.parametrize('M,N,K', [(200, 48, 48), (129, 384, 1536), (64, 32, 128),
                                   (30000, 48, 192), (17000, 384, 1536), (65000, 48, 48), (9001, 64, 256),
                                   (40005, 48, 144)])
def test_linear_lsres_fwd(ops, M, N, K):
    a, W, b, g, res = rnd((M, K), 1), rnd((N, K), 2, 0.2), rnd((N,), 3, 0.1), rnd((N,), 4), rnd((M, N), 5)
    t = F.linear(a, W, b)
    out, tt = ops.linear_lsres_fwd(a.to(DEV), W.to(DEV), b.to(DEV), g.to(DEV), res.to(DEV))
    close(tt, t, fwd=True)
    close(out, res + g * t, fwd=True)
    # the training step does not keep t (LayerScale gradient from the un-scaled wgrad): K = 144 / 192 -> 48 then runs on the
    # row-streaming kernel
    out2, none = ops.linear_lsres_fwd(a.to(DEV), W.to(DEV), b.to(DEV), g.to(DEV), res.to(DEV), want_t=False)
    assert none is None
    close(out2, res + g * t, fwd=True)


def _attn_ref(qkv, heads, part, window):
    """attention core on [B,H,W,3C] via the oracle's partition helpers (maxvit.py:343-354)."""
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    d = C // heads
    p = ob.window_partition(qkv, part) if window else ob.grid_partition(qkv, part)
    Bp = p.shape[0]
    q, k, v = p.reshape(Bp, -1, heads, 3 * d).transpose(1, 2).chunk(3, dim=3)
    att = ((q @ k.transpose(-2, -1)) * d ** -0.5).softmax(-1)
    o = (att @ v).transpose(1, 2).reshape(Bp, part[0], part[1], C)
    return ob.window_reverse(o, part, (H, W)) if window else ob.grid_reverse(o, part, (H, W))


@pytest.mark.parametrize('B,H,W,C,heads,part', [(2, 16, 20, 48, 2, (8, 10)), (1, 8, 10, 384, 16, (8, 10)),
                                                (2, 4, 6, 16, 2, (2, 3)), (1, 12, 20, 64, 2, (6, 10)),
                                                (3, 32, 40, 96, 4, (8, 10)),
                                                (1, 8, 10, 24, 1, (8, 10)),        # one head: single-head LDS workgroups
                                                (1, 12, 20, 64, 2, (12, 20)),      # 240-token partition (Gen4 720p stress)
                                                (2, 16, 20, 96, 3, (8, 10)),       # d = 32, odd head count
                                                (1, 12, 20, 96, 3, (6, 10)),       # odd head count AND padded partitions (60 of 64)
                                                (1, 12, 20, 32, 1, (6, 10)),       # RVT-tiny stage 1 on Gen4: one head, padded
                                                # the <PT = 4, d = 32, one head> instantiation of the fused LDS backward, padded or not
                                                # (hipcc mis-merged two of its epilogue stores: dq of accumulator rows 2, 3 misplaced)
                                                (1, 16, 16, 32, 1, (8, 8)), (1, 14, 16, 32, 1, (7, 8)), (2, 16, 16, 96, 3, (8, 8)),
                                                (1, 14, 16, 24, 1, (7, 8)), (1, 12, 12, 32, 1, (6, 6)), (1, 10, 14, 32, 1, (5, 7))])
@pytest.mark.parametrize('window', [True, False])
def test_partition_attn(ops, B, H, W, C, heads, part, window):
    qkv = rnd((B, H, W, 3 * C), 7).requires_grad_(True)
    ref = _attn_ref(qkv, heads, part, window)
    dout = rnd(ref.shape, 8)
    ref.backward(dout)
    q = qkv.detach().to(DEV)
    out, lse = ops.partition_attn_fwd(q, heads, part, window, want_lse=True)
    close(out, ref, what='attn fwd', fwd=True)
    dq = ops.partition_attn_bwd(q, dout.to(DEV), lse, heads, part, window)
    close(dq, qkv.grad, rtol=5e-5, atol=5e-6, what='attn bwd')


@pytest.mark.parametrize('M,C,state', [(160, 32, True), (160, 32, False), (70, 48, True), (640, 384, True),
                                       (40960, 48, True), (10240, 96, False), (9000, 192, True)])
def test_convlstm(ops, M, C, state):
    x, h0, c0 = rnd((M, C), 1), rnd((M, C), 2, 0.5), rnd((M, C), 3, 0.5)
    W, b = rnd((4 * C, 2 * C), 4, 0.15).requires_grad_(True), rnd((4 * C,), 5, 0.1).requires_grad_(True)
    xr, hr, cr = x.clone().requires_grad_(True), h0.clone().requires_grad_(True), c0.clone().requires_grad_(True)
    sd = {'l.conv1x1.weight': W.view(4 * C, 2 * C, 1, 1), 'l.conv1x1.bias': b}
    nchw = lambda t: t.t().reshape(1, C, M, 1)  # noqa
    h, c = ob.conv_lstm(nchw(xr), (nchw(hr), nchw(cr)) if state else None, sd, 'l')
    dh, dc = rnd((M, C), 6), rnd((M, C), 7)
    (h.reshape(C, M).t() * dh).sum().add((c.reshape(C, M).t() * dc).sum()).backward()
    xd, hd, cd = x.to(DEV), h0.to(DEV) if state else None, c0.to(DEV) if state else None
    hh, cc, gates = ops.convlstm_fwd(xd, hd, cd, W.detach().to(DEV), b.detach().to(DEV), want_gates=True)
    close(hh, h.reshape(C, M).t(), what='h', fwd=True)
    close(cc, c.reshape(C, M).t(), what='c', fwd=True)
    dg, dcp = ops.convlstm_gates_bwd(dh.to(DEV), dc.to(DEV), gates, cd, cc)
    dW = torch.zeros((4 * C, 2 * C), device=DEV)
    db = torch.zeros((4 * C,), device=DEV)
    if state:
        ops.linear_wgrad(dg, xd, dW, db, x2=hd)
        dx, dhp = ops.linear_dgrad(dg, W.detach().to(DEV), split=C)
        close(dhp, hr.grad, rtol=1e-4, atol=1e-5, what='dh_prev')
        close(dcp, cr.grad, rtol=1e-4, atol=1e-5, what='dc_prev')
        close(dW, W.grad, rtol=2e-4, atol=2e-5, what='dW')
    else:
        dWx = torch.zeros((4 * C, C), device=DEV)
        ops.linear_wgrad(dg, xd, dWx, db)
        dx = ops.linear_dgrad(dg, W.detach()[:, :C].contiguous().to(DEV))
        close(dWx, W.grad[:, :C], rtol=2e-4, atol=2e-5, what='dW')
    close(dx, xr.grad, rtol=1e-4, atol=1e-5, what='dx')
    close(db, b.grad, rtol=2e-4, atol=2e-5, what='db')


@pytest.mark.parametrize('T,B,H,W,C,state', [(4, 1, 7, 10, 48, True),       # 70 rows: ragged last 16-row tile, fused [x|h] kernel
                                             (3, 2, 8, 10, 32, False),      # zero incoming state
                                             (5, 4, 16, 40, 96, True),      # RVT-S stage 2 (fp32 mode: hoisted x projection)
                                             (3, 2, 16, 20, 192, True),     # stage 3 (bf16 mode: hoisted, 12 waves; fp32 mode: per-timestep fallback)
                                             (2, 1, 5, 8, 384, True),       # stage 4: per-timestep kernels
                                             (21, 2, 16, 20, 48, True)])    # the benchmark's sequence length
def test_convlstm_sequence(ops, T, B, H, W, C, state):
    """DWSConvLSTM2d.forward_sequence (ONE launch per direction, csrc/k_lstm.hip, + the time-batched dx / weight-gradient GEMMs) against
    T chained steps of the oracle cell (models/layers/rnn.py:37-70), forward and backward through time."""
    from leod_amd.models.layers.rnn import DWSConvLSTM2d
    x = rnd((T * B, C, H, W), 1)
    h0, c0 = rnd((B, C, H, W), 2, 0.5), rnd((B, C, H, W), 3, 0.5)
    # weight scale ~ 2 / sqrt(2C): pre-activations of unit order (0.15 at C = 384 saturates the gates and amplifies operand rounding)
    Wt, b = rnd((4 * C, 2 * C, 1, 1), 4, 0.15 if C <= 192 else 0.06), rnd((4 * C,), 5, 0.1)
    xr, hr, cr = x.clone().requires_grad_(True), h0.clone().requires_grad_(True), c0.clone().requires_grad_(True)
    Wr, br = Wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    sd = {'l.conv1x1.weight': Wr, 'l.conv1x1.bias': br}
    st = (hr, cr) if state else None
    hs = []
    for t in range(T):
        st = ob.conv_lstm(xr[t * B:(t + 1) * B], st, sd, 'l')
        hs.append(st[0])
    href = torch.cat(hs, 0)
    dh, dc = rnd(href.shape, 6), rnd(st[1].shape, 7)
    (href * dh).sum().add((st[1] * dc).sum()).backward()
    mod = DWSConvLSTM2d(C, dws_conv=False).to(DEV)
    with torch.no_grad():
        mod.conv1x1.weight.copy_(Wt.to(DEV)); mod.conv1x1.bias.copy_(b.to(DEV))
    cl = lambda t: t.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)  # noqa
    xd, hd, cd = cl(x), cl(h0), cl(c0)
    hseq, (hl, clast) = mod.forward_sequence(xd, T, (hd, cd) if state else None)
    close(hseq, href, what='h of all timesteps', fwd=True)
    close(clast, st[1], what='final c', fwd=True)
    close(hl, hs[-1], what='final h', fwd=True)
    (hseq * dh.to(DEV)).sum().add((clast * dc.to(DEV)).sum()).backward()
    close(xd.grad, xr.grad, rtol=1e-4, atol=1e-5, what='dx')
    if state:
        close(hd.grad, hr.grad, rtol=1e-4, atol=1e-5, what='dh0')
        close(cd.grad, cr.grad, rtol=1e-4, atol=1e-5, what='dc0')
        close(mod.conv1x1.weight.grad, Wr.grad, rtol=2e-4, atol=2e-5, what='dW')
    else:
        close(mod.conv1x1.weight.grad[:, :C], Wr.grad[:, :C], rtol=2e-4, atol=2e-5, what='dW_x')
    close(mod.conv1x1.bias.grad, br.grad, rtol=2e-4, atol=2e-5, what='db')


@pytest.mark.parametrize('M,N,K', [(300, 144, 48), (1000, 48, 192), (257, 64, 32), (5000, 96, 96), (100, 1536, 384),
                                   # large M: wave-tiled wgrad (192x48, 48x192, 96x96, 48x48 tiles) and LDS dgrad
                                   (30000, 192, 48), (30000, 48, 192), (20000, 288, 96), (65000, 48, 48),
                                   (9000, 1536, 384), (12345, 144, 48), (16000, 128, 64),
                                   # dgrad of fc2 on the row-streaming kernel (contraction 48 / 96, ragged last tile)
                                   (40007, 48, 192), (20000, 96, 384),
                                   # plain dgrad 192 / 144 -> 48 columns on the narrow row-streaming kernel
                                   (40009, 192, 48), (20011, 144, 48),
                                   (24001, 64, 256)])                # RVT-base stage 1: dgrad of fc2 (contraction 64)
def test_linear_backward(ops, M, N, K):
    x = rnd((M, K), 1).requires_grad_(True)
    W, b = rnd((N, K), 2, 0.2).requires_grad_(True), rnd((N,), 3).requires_grad_(True)
    lw, lb = (1 + 0.2 * rnd((K,), 4)), (0.1 * rnd((K,), 5))
    u = F.linear(F.layer_norm(x, (K,), lw, lb, 1e-5), W, b)
    dy = rnd((M, N), 6)
    u.backward(dy)
    _, _, st = ops.ln_linear_fwd(x.detach().to(DEV), lw.to(DEV), lb.to(DEV), W.detach().to(DEV), b.detach().to(DEV), want_stats=True)
    dW, db = torch.zeros((N, K), device=DEV), torch.zeros((N,), device=DEV)
    ops.linear_wgrad(dy.to(DEV), x.detach().to(DEV), dW, db, stats=st, ln_w=lw.to(DEV), ln_b=lb.to(DEV))
    close(dW, W.grad, rtol=2e-4, atol=5e-5, what='dW (LN input)')
    close(db, b.grad, rtol=2e-4, atol=5e-5, what='db')
    # dgrad with per-column scale, GELU derivative and column sums
    ks = rnd((N,), 7)
    uu = rnd((M, K), 8)
    dn = ops.linear_dgrad(dy.to(DEV), W.detach().to(DEV), kscale=ks.to(DEV))
    close(dn, (dy * ks) @ W.detach(), rtol=5e-5, atol=5e-6, what='dgrad')
    close(ops.linear_dgrad(dy.to(DEV), W.detach().to(DEV)), dy @ W.detach(), rtol=5e-5, atol=5e-6, what='plain dgrad')
    cs = torch.zeros((K,), device=DEV)
    du = ops.linear_dgrad(dy.to(DEV), W.detach().to(DEV), aux_u=uu.to(DEV), colsum=cs)
    uu2 = uu.clone().requires_grad_(True)
    F.gelu(uu2).backward(dy @ W.detach())
    close(du, uu2.grad, rtol=5e-5, atol=5e-6, what='dgrad*gelu')
    close(cs, uu2.grad.sum(0), rtol=2e-4, atol=5e-5, what='colsum')
    du2 = ops.linear_dgrad(dy.to(DEV), W.detach().to(DEV), kscale=ks.to(DEV), aux_u=uu.to(DEV))       # as AttnBlockFn calls it
    uu3 = uu.clone().requires_grad_(True)
    F.gelu(uu3).backward((dy * ks) @ W.detach())
    close(du2, uu3.grad, rtol=5e-5, atol=5e-6, what='dgrad*kscale*gelu')


@pytest.mark.parametrize('M,C', [(500, 48), (70, 384), (1000, 32), (33, 512)])
def test_layernorm_layerscale(ops, M, C):
    x = rnd((M, C), 1).requires_grad_(True)
    w, b = (1 + 0.2 * rnd((C,), 2)).requires_grad_(True), (0.1 * rnd((C,), 3)).requires_grad_(True)
    y = F.layer_norm(x, (C,), w, b, 1e-5)
    dn, dres = rnd((M, C), 4), rnd((M, C), 5)
    y.backward(dn)
    yy, st = ops.layernorm_fwd(x.detach().to(DEV), w.detach().to(DEV), b.detach().to(DEV), want_stats=True)
    close(yy, y)
    for stats in (st, None):
        dw, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        dx = ops.layernorm_bwd(dn.to(DEV), x.detach().to(DEV), stats, w.detach().to(DEV), dres.to(DEV), dw, db)
        close(dx, x.grad + dres, rtol=5e-5, atol=5e-6, what='ln dx')
        close(dw, w.grad, rtol=2e-4, atol=5e-5)
        close(db, b.grad, rtol=2e-4, atol=5e-5)
    g, t, dz = rnd((C,), 6), rnd((M, C), 7), rnd((M, C), 8)
    dg = torch.zeros(C, device=DEV)
    dt = ops.layerscale_bwd(dz.to(DEV), t.to(DEV), g.to(DEV), dg)
    close(dt, dz * g)
    close(dg, (dz * t).sum(0), rtol=2e-4, atol=5e-5)


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('u8,B,H,W,Hp,Wp,N', [
    (True, 2, 60, 90, 64, 96, 48), (False, 2, 60, 90, 64, 96, 48),     # W % 4 != 0 / fp32 input: generic implicit-GEMM path
    (True, 2, 60, 88, 64, 96, 48), (True, 1, 240, 304, 256, 320, 48),   # uint8, W % 4 == 0: LDS-patch stem kernel (Gen1 size)
    (True, 2, 36, 52, 40, 64, 32), (True, 1, 50, 100, 64, 128, 64)])    # tiny / base widths, ragged tiles
def test_stem_conv(ops, u8, B, H, W, Hp, Wp, N):
    Cin = 20
    g = torch.Generator().manual_seed(3)
    x = (torch.rand((B, Cin, H, W), generator=g) < 0.1) * torch.randint(1, 10, (B, Cin, H, W), generator=g)
    x = x.to(torch.uint8) if u8 else x.float() + 0.25
    w = rnd((N, Cin, 7, 7), 4, 0.05).requires_grad_(True)
    xp = F.pad(x.float(), [0, Wp - W, 0, Hp - H])
    ref = F.conv2d(xp, w, None, stride=4, padding=3)
    dy = rnd(ref.shape, 5)
    ref.backward(dy)
    y = ops.stem_conv_fwd(x.to(DEV), w.detach().to(DEV), (Hp, Wp), 4, 3)
    close(y, ref.permute(0, 2, 3, 1), rtol=5e-5, atol=1e-5, fwd=True)
    dw = torch.zeros_like(w.detach(), device=DEV)
    ops.stem_conv_wgrad(dy.permute(0, 2, 3, 1).contiguous().to(DEV), x.to(DEV), dw, (Hp, Wp), 4, 3)
    close(dw, w.grad, rtol=2e-4, atol=1e-4)


@pytest.mark.parametrize('B,H,W,Cin,N,ks,stride', [(2, 16, 24, 16, 32, 3, 2), (2, 32, 40, 48, 96, 3, 2),
                                                   (3, 8, 12, 32, 32, 3, 1), (2, 8, 10, 96, 96, 3, 1),
                                                   (2, 16, 20, 192, 96, 1, 1), (1, 7, 9, 64, 48, 3, 2),
                                                   # large M: LDS GEMM with repacked weights, parity-class dgrad (Q % 128 == 0)
                                                   (8, 64, 80, 48, 96, 3, 2), (16, 32, 40, 96, 96, 3, 1),
                                                   (24, 32, 40, 96, 192, 3, 2), (32, 16, 20, 192, 192, 1, 1)])
def test_conv_nhwc(ops, B, H, W, Cin, N, ks, stride):
    x = rnd((B, Cin, H, W), 1).requires_grad_(True)
    w, bias = rnd((N, Cin, ks, ks), 2, 0.1).requires_grad_(True), rnd((N,), 3).requires_grad_(True)
    ref = F.conv2d(x, w, bias, stride=stride, padding=(ks - 1) // 2)
    dy = rnd(ref.shape, 4)
    ref.backward(dy)
    xn = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
    dyn = dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    cs = torch.zeros((2, N), dtype=torch.float64, device=DEV)
    y = ops.conv_nhwc_fwd(xn, w.detach().to(DEV), bias.detach().to(DEV), stride=stride, colstats=cs)
    refn = ref.detach().permute(0, 2, 3, 1)
    close(y, refn, rtol=5e-5, atol=1e-5, fwd=True)
    close(cs[0], refn.reshape(-1, N).double().sum(0), rtol=1e-5, atol=1e-4)
    close(cs[1], (refn.reshape(-1, N).double() ** 2).sum(0), rtol=1e-5, atol=1e-4)
    dx = ops.conv_nhwc_dgrad(dyn, w.detach().to(DEV), xn.shape, stride=stride)
    close(dx, x.grad.permute(0, 2, 3, 1), rtol=1e-4, atol=1e-5, what='dgrad')
    acc = rnd(tuple(xn.shape), 9).to(DEV)
    dx2 = ops.conv_nhwc_dgrad(dyn, w.detach().to(DEV), xn.shape, stride=stride, out=acc.clone(), accumulate=True)
    close(dx2, x.grad.permute(0, 2, 3, 1) + acc.cpu(), rtol=1e-4, atol=1e-5, what='dgrad acc')
    dw, db = torch.zeros_like(w.detach(), device=DEV), torch.zeros(N, device=DEV)
    ops.conv_nhwc_wgrad(dyn, xn, dw, db, stride=stride)
    close(dw, w.grad, rtol=2e-4, atol=1e-4, what='wgrad')
    close(db, bias.grad, rtol=2e-4, atol=1e-4)


@pytest.mark.parametrize('stat_rep', [1, 32])
def test_conv_bn_eval_and_train(ops, stat_rep):
    B, H, W, Cin, N = 3, 8, 12, 32, 64
    x = rnd((B, Cin, H, W), 1).requires_grad_(True)
    w = rnd((N, Cin, 3, 3), 2, 0.1).requires_grad_(True)
    bw, bb = (1 + 0.2 * rnd((N,), 3)).requires_grad_(True), (0.1 * rnd((N,), 4)).requires_grad_(True)
    rm, rv = 0.2 * rnd((N,), 5), 0.5 + torch.rand(N, generator=torch.Generator().manual_seed(6))
    xn = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
    # eval: folded BN + SiLU epilogue
    ref = F.silu(F.batch_norm(F.conv2d(x, w, None, padding=1), rm, rv, bw, bb, False, 0.1, 1e-5)).detach()
    y = ops.conv_nhwc_fwd(xn, w.detach().to(DEV), None, bn=(bw.detach().to(DEV), bb.detach().to(DEV), rm.to(DEV), rv.to(DEV)))
    close(y, ref.permute(0, 2, 3, 1), rtol=5e-5, atol=1e-5, fwd=True)
    # train: batch statistics
    rm2, rv2 = rm.clone(), rv.clone()
    z = F.conv2d(x, w, None, padding=1)
    ref = F.silu(F.batch_norm(z, rm2, rv2, bw, bb, True, 0.1, 1e-5))
    dy = rnd(ref.shape, 7)
    ref.backward(dy)
    # stat_rep > 1: the conv epilogue spreads its (sum, sumsq) atomics over that many copies, bn_silu_fwd folds them
    cs = torch.zeros((2, N) if stat_rep == 1 else (stat_rep, 2, N), dtype=torch.float64, device=DEV)
    zz = ops.conv_nhwc_fwd(xn, w.detach().to(DEV), None, colstats=cs)
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    M = B * H * W
    yy, mean, rstd = ops.bn_silu_fwd(zz, cs, bw.detach().to(DEV), bb.detach().to(DEV), rmd, rvd, M)
    close(yy, ref.detach().permute(0, 2, 3, 1), rtol=5e-5, atol=1e-5, fwd=True)
    close(rmd, rm2, rtol=1e-5, atol=1e-6)
    close(rvd, rv2, rtol=1e-5, atol=1e-6)
    dyn = dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    sums = ops.bn_silu_bwd_reduce(dyn, zz, mean, rstd, bw.detach().to(DEV), bb.detach().to(DEV))
    dbw, dbb = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV)
    dz = ops.bn_silu_bwd_apply(dyn, zz, mean, rstd, bw.detach().to(DEV), bb.detach().to(DEV), sums, dbw, dbb, M)
    close(dbw, bw.grad, rtol=2e-4, atol=5e-5)
    close(dbb, bb.grad, rtol=2e-4, atol=5e-5)
    dx = ops.conv_nhwc_dgrad(dz, w.detach().to(DEV), xn.shape)
    close(dx, x.grad.permute(0, 2, 3, 1), rtol=2e-4, atol=2e-5)
    # dy as the channel slice torch.cat's backward hands to one branch of a concat: read in place with its row stride, same bits
    wide = torch.randn((B, H, W, N + 96), device=DEV)
    wide[..., 32:32 + N] = dyn
    dys = wide[..., 32:32 + N]
    assert not dys.is_contiguous() and ops.row_stride(dys, N) == N + 96 and ops.row_stride(dys.permute(0, 2, 1, 3), N) == 0
    sums2 = ops.bn_silu_bwd_reduce(dys, zz, mean, rstd, bw.detach().to(DEV), bb.detach().to(DEV))
    dz2 = ops.bn_silu_bwd_apply(dys, zz, mean, rstd, bw.detach().to(DEV), bb.detach().to(DEV), sums2, torch.zeros(N, device=DEV),
                                torch.zeros(N, device=DEV), M)
    close(sums2, sums, rtol=1e-6, atol=1e-7)
    close(dz2, dz, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize('M,N,rep', [(5000, 96, 1), (5000, 96, 4), (40960, 48, 16), (2563, 384, 2), (77, 192, 1), (10240, 192, 4)])
def test_bn_silu_bwd_replicas(ops, M, N, rep):
    """BatchNorm + SiLU backward (network_blocks.py:47-51) on rows: the reduce kernel's 4 / 8-row threads with a ragged tail, its
    closing atomics spread over `rep` copies of the sums, the apply kernel folding them -- against torch autograd in fp64."""
    z = rnd((M, N), 11).double().requires_grad_(True)
    bw, bb = (1 + 0.2 * rnd((N,), 12)).double().requires_grad_(True), (0.1 * rnd((N,), 13)).double().requires_grad_(True)
    dy = rnd((M, N), 14).double()
    mean, var = z.mean(0), z.var(0, unbiased=False)
    F.silu((z - mean) / torch.sqrt(var + 1e-5) * bw + bb).backward(dy)
    zd, dyd = z.detach().float().to(DEV), dy.float().to(DEV)
    meand, rstdd = mean.detach().float().to(DEV), (1.0 / torch.sqrt(var.detach() + 1e-5)).float().to(DEV)
    bwd, bbd = bw.detach().float().to(DEV), bb.detach().float().to(DEV)
    sums = torch.zeros((rep, 2, N) if rep > 1 else (2, N), dtype=torch.float64, device=DEV)
    ops.bn_silu_bwd_reduce(dyd, zd, meand, rstdd, bwd, bbd, out=sums)
    if rep > 1:
        assert int((sums.abs().sum((1, 2)) > 0).sum()) == rep          # every copy took some workgroups' atomics
    dbw, dbb = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV)
    dz = ops.bn_silu_bwd_apply(dyd, zd, meand, rstdd, bwd, bbd, sums, dbw, dbb, M)
    close(dbw, bw.grad.float(), rtol=2e-4, atol=2e-4 * float(bw.grad.abs().max()))
    close(dbb, bb.grad.float(), rtol=2e-4, atol=2e-4 * float(bb.grad.abs().max()))
    close(dz, z.grad.float(), rtol=2e-4, atol=2e-5)


def test_conv_pack_cache_is_keyed_by_tensor_not_address(ops):
    """The packed-weight cache of the 3x3 convs must not serve the pack of a DEAD weight tensor to a new tensor that the allocator
    placed at the same address (same geometry, same version counter): two evaluation models in one process."""
    B, H, W, Cin, N = 2, 8, 10, 32, 48
    xn = rnd((B, H, W, Cin), 1).to(DEV)
    ptrs = set()
    for seed in (2, 3, 4):
        w = rnd((N, Cin, 3, 3), seed, 0.1).to(DEV)                # freed at the end of the iteration: the next one reuses the block
        ptrs.add(w.data_ptr())
        y = ops.conv_nhwc_fwd(xn, w, None)
        ref = F.conv2d(xn.permute(0, 3, 1, 2).cpu(), w.cpu(), None, padding=1).permute(0, 2, 3, 1)
        close(y, ref, rtol=2e-4, atol=2e-5)
        dx = ops.conv_nhwc_dgrad(y, w, xn.shape)
        refdx = F.conv_transpose2d(y.permute(0, 3, 1, 2).cpu(), w.cpu(), None, padding=1).permute(0, 2, 3, 1)
        close(dx, refdx, rtol=2e-4, atol=2e-4)
        del w, y, dx
    assert len(ptrs) < 3, 'the allocator did not reuse the weight block: the scenario was not exercised'


def test_state_plumbing_multi_buffer_kernels(ops):
    """rows_masked_zero == `t[mask] = 0` per tensor (reference modules/utils/detection.py:60-75), copy_multi == copy_ per pair --
    over contiguous and NCHW-shaped-over-NHWC tensors of different row sizes, bit-exact."""
    B = 8
    g = torch.Generator().manual_seed(3)
    ts = [torch.randn((B, 64, 80, 48), generator=g), torch.randn((B, 8, 10, 384), generator=g).permute(0, 3, 1, 2),
          torch.randn((B, 20), generator=g), torch.randn((B, 16, 20, 192), generator=g)]
    mask = torch.tensor([1, 0, 0, 1, 0, 1, 1, 0], dtype=torch.bool)
    ref = [t.clone() for t in ts]
    for r in ref:
        r[mask] = 0
    dev = [t.to(DEV) for t in ts]
    assert not dev[1].is_contiguous() and ops.multi_ok(dev)
    ops.rows_masked_zero(dev, mask.to(DEV))
    for d, r in zip(dev, ref):
        assert torch.equal(d.cpu(), r)
    none = torch.zeros(B, dtype=torch.bool, device=DEV)
    before = [d.clone() for d in dev]
    ops.rows_masked_zero(dev, none)
    assert all(torch.equal(a, b) for a, b in zip(dev, before))
    dst = [torch.zeros_like(d) for d in dev]
    ops.copy_multi(dst, dev)
    assert all(torch.equal(a, b) for a, b in zip(dst, dev))
    with pytest.raises(Exception):
        ops.copy_multi([torch.zeros((B, 3), device=DEV)], [torch.zeros((B, 3), device=DEV)])     # 12-byte rows: refused, not mis-copied


# ---------------------------------------------------------------------------------------------------
HEAD_GEOMS = {'gen1': ((256, 320), (240, 304), 2), 'gen4': ((384, 640), (360, 640), 3), '1mpx': ((768, 1280), (720, 1280), 3)}


def _gen1_case(B, seed, nc=2, with_ignore=False, geom='gen1'):
    (Hp, Wp), frame_hw, nc = HEAD_GEOMS[geom][0], HEAD_GEOMS[geom][1], (nc if geom == 'gen1' else HEAD_GEOMS[geom][2])
    hws, strides = [(Hp // s_, Wp // s_) for s_ in (8, 16, 32)], (8, 16, 32)
    gx, gy, gs = oh.make_grids(hws, strides)
    A = gx.numel()
    g = torch.Generator().manual_seed(seed)
    outputs = torch.cat([torch.stack([(gx + 0.5) * gs, (gy + 0.5) * gs, gs * 3, gs * 2.5], 1)[None].repeat(B, 1, 1)
                         + torch.randn(B, A, 4, generator=g) * 2, torch.randn(B, A, 1 + nc, generator=g) * 2], -1)
    labs = synth_labels(B, frame_hw, nc, seed=seed, max_boxes=6)
    tg = op.batched_yolox_labels(labs)
    if B > 2:
        tg[B - 1] = 0                                   # an image without labels
    if with_ignore:
        tg[0, 0, 0] = 1024
        tg[1, :, 0] = torch.where(tg[1].sum(1) > 0, torch.full_like(tg[1, :, 0], 1024.), tg[1, :, 0])
    return hws, strides, (gx, gy, gs), outputs, tg


@pytest.mark.parametrize('with_ignore', [False, True])
@pytest.mark.parametrize('seed,geom', [(1, 'gen1'), (2, 'gen1'), (3, 'gen1'), (4, 'gen4'), (5, '1mpx')])
def test_simota_and_loss(ops, seed, with_ignore, geom):
    """Gen1 (1680 anchors), Gen4-ds2 (5040) and 1 Mpx (20160 anchors, BASELINE configs[3]) heads."""
    B = 5
    hws, strides, (gx, gy, gs), outputs, tg = _gen1_case(B, seed, with_ignore=with_ignore, geom=geom)
    outr = outputs.clone().requires_grad_(True)
    ref = oh.get_losses(gx, gy, gs, tg.clone(), outr, num_classes=HEAD_GEOMS[geom][2], return_assign=True)
    ref['loss'].backward()
    od, td = outputs.to(DEV), tg.to(DEV)
    asg = ops.simota_assign(od, td, hws, strides)
    assert np.array_equal(asg['fg_mask'].cpu().numpy().astype(bool), ref['_fg_mask'].numpy())
    assert np.array_equal(asg['ignore_mask'].cpu().numpy().astype(bool), ref['_ignore_mask'].numpy())
    for b in range(B):
        r = ref['_assign'][b]
        fg = ref['_fg_mask'][b]
        if r is None:
            assert int(asg['num_fg_img'][b]) == 0
            continue
        assert int(asg['num_fg_img'][b]) == r['num_fg']
        assert np.array_equal(asg['matched_valid_idx'][b].cpu()[fg].numpy(), r['matched_gt_inds'].numpy())
        close(asg['pred_iou'][b].cpu()[fg], r['pred_ious'], rtol=1e-6, atol=0)
    losses, d_raw = ops.yolox_loss(od, td, asg, hws, strides)
    want = torch.tensor([float(ref[k]) for k in ('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss', 'num_fg')])
    close(losses, want, rtol=2e-5, atol=1e-6)
    # d_raw is the gradient wrt the raw conv outputs: chain the decode by hand on the reference side
    gref = outr.grad.clone()
    gref[..., 0:2] *= gs[None, :, None]
    gref[..., 2:4] *= outputs[..., 2:4]
    close(d_raw, gref, rtol=1e-4, atol=1e-7)


HEAD_OPTION_CASES = dict(w_obj=dict(bbox_loss_weighting='obj'), w_cls=dict(bbox_loss_weighting='cls'), w_objxcls=dict(bbox_loss_weighting='objxcls'),
                         w_cls_sq=dict(bbox_loss_weighting='cls-w**2'), bg05=dict(ignore_bg_k=0.05), bg30=dict(ignore_bg_k=0.3),
                         w_obj_bg10=dict(bbox_loss_weighting='obj', ignore_bg_k=0.1), w_obj_focal=dict(bbox_loss_weighting='obj', obj_focal_loss=True))


def _head_with_options(strides, nc=2, **kw):
    from leod_amd.models.detection.yolox.models.yolo_head import YOLOXHead
    return YOLOXHead(num_classes=nc, strides=strides, in_channels=(32, 64, 128), **kw)


def _loss_with_options(ops, head, od, td, hws, strides):
    """The option steps in the order of HeadTailFn.forward: ignore thresholds, SimOTA, top-k background, weighted losses."""
    td = head._ignore_bbox(td.clone())
    asg = ops.simota_assign(od, td, hws, strides, ignore_label=float(head.ignore_label))
    if head.ignore_bg_k > 0:
        ops.bg_topk_ignore(od, td, asg, head.ignore_bg_k, ignore_label=float(head.ignore_label))
    losses, d_raw = ops.yolox_loss(od, td, asg, hws, strides, focal=head.obj_focal_loss, label_w=head._bbox_label_weights(td))
    return losses, d_raw, asg


@pytest.mark.parametrize('name', sorted(HEAD_OPTION_CASES))
def test_head_loss_options_golden(ops, golden_dir, name):
    """bbox_loss_weighting / ignore_bg_k (yolo_head.py:335-381) against the losses AND gradients the reference produced (g19); '_ign' is the
    same batch with one ignore box, which takes the reference through get_losses_w_ignore (weights apply, no top-k step)."""
    g = np.load(os.path.join(golden_dir, 'g19_head_options.npz'))
    hws, strides = [(32, 40), (16, 20), (8, 10)], (8, 16, 32)
    gx, gy, gs = oh.make_grids(hws, strides)
    outp = torch.from_numpy(g['outputs'])
    head = _head_with_options(strides, **HEAD_OPTION_CASES[name])
    for suffix, key in (('', 'targets'), ('_ign', 'targets_ign')):
        losses, d_raw, _ = _loss_with_options(ops, head, outp.to(DEV), torch.from_numpy(g[key]).to(DEV), hws, strides)
        close(losses, g[f'{name}{suffix}_losses'], rtol=2e-5, atol=1e-6)
        gref = torch.from_numpy(g[f'{name}{suffix}_grad']).clone()           # reference gradient wrt the decoded boxes -> raw conv outputs
        gref[..., 0:2] *= gs[None, :, None]
        gref[..., 2:4] *= outp[..., 2:4]
        close(d_raw, gref, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize('geom,kfrac,quant', [('gen1', 0.02, 0), ('gen4', 0.25, 0), ('1mpx', 0.1, 0), ('gen1', 0.3, 8), ('1mpx', 0.07, 4)])
def test_bg_topk_ignore_vs_oracle(ops, geom, kfrac, quant):
    """The per-image top-k background mask at the three head sizes against the oracle's topk; ``quant`` rounds the objectness logits to a
    grid so that many anchors tie at the threshold: the mask then has the oracle's COUNT and the same multiset of values (which of the
    tied anchors are taken is unspecified in torch.topk)."""
    B = 4
    hws, strides, (gx, gy, gs), outputs, tg = _gen1_case(B, 11, geom=geom)
    if quant:
        outputs[..., 4] = torch.round(outputs[..., 4] * quant) / quant
    nc = HEAD_GEOMS[geom][2]
    od, td = outputs.to(DEV), tg.to(DEV)
    asg = ops.simota_assign(od, td, hws, strides)
    fg = asg['fg_mask'].cpu().bool()
    assert not bool(asg['ignore_mask'].any())
    ops.bg_topk_ignore(od, td, asg, kfrac)
    got = asg['ignore_mask'].cpu().bool()
    for b in range(B):
        want = oh.highest_score_mask(outputs[b, :, 4:5], kfrac, fg[b])
        assert int(got[b].sum()) == int(want.sum()) > 0
        assert not bool((got[b] & fg[b]).any())
        if quant:
            assert torch.equal(outputs[b, :, 4][got[b]].sort()[0], outputs[b, :, 4][want].sort()[0])
        else:
            assert torch.equal(got[b], want)
    # a batch with an ignore box: the step is skipped altogether
    tg2 = tg.clone()
    tg2[0, 0, 0] = 1024
    asg2 = ops.simota_assign(od, tg2.to(DEV), hws, strides)
    before = asg2['ignore_mask'].clone()
    ops.bg_topk_ignore(od, tg2.to(DEV), asg2, kfrac)
    assert torch.equal(before, asg2['ignore_mask'])
    # full loss with both options at this size against the oracle
    head = _head_with_options(strides, nc=nc, bbox_loss_weighting='objxcls-w**0.5', ignore_bg_k=kfrac)
    g = torch.Generator().manual_seed(5)
    tg3 = tg.clone()
    nz = (tg3.sum(2) > 0).float()
    tg3[:, :, 5] = (0.2 + 0.8 * torch.rand(tg3.shape[:2], generator=g)) * nz
    tg3[:, :, 6] = (0.2 + 0.8 * torch.rand(tg3.shape[:2], generator=g)) * nz
    outr = outputs.clone().requires_grad_(True)
    ref = oh.get_losses(gx, gy, gs, tg3.clone(), outr, num_classes=nc, bbox_loss_weighting='objxcls-w**0.5', ignore_bg_k=kfrac)
    ref['loss'].backward()
    losses, d_raw, _ = _loss_with_options(ops, head, od, tg3.to(DEV), hws, strides)
    want = torch.tensor([float(ref[k]) for k in ('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss', 'num_fg')])
    if not quant:
        close(losses, want, rtol=2e-5, atol=1e-6)
        gref = outr.grad.clone()
        gref[..., 0:2] *= gs[None, :, None]
        gref[..., 2:4] *= outputs[..., 2:4]
        close(d_raw, gref, rtol=1e-4, atol=1e-7)
    else:                                                   # tied picks may differ, their loss contributions do not
        close(losses, want, rtol=2e-5, atol=1e-6)


def test_simota_golden(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, 'g06_simota.npz'))
    hws, strides = [(32, 40), (16, 20), (8, 10)], (8, 16, 32)
    for c in range(3):
        T = lambda k: torch.from_numpy(g[f'c{c}_{k}'])  # noqa
        gt, cls, bp, cl, ol = T('gt'), T('cls'), T('bp'), T('cls_l'), T('obj_l')
        out = torch.cat([bp, ol, cl], 1)[None].contiguous()
        for ign in (False, True):
            lab = torch.zeros((1, gt.shape[0] + 2, 7))
            lab[0, :gt.shape[0], 0] = cls
            lab[0, :gt.shape[0], 1:5] = gt
            lab[0, :gt.shape[0], 5:7] = 1
            pre = 'ig_' if ign else ''
            if ign:
                lab[0, :gt.shape[0], 0] = torch.where(T('valid'), cls, torch.full_like(cls, 1024.))
            asg = ops.simota_assign(out.to(DEV), lab.to(DEV), hws, strides)
            fg = g[f'c{c}_{pre}fg_mask']
            assert np.array_equal(asg['fg_mask'][0].cpu().numpy().astype(bool), fg)
            assert np.array_equal(asg['matched_valid_idx'][0].cpu().numpy()[fg], g[f'c{c}_{pre}matched'])
            assert int(asg['num_fg_img'][0]) == int(g[f'c{c}_{pre}nfg'])
            close(asg['pred_iou'][0].cpu()[torch.from_numpy(fg)], g[f'c{c}_{pre}pious'], rtol=1e-6, atol=0)
            if ign:
                assert np.array_equal(asg['ignore_mask'][0].cpu().numpy().astype(bool), g[f'c{c}_ig_ignore_mask'])
    # full loss vectors of the reference (ignore rows, all-ignore image, empty image, focal)
    tgt, outp = torch.from_numpy(g['ign_targets']), torch.from_numpy(g['ign_outputs'])
    for targets, want, focal in [(tgt, g['ign_losses'], False)]:
        asg = ops.simota_assign(outp.to(DEV), targets.to(DEV), hws, strides)
        losses, _ = ops.yolox_loss(outp.to(DEV), targets.to(DEV), asg, hws, strides, want_grad=False, focal=focal)
        close(losses, want, rtol=2e-5, atol=1e-6)
    tg2 = tgt.clone()
    tg2[:, :, 0] = torch.where(tg2[:, :, 0] == 1024, torch.zeros_like(tg2[:, :, 0]), tg2[:, :, 0])
    asg = ops.simota_assign(outp.to(DEV), tg2.to(DEV), hws, strides)
    close(ops.yolox_loss(outp.to(DEV), tg2.to(DEV), asg, hws, strides, want_grad=False)[0], g['noign_losses'], rtol=2e-5, atol=1e-6)
    close(ops.yolox_loss(outp.to(DEV), tg2.to(DEV), asg, hws, strides, want_grad=False, focal=True)[0], g['focal_losses'], rtol=2e-5, atol=1e-6)
    # self-training head (model=rnndet-soft): pseudo boxes under the per-class confidence thresholds become ignore boxes
    # (yolo_head.py:383-401) before the assignment
    from leod_amd.config import full_config, dynamically_modify_train_config
    from leod_amd.models.detection.yolox_extension.models.build import build_yolox_head
    cfg = dynamically_modify_train_config(full_config('gen1', 'small', 'rnndet-soft')).model
    head = build_yolox_head(cfg.head, in_channels=(96, 192, 384), strides=strides)
    assert list(head.ignore_bbox_thresh) == [0.7, 0.35]
    thr = head._ignore_bbox(torch.from_numpy(g['thr_targets']).clone().to(DEV))
    assert int((thr[:, :, 0] == 1024).sum()) > 0
    asg = ops.simota_assign(outp.to(DEV), thr, hws, strides)
    close(ops.yolox_loss(outp.to(DEV), thr, asg, hws, strides, want_grad=False)[0], g['thr_losses'], rtol=2e-5, atol=1e-6)


def test_focal_grad(ops):
    hws, strides, (gx, gy, gs), outputs, tg = _gen1_case(3, 11)
    outr = outputs.clone().requires_grad_(True)
    ref = oh.get_losses(gx, gy, gs, tg.clone(), outr, num_classes=2, obj_focal_loss=True)
    ref['loss'].backward()
    asg = ops.simota_assign(outputs.to(DEV), tg.to(DEV), hws, strides)
    losses, d_raw = ops.yolox_loss(outputs.to(DEV), tg.to(DEV), asg, hws, strides, focal=True)
    close(losses[0], float(ref['loss']), rtol=2e-5)
    close(d_raw[..., 4], outr.grad[..., 4], rtol=1e-4, atol=1e-8)


def test_head_pred(ops):
    B, h, w, Hd, nc, stride = 3, 8, 10, 96, 2, 32
    A, a0 = 200, 100
    cf, rf = rnd((B, h, w, Hd), 1).requires_grad_(True), rnd((B, h, w, Hd), 2).requires_grad_(True)
    cw, cb = rnd((nc, Hd), 3, 0.1).requires_grad_(True), rnd((nc,), 4).requires_grad_(True)
    rw, rb = rnd((4, Hd), 5, 0.05).requires_grad_(True), rnd((4,), 6, 0.1).requires_grad_(True)
    ow, obb = rnd((1, Hd), 7, 0.1).requires_grad_(True), rnd((1,), 8).requires_grad_(True)
    raw = torch.cat([F.linear(rf, rw, rb), F.linear(rf, ow, obb), F.linear(cf, cw, cb)], -1).reshape(B, h * w, 5 + nc)
    yv, xv = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing='ij')
    grid = torch.stack([xv, yv], -1).reshape(1, -1, 2)
    dec = torch.cat([(raw[..., :2] + grid) * stride, torch.exp(raw[..., 2:4]) * stride, raw[..., 4:]], -1)
    d_raw = torch.zeros((B, A, 5 + nc))
    d_raw[:, a0:a0 + h * w] = rnd((B, h * w, 5 + nc), 9)
    raw.backward(d_raw[:, a0:a0 + h * w])
    ot, oi = torch.zeros((B, A, 5 + nc), device=DEV), torch.zeros((B, A, 5 + nc), device=DEV)
    D = lambda t: t.detach().to(DEV)  # noqa
    ops.head_pred_fwd(D(cf), D(rf), D(cw), D(cb), D(rw), D(rb), D(ow), D(obb), ot, oi, stride, a0)
    close(ot[:, a0:a0 + h * w], dec, rtol=5e-5, atol=1e-5)
    close(oi[:, a0:a0 + h * w, :4], dec[..., :4], rtol=5e-5, atol=1e-5)
    close(oi[:, a0:a0 + h * w, 4:], dec[..., 4:].sigmoid(), rtol=5e-5, atol=1e-6)
    grads = [torch.zeros_like(D(t)) for t in (cw, cb, rw, rb, ow, obb)]
    dcf, drf = ops.head_pred_bwd(d_raw.to(DEV), D(cf), D(rf), D(cw), D(rw), D(ow), *grads, a0)
    close(dcf, cf.grad, rtol=1e-4, atol=1e-6)
    close(drf, rf.grad, rtol=1e-4, atol=1e-6)
    for gg, t in zip(grads, (cw, cb, rw, rb, ow, obb)):
        close(gg, t.grad, rtol=2e-4, atol=1e-4)


# ---------------------------------------------------------------------------------------------------
def _check_dets(det, cnt, ref_list):
    det, cnt = det.cpu(), cnt.cpu()
    assert [int(c) for c in cnt] == [len(r) for r in ref_list]
    for b, r in enumerate(ref_list):
        assert np.array_equal(det[b, :len(r)].numpy(), r.numpy()), f'image {b}'


@pytest.mark.parametrize('name,nc,conf,agn', [
    ('rand_c0.1', 2, 0.1, False), ('rand_c0.01', 2, 0.01, False), ('rand_c0.001', 2, 0.001, False),
    ('rand_agnostic', 2, 0.1, True), ('adv_c0.1', 3, 0.1, False), ('adv_c0.001', 3, 0.001, False),
    ('many_c0.001', 2, 0.001, False), ('none', 2, 0.5, False)])
def test_postprocess_golden(ops, golden_dir, name, nc, conf, agn):
    g = np.load(os.path.join(golden_dir, 'g07_postprocess.npz'))
    pred = torch.from_numpy(g[name + '_pred']).clone()
    p = pred.to(DEV)
    # the golden vectors were recorded with torchvision's CPU rule (per-class loop above 4000 box elements)
    det, cnt = ops.postprocess_nms(p, nc, conf, 0.45, class_agnostic=agn, vanilla_limit=4000)
    n = list(g[name + '_n'])
    assert [int(c) for c in cnt.cpu()] == n
    flat = torch.cat([det[b, :n[b]].cpu() for b in range(len(n))], 0).numpy()
    assert np.array_equal(flat, g[name + '_det'])
    close(p[..., 0].cpu(), pred[..., 0] - pred[..., 2] / 2, rtol=0, atol=0, what='in-place xyxy')


@pytest.mark.parametrize('seed', [0, 1])
@pytest.mark.parametrize('limit', [20000, 4000])
def test_postprocess_full_size(ops, seed, limit):
    """Gen1 size (1680 anchors) x 24 images and Gen4-ds2 size (5040 anchors), both NMS regimes."""
    g = torch.Generator().manual_seed(seed)
    for A, B, nc in [(1680, 24, 2), (5040, 4, 3)]:
        pred = torch.cat([torch.rand(B, A, 2, generator=g) * torch.tensor([300., 230.]), 8 + 40 * torch.rand(B, A, 2, generator=g),
                          torch.rand(B, A, 1, generator=g), torch.rand(B, A, nc, generator=g)], -1)
        ref = op.postprocess(pred.clone(), nc, 0.1, 0.45, pad=torch.zeros((0, 7)),
                             device_semantics='gpu' if limit == 20000 else 'cpu')
        det, cnt = ops.postprocess_nms(pred.to(DEV), nc, 0.1, 0.45, vanilla_limit=limit)
        _check_dets(det, cnt, ref)


def test_postprocess_1mpx_head(ops):
    """20160 anchors (768x1280): the workgroup keeps up to 4096 candidates in LDS; an image with more sorts and suppresses in
    its slice of the global workspace -- exact against the oracle on both sides of that boundary and in both torchvision
    regimes (coordinate trick up to 5000 boxes, per-class loop above).  No candidate limit, like the reference."""
    from leod_amd._lib import LeodHipError, lib
    from leod_amd.ops import host_counts
    g = torch.Generator().manual_seed(3)
    A, B, nc = 20160, 3, 3
    pred = torch.cat([torch.rand(B, A, 2, generator=g) * torch.tensor([1270., 710.]), 8 + 60 * torch.rand(B, A, 2, generator=g),
                      torch.rand(B, A, 1, generator=g), torch.rand(B, A, nc, generator=g)], -1)
    # candidates per image: ~1700 (LDS tier), ~4200 (workspace tier, coordinate trick), ~6300 and ~17200 (per-class regime)
    for conf in (0.75, 0.6, 0.5, 0.1):
        ref = op.postprocess(pred.clone(), nc, conf, 0.45, pad=torch.zeros((0, 7)), device_semantics='gpu')
        det, cnt = ops.postprocess_nms(pred.to(DEV), nc, conf, 0.45)
        assert 0 < int(cnt.min())
        _check_dets(det, cnt, ref)
    # the raw C ABI without a workspace reports the overflow instead of truncating
    p = pred.to(DEV)
    det = torch.empty((B, 4096, 7), device=DEV)
    cnt = torch.empty((B,), dtype=torch.int32, device=DEV)
    rc = lib().leod_postprocess_nms(p.data_ptr(), det.data_ptr(), cnt.data_ptr(), None, B, A, nc, 0.1, 0.45, 0, 4096, 20000,
                                    torch.cuda.current_stream().cuda_stream)
    assert rc == 0 and [int(c) for c in cnt.cpu()] == [-1] * B
    with pytest.raises(LeodHipError):
        host_counts(cnt)
    lab, lcnt = ops.pseudo_filter(det, cnt, 0.5, 0.5, True, (720, 1280))
    with pytest.raises(LeodHipError):
        host_counts(lcnt, 'pred2label')


@pytest.mark.parametrize('geom,n_gt', [('gen1', 150), ('1mpx', 260)])
def test_simota_crowded_frame_beyond_lds_tiers(ops, geom, n_gt):
    """More boxes per frame than the per-gt LDS arrays hold (128) and, on the 1 Mpx head, more candidate anchors than the LDS
    candidate arrays hold (10240): the assignment and the losses come out of the workspace tier, exact against the oracle --
    the reference has no limit on boxes per frame (yolo_head.py:606-700)."""
    (Hp, Wp), frame_hw, nc = HEAD_GEOMS[geom]
    hws, strides = [(Hp // s_, Wp // s_) for s_ in (8, 16, 32)], (8, 16, 32)
    gx, gy, gs = oh.make_grids(hws, strides)
    A, B = gx.numel(), 2
    g = torch.Generator().manual_seed(n_gt)
    outputs = torch.cat([torch.stack([(gx + 0.5) * gs, (gy + 0.5) * gs, gs * 3, gs * 2.5], 1)[None].repeat(B, 1, 1)
                         + torch.randn(B, A, 4, generator=g) * 2, torch.randn(B, A, 1 + nc, generator=g) * 2], -1)
    tg = torch.zeros(B, n_gt, 7)
    for b in range(B):
        n = n_gt - 9 * b
        wh = 12 + 50 * torch.rand(n, 2, generator=g)
        c = wh / 2 + torch.rand(n, 2, generator=g) * (torch.tensor([frame_hw[1], frame_hw[0]]) - 1 - wh)
        tg[b, :n] = torch.cat([torch.randint(0, nc, (n, 1), generator=g).float(), c, wh, torch.ones(n, 2)], 1)
    tg[1, 5, 0] = 1024                                   # one ignore box: the _w_ignore path
    ref = oh.get_losses(gx, gy, gs, tg.clone(), outputs.clone(), num_classes=nc, return_assign=True)
    od, td = outputs.to(DEV), tg.to(DEV)
    asg = ops.simota_assign(od, td, hws, strides)
    assert int(asg['totals'][2]) & 2 == 0
    assert np.array_equal(asg['fg_mask'].cpu().numpy().astype(bool), ref['_fg_mask'].numpy())
    assert np.array_equal(asg['ignore_mask'].cpu().numpy().astype(bool), ref['_ignore_mask'].numpy())
    for b in range(B):
        r, fg = ref['_assign'][b], ref['_fg_mask'][b]
        assert int(asg['num_fg_img'][b]) == r['num_fg'] > n_gt
        assert np.array_equal(asg['matched_valid_idx'][b].cpu()[fg].numpy(), r['matched_gt_inds'].numpy())
    losses, _ = ops.yolox_loss(od, td, asg, hws, strides)
    want = torch.tensor([float(ref[k]) for k in ('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss', 'num_fg')])
    close(losses, want, rtol=2e-5, atol=1e-6)


def test_tta_merge_and_pseudo_filter(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, 'g08_pseudo.npz'))
    views = [torch.from_numpy(g['tta_in0']), torch.from_numpy(g['tta_in1'])]
    A = max(len(v) for v in views)
    pred = torch.zeros((2, A, 7))
    for i, v in enumerate(views):
        pred[i, :len(v)] = v
        pred[i, len(v):, 4] = -1.0          # padding rows can never pass the confidence test
    det, cnt = ops.postprocess_nms(pred.to(DEV), 0, 0.01, 0.45, vanilla_limit=4000)
    n = list(g['tta_n'])[:2]
    assert [int(c) for c in cnt.cpu()] == n
    assert np.array_equal(torch.cat([det[b, :n[b]].cpu() for b in range(2)]).numpy(), g['tta_out'])
    # pred2label
    lens = list(g['p2l_lens_in'])
    allp = torch.from_numpy(g['p2l_in'])
    det = torch.zeros((3, max(lens), 7))
    s = 0
    for i, n_ in enumerate(lens):
        det[i, :n_] = allp[s:s + n_]
        s += n_
    cnt = torch.tensor(lens, dtype=torch.int32)
    lab, lc = ops.pseudo_filter(det.to(DEV), cnt.to(DEV), [0.6, 0.3], [0.6, 0.3], True, (240, 304))
    assert [int(c) for c in lc.cpu()] == list(g['p2l_lens'])
    assert np.array_equal(torch.cat([lab[b, :int(lc[b])].cpu() for b in range(3)]).numpy(), g['p2l_out'])
    lab, lc = ops.pseudo_filter(det.to(DEV), cnt.to(DEV), 0.5, 0.4, False, (240, 304))
    assert [int(c) for c in lc.cpu()] == list(g['p2lf_lens'])
    assert np.array_equal(torch.cat([lab[b, :int(lc[b])].cpu() for b in range(3)]).numpy(), g['p2lf_out'])
    all4 = torch.from_numpy(g['p2l4_in'])
    s = 0
    for i, n_ in enumerate(lens):
        det[i, :n_] = all4[s:s + n_]
        s += n_
    lab, lc = ops.pseudo_filter(det.to(DEV), cnt.to(DEV), [0.3, 0.3, 0.6], [0.3, 0.3, 0.6], True, (360, 640))
    assert [int(c) for c in lc.cpu()] == list(g['p2l4_lens'])
    assert np.array_equal(torch.cat([lab[b, :int(lc[b])].cpu() for b in range(3)]).numpy(), g['p2l4_out'])


def test_voxelize_golden(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, 'g10_voxel.npz'))
    for name, fast, cutoff in [('a', True, None), ('b', False, 10), ('c', True, 3)]:
        ev = [torch.from_numpy(g[f'{name}_{k}'].astype(np.int64)).to(DEV) for k in 'xypt']
        rep = ops.voxelize_u8(*ev, 10, 24, 30, count_cutoff=cutoff, fastmode=fast)
        assert np.array_equal(rep.cpu().numpy(), g[f'{name}_rep'])


@pytest.mark.parametrize('T,B,C,H,W', [(21, 4, 20, 24, 304), (3, 2, 5, 7, 20), (2, 3, 4, 5, 7), (40, 1, 2, 3, 16)])
def test_stack_hflip_u8(ops, T, B, C, H, W):
    """hflip TTA input of the pseudo-label pass (pseudo_labeler.py:469-470) in one launch: 16-byte, 4-byte and scalar row vectors, more
    than 32 frames (two launches) -- exactly torch.cat([stack, flip(stack)], batch)."""
    g = torch.Generator().manual_seed(5)
    frames = [torch.randint(0, 256, (B, C, H, W), generator=g, dtype=torch.uint8).to(DEV) for _ in range(T)]
    out = ops.stack_hflip_u8(frames)
    ev = torch.stack(frames)
    assert torch.equal(out, torch.cat([ev, torch.flip(ev, dims=[-1])], dim=1))


def test_mixed_density_golden(ops, golden_dir):
    """MixedDensityEventStack (data/utils/representations.py:132-221) through the package's class: the outputs recorded from the reference
    (bin edges at exact powers of 1/2, int8 wrap-around, cutoffs 0 / 5 / 127 / none), the oracle on a larger stream, empty input"""
    from leod_amd.data.utils.representations import MixedDensityEventStack, StackedHistogram
    from oracle import postproc as op
    g = np.load(os.path.join(golden_dir, 'g25_mixed_density.npz'))
    for name, bins, cutoff in [('a', 10, None), ('b', 6, 5), ('c', 12, 0), ('e', 8, 127)]:
        ev = [torch.from_numpy(g[f'{name}_{k}'].astype(np.int64)).to(DEV) for k in 'xypt']
        rep = MixedDensityEventStack(bins, 24, 30, count_cutoff=cutoff).construct(*ev)
        assert rep.dtype is torch.int8 and np.array_equal(rep.cpu().numpy(), g[f'{name}_rep']), name
    small = [torch.tensor(a, device=DEV) for a in ([1, 2, 2], [0, 1, 1], [0, 1, 1], [5, 5, 5])]
    assert np.array_equal(MixedDensityEventStack(4, 3, 4).construct(*small).cpu().numpy(), g['d_rep'])
    none = [torch.zeros(0, dtype=torch.int64, device=DEV)] * 4
    assert not MixedDensityEventStack(3, 4, 5).construct(*none).any()
    rng = np.random.RandomState(7)
    n, H, W = 2_000_000, 360, 640
    x, y, p = rng.randint(0, W, n), rng.randint(0, H, n), rng.randint(0, 2, n)
    t = np.sort(rng.randint(0, 50_000, n))
    ev = [torch.from_numpy(a.astype(np.int64)).to(DEV) for a in (x, y, p, t)]
    rep = MixedDensityEventStack(10, H, W, count_cutoff=20).construct(*ev)
    assert np.array_equal(rep.cpu().numpy(), op.mixed_density_stack(x, y, p, t, 10, H, W, count_cutoff=20))
    # the histogram class over the same stream
    hist = StackedHistogram(10, H, W, count_cutoff=10).construct(*ev)
    assert np.array_equal(hist.cpu().numpy(), op.stacked_histogram(x, y, p, t, 10, H, W, count_cutoff=10))


def test_adamw_clip(ops):
    n = 10007
    p0, g0 = rnd((n,), 1), rnd((n,), 2, 2.0)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p], lr=2e-4, weight_decay=0.01)
    pd, m, v = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 4):
        gi = g0 * (1 + 0.1 * step)
        p.grad = gi.clone()
        torch.nn.utils.clip_grad_value_([p], 1.0)
        opt.step()
        ops.adamw_clip_step(pd, gi.to(DEV), m, v, 2e-4, step, weight_decay=0.01, clip_value=1.0)
        close(pd, p.detach(), rtol=1e-6, atol=1e-7)


# ---------------------------------------------------------------------------------------------------
def test_augment_u8_golden(golden_dir):
    """leod_augment_u8 (one gather kernel per batch) == the reference's per-sample flip / interpolate / paste pipeline on the
    recorded augmentation states, batched as [T, B] with a different state per sample."""
    from oracle.synth import synth_augment_sample, AUGMENT_CASES
    from leod_amd.data.utils.augmentor import AugmentationState, ZoomInState, ZoomOutState, augment_events
    g = np.load(os.path.join(golden_dir, 'g14_augment.npz'))
    for hw in ((60, 76), (48, 64)):
        cases = [c for c in AUGMENT_CASES if (c[1], c[2]) == hw]
        evs, states = [], []
        for seed, H, W in cases:
            ev, _ = synth_augment_sample(seed, H, W)
            evs.append(torch.stack(ev))
            s = g[f's{seed}_state']
            states.append(AugmentationState(apply_h_flip=bool(s[0]),
                                            zoom_in=ZoomInState(bool(s[1]), int(s[2]), int(s[3]), float(s[4])),
                                            zoom_out=ZoomOutState(bool(s[5]), int(s[6]), int(s[7]), float(s[8]))))
        batch = torch.stack(evs, 1).contiguous().to(DEV)                   # [T, B, 20, H, W]
        out = augment_events(batch, states).cpu().numpy()
        for b, (seed, _, _) in enumerate(cases):
            np.testing.assert_array_equal(out[:, b], g[f's{seed}_ev'])


@pytest.mark.parametrize('hflip', [False, True])
@pytest.mark.parametrize('mode,x0,y0,factor', [(0, 0, 0, 1.0), (1, 37, 21, 1.37), (1, 0, 0, 1.5), (2, 11, 9, 1.13), (2, 0, 0, 1.2)])
def test_augment_u8_full_size_vs_torch(hflip, mode, x0, y0, factor):
    """Gen1 frame size: the kernel against plain PyTorch (flip + interpolate(nearest-exact) + paste) on the device."""
    from leod_amd.data.utils.augmentor import AugmentationState, ZoomInState, ZoomOutState, augment_events
    T, B, H, W = 3, 2, 240, 304
    g = torch.Generator().manual_seed(11)
    ev = ((torch.rand((T, B, 20, H, W), generator=g) < 0.1) * torch.randint(1, 200, (T, B, 20, H, W), generator=g)).to(torch.uint8).to(DEV)
    st = AugmentationState(apply_h_flip=hflip, zoom_in=ZoomInState(mode == 1, x0, y0, factor if mode == 1 else 1.0),
                           zoom_out=ZoomOutState(mode == 2, x0, y0, factor if mode == 2 else 1.0))
    ident = AugmentationState()
    out = augment_events(ev, [st, ident])
    ref = ev[:, 0].reshape(T * 20, H, W)
    if hflip:
        ref = torch.flip(ref, dims=[-1])
    wh, ww = int(H / factor), int(W / factor)
    if mode == 1:
        ref = F.interpolate(ref[None, :, y0:y0 + wh, x0:x0 + ww].float(), size=(H, W), mode='nearest-exact')[0].to(torch.uint8)
    elif mode == 2:
        win = F.interpolate(ref[None].float(), size=(wh, ww), mode='nearest-exact')[0].to(torch.uint8)
        ref = torch.zeros_like(ref)
        ref[:, y0:y0 + wh, x0:x0 + ww] = win
    assert torch.equal(out[:, 0].reshape(T * 20, H, W), ref)
    assert torch.equal(out[:, 1], ev[:, 1])


def test_augment_u8_time_flip():
    """tflip rides in the same pass: out[t,b,c] = aug(in[T-1-t,b,C-1-c]) (time_flip_data, sequence_base.py:207-227:
    `[x.flip(0) for x in ev_repr[::-1]]`), combined with the per-sample spatial state."""
    from leod_amd.data.utils.augmentor import AugmentationState, ZoomInState, augment_events
    T, B, H, W = 5, 3, 48, 64
    g = torch.Generator().manual_seed(5)
    ev = torch.randint(0, 255, (T, B, 20, H, W), generator=g).to(torch.uint8).to(DEV)
    sts = [AugmentationState(apply_t_flip=True), AugmentationState(apply_h_flip=True),
           AugmentationState(apply_h_flip=True, apply_t_flip=True, zoom_in=ZoomInState(True, 5, 3, 1.25))]
    out = augment_events(ev, sts)
    plain = [AugmentationState(), sts[1], AugmentationState(apply_h_flip=True, zoom_in=ZoomInState(True, 5, 3, 1.25))]
    spatial = augment_events(ev, plain)
    for b in (0, 2):
        assert torch.equal(out[:, b], torch.flip(spatial[:, b], dims=[0, 1]))
    assert torch.equal(out[:, 1], spatial[:, 1])
    assert torch.equal(spatial[:, 0], ev[:, 0])


@pytest.mark.parametrize('M,N,K,with_res', [(40009, 192, 48, True), (20011, 144, 48, True), (33000, 192, 48, False),
                                            (5000, 192, 48, True), (20000, 288, 96, True)])
def test_linear_dgrad_ln_bwd(ops, M, N, K, with_res):
    """x -> LayerNorm -> Linear: gradient wrt x from dy in one launch (stage-1 shapes: dn = dy W stays in registers, the
    LayerNorm backward runs in the epilogue of the row-streaming dgrad) or through the two-kernel fallback -- same result
    as autograd either way, including the LayerNorm weight / bias gradients."""
    x = rnd((M, K), 1).requires_grad_(True)
    lw, lb = (1 + 0.2 * rnd((K,), 2)).requires_grad_(True), (0.1 * rnd((K,), 3)).requires_grad_(True)
    W = rnd((N, K), 4, 0.2)
    dy, dres = rnd((M, N), 5), rnd((M, K), 6)
    F.linear(F.layer_norm(x, (K,), lw, lb, 1e-5), W).backward(dy)
    _, st = ops.layernorm_fwd(x.detach().to(DEV), lw.detach().to(DEV), lb.detach().to(DEV), want_stats=True)
    dw, db = torch.zeros(K, device=DEV), torch.zeros(K, device=DEV)
    dx = ops.linear_dgrad_ln_bwd(dy.to(DEV), W.to(DEV), x.detach().to(DEV), st, lw.detach().to(DEV),
                                 dres.to(DEV) if with_res else None, dw, db)
    close(dx, x.grad + (dres if with_res else 0), rtol=1e-4, atol=1e-5, what='dx')
    close(dw, lw.grad, rtol=3e-4, atol=1e-4, what='d ln weight')
    close(db, lb.grad, rtol=3e-4, atol=1e-4, what='d ln bias')


@pytest.mark.parametrize('B,H,W,Ca,Cb,up', [(2, 16, 20, 192, 192, True), (3, 32, 40, 96, 96, True), (2, 8, 10, 48, 96, False), (1, 6, 8, 4, 12, True),
                                            (5, 16, 20, 96, 192, False)])
def test_cat2_up_matches_torch(ops, B, H, W, Ca, Cb, up):
    """Channel concat of two NHWC maps with the first upsampled x2 (nearest) on the way -- the PAFPN top-down joins and the CSPLayer join
    (reference yolo_pafpn.py:113-123, network_blocks.py:160-166) -- forward and backward against the torch ops it replaces (exact: copies
    and 4-term sums)."""
    from leod_amd import functions as Fn
    a = rnd((B, H // 2, W // 2, Ca) if up else (B, H, W, Ca), 1).to(DEV).requires_grad_(True)
    b = rnd((B, H, W, Cb), 2).to(DEV).requires_grad_(True)
    out = Fn.cat2_nhwc(a, b, up=up)
    a2, b2 = a.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    au = a2[:, :, None, :, None, :].expand(B, H // 2, 2, W // 2, 2, Ca).reshape(B, H, W, Ca) if up else a2
    want = torch.cat([au, b2], -1)
    assert torch.equal(out, want)
    g = rnd(tuple(out.shape), 3).to(DEV)
    out.backward(g)
    want.backward(g)
    assert torch.equal(b.grad, b2.grad)
    assert torch.allclose(a.grad, a2.grad, rtol=0, atol=1e-6)


@pytest.mark.parametrize('N,row,nsel', [(168, (16, 20, 192), 32), (40, (8, 10, 384), 7), (21, (32, 40, 96), 21), (5, (3, 4), 1)])
def test_rows_index_add_matches_torch(ops, N, row, nsel):
    """dst[idx[j]] += src[j] with unique indices (the labelled frames' gradient added into a stage output's gradient, functions.ForkSelectFn --
    the gather of BackboneFeatureSelector, reference modules/utils/detection.py:120-157, differentiated) against torch.index_add_ (exact)."""
    g = torch.Generator().manual_seed(3)
    dst = rnd((N,) + row, 1).to(DEV)
    src = rnd((nsel,) + row, 2).to(DEV)
    idx = torch.randperm(N, generator=g)[:nsel].sort().values.to(DEV)
    want = dst.clone().index_add_(0, idx, src)
    ops.rows_index_add(dst, src, idx)
    assert torch.equal(dst, want)
