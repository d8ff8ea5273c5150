"""The 16-bit precision modes (ops.set_precision; the reference's Lightning precision=16 placement, train.py:236-243): the contraction
tests of tests/test_kernels_gpu.py run again with 16-bit MFMA operands against the SAME fp32 CPU references, at the tolerance stated
there (worst element 3e-2 of the tensor's magnitude, rms error 4x tighter).  Every case runs in BOTH modes: 'bf16' (bf16 operands in both
directions) and '16f' (fp16 operands and fp16 activation rows in the forward pass -- the reference's autocast dtype -- bf16 gradients).
``pytest -m gpu``."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import test_kernels_gpu as tk  # noqa: E402


def A16():
    """dtype of the forward-stored 16-bit activation rows (qkv, attention output) in the mode under test"""
    from leod_amd import ops
    return ops.act16_dtype()


@pytest.fixture(params=['bf16', '16f'])
def bf16_ops(request):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from leod_amd import ops
    prev = ops.set_precision(request.param)
    tk.MODE['bf16'] = True
    tk.MODE['fwd16f'] = request.param == '16f'        # forward tensors: the fp16-sized bound of test_kernels_gpu.F16_FWD_RTOL
    try:
        yield ops
    finally:
        tk.MODE['bf16'] = tk.MODE['fwd16f'] = False
        ops.set_precision(prev)


def test_precision_switch(bf16_ops):
    mode = bf16_ops.get_precision()
    assert mode in ('bf16', '16f') and bf16_ops.is_16bit()
    assert bf16_ops.set_precision('f32') == bf16_ops.PRECISIONS[mode] and bf16_ops.get_precision() == 'f32' and not bf16_ops.is_16bit()
    bf16_ops.set_precision(mode)
    assert bf16_ops.act16_dtype() is (torch.float16 if mode == '16f' else torch.bfloat16)
    # the reference's training.precision key (config/general.yaml): 16 = fp16 autocast -> mode 16f (LEOD_PRECISION, set by conftest, overrides it)
    import os
    env = os.environ.pop('LEOD_PRECISION', None)
    try:
        _precision_keys(bf16_ops)
    finally:
        if env is not None:
            os.environ['LEOD_PRECISION'] = env


def _precision_keys(bf16_ops):
    assert bf16_ops.precision_from_config({'precision': 16}) == '16f' and bf16_ops.precision_from_config({'precision': 'bf16'}) == 'bf16'
    assert bf16_ops.precision_from_config({'precision': 32}) == 'f32' and bf16_ops.precision_from_config(None) == 'f32'


@pytest.mark.parametrize('M,N,K,ln,act', [(200, 144, 48, True, False), (64, 1152, 384, True, False), (77, 64, 16, True, True),
                                          (4096, 256, 64, True, True), (20000, 144, 48, True, False), (17000, 1152, 384, True, False),
                                          (24000, 96, 96, False, True), (9000, 128, 64, True, True), (70001, 144, 48, True, False),
                                          (40003, 288, 96, True, False)])
def test_ln_linear_fwd_bf16(bf16_ops, M, N, K, ln, act):
    tk.test_ln_linear_fwd(bf16_ops, M, N, K, ln, act)


@pytest.mark.parametrize('M,N,K', [(200, 48, 48), (129, 384, 1536), (30000, 48, 192), (17000, 384, 1536), (9001, 64, 256)])
def test_linear_lsres_fwd_bf16(bf16_ops, M, N, K):
    tk.test_linear_lsres_fwd(bf16_ops, M, N, K)


@pytest.mark.parametrize('M,N,K', [(300, 144, 48), (100, 1536, 384), (30000, 192, 48), (30000, 48, 192), (20000, 288, 96), (65000, 48, 48),
                                   (9000, 1536, 384), (40007, 48, 192), (40009, 192, 48), (24001, 64, 256)])
def test_linear_backward_bf16(bf16_ops, M, N, K):
    tk.test_linear_backward(bf16_ops, M, N, K)


@pytest.mark.parametrize('B,H,W,Cin,N,ks,stride', [(2, 16, 24, 16, 32, 3, 2), (3, 8, 12, 32, 32, 3, 1), (2, 16, 20, 192, 96, 1, 1),
                                                   (8, 64, 80, 48, 96, 3, 2), (16, 32, 40, 96, 96, 3, 1), (32, 16, 20, 192, 192, 1, 1)])
def test_conv_nhwc_bf16(bf16_ops, B, H, W, Cin, N, ks, stride):
    tk.test_conv_nhwc(bf16_ops, B, H, W, Cin, N, ks, stride)


@pytest.mark.parametrize('M,C,state', [(160, 32, True), (70, 48, True), (640, 384, True), (40960, 48, True), (10240, 96, False), (9000, 192, True)])
def test_convlstm_bf16(bf16_ops, M, C, state):
    tk.test_convlstm(bf16_ops, M, C, state)


@pytest.mark.parametrize('B,H,W,Cin,N', [(8, 32, 40, 96, 96),      # head / PAFPN level 0: 4 rows x 40 pixels per workgroup
                                         (4, 16, 20, 96, 96),      # level 1: 8 rows x 20
                                         (3, 8, 10, 192, 192),     # level 2: one workgroup per image and 96-channel slab
                                         (2, 16, 20, 192, 192),    # LDS-limited row count
                                         (2, 32, 40, 48, 48), (2, 9, 12, 48, 96), (1, 5, 7, 96, 48), (2, 24, 80, 96, 96), (70, 9, 12, 96, 96), (1, 5, 7, 96, 96),
                                         # RVT-B widths; 160-pixel rows of 128 channels are cut into two column segments
                                         (2, 24, 40, 128, 128), (1, 48, 80, 128, 128), (1, 12, 160, 128, 128), (2, 24, 40, 256, 256), (1, 13, 160, 64, 64), (2, 7, 12, 128, 128)])
def test_conv3x3_direct_bf16(bf16_ops, B, H, W, Cin, N):
    """The direct 3x3 / stride-1 convolution of csrc/k_conv3.hip (forward with BatchNorm statistics, and as dgrad) against
    torch's fp32 conv2d on the CPU; ragged last row block (H = 9, 5), widths that are no multiple of 16."""
    import torch.nn.functional as F
    ops = bf16_ops
    x = tk.rnd((B, Cin, H, W), 1).requires_grad_(True)
    w = tk.rnd((N, Cin, 3, 3), 2, 0.05).requires_grad_(True)
    ref = F.conv2d(x, w, None, padding=1)
    dy = tk.rnd(ref.shape, 3)
    ref.backward(dy)
    xn = x.detach().permute(0, 2, 3, 1).contiguous().to(tk.DEV)
    R = 8
    cs = torch.zeros((R, 2, N), dtype=torch.float64, device=tk.DEV)
    y = ops.conv_nhwc_fwd(xn, w.detach().to(tk.DEV), None, colstats=cs)
    tk.close(y, ref.detach().permute(0, 2, 3, 1), what='conv3x3 fwd', fwd=True)
    # the statistics are sums of the kernel's own outputs
    yd = y.double().reshape(-1, N)
    # (fp32 partial sums of <= 12 rows per lane, double beyond)
    assert torch.allclose(cs.sum(0)[0], yd.sum(0), rtol=1e-4, atol=1e-4 * yd.abs().sum(0).max().item())
    assert torch.allclose(cs.sum(0)[1], (yd * yd).sum(0), rtol=1e-4)
    dyn = dy.permute(0, 2, 3, 1).contiguous().to(tk.DEV)
    dx = ops.conv_nhwc_dgrad(dyn, w.detach().to(tk.DEV), xn.shape)
    tk.close(dx, x.grad.permute(0, 2, 3, 1), what='conv3x3 dgrad')
    dx2 = ops.conv_nhwc_dgrad(dyn, w.detach().to(tk.DEV), xn.shape, out=dx.clone(), accumulate=True)
    tk.close(dx2, 2 * x.grad.permute(0, 2, 3, 1), what='conv3x3 dgrad accumulate')
    # weight gradient (direct kernel for 96 -> 96, the im2col kernel otherwise); it accumulates: two calls = twice the gradient
    dw = torch.zeros_like(w.detach(), device=tk.DEV)
    ops.conv_nhwc_wgrad(dyn, xn, dw, None)
    tk.close(dw, w.grad, what='conv3x3 wgrad')
    ops.conv_nhwc_wgrad(dyn, xn, dw, None)
    tk.close(dw, 2 * w.grad, what='conv3x3 wgrad accumulates')


@pytest.mark.parametrize('M,C', [(13440, 384), (53760, 192), (21120, 512), (8256, 96)])
def test_linear_wgrad_group_equals_single_calls(bf16_ops, M, C):
    """The four weight gradients of an attention block (qkv and fc1 behind LayerNorm with bf16 gradient rows, proj from 16-bit attention rows,
    fc2 through GELU of the fp16 pre-activation; maxvit.py:110-118,252-270) as ONE grouped launch of the LDS-DMA kernel
    (leod_linear_wgrad_group) against the four single calls; a second call accumulates."""
    ops = bf16_ops
    d = lambda t: t.to(tk.DEV)  # noqa
    x, y = d(tk.rnd((M, C), 1)), d(tk.rnd((M, C), 2))
    lw1, lb1, lw2, lb2 = (d(1 + 0.2 * tk.rnd((C,), 3)), d(0.1 * tk.rnd((C,), 4)), d(1 + 0.2 * tk.rnd((C,), 5)), d(0.1 * tk.rnd((C,), 6)))
    _, st1 = ops.layernorm_fwd(x, lw1, lb1, want_stats=True)
    _, st2 = ops.layernorm_fwd(y, lw2, lb2, want_stats=True)
    dqkv, du = d(tk.rnd((M, 3 * C), 7)).to(torch.bfloat16), d(tk.rnd((M, 4 * C), 8)).to(torch.bfloat16)
    dz, dy = d(tk.rnd((M, C), 9)), d(tk.rnd((M, C), 10))
    o16, u16 = d(tk.rnd((M, C), 11)).to(A16()), d(tk.rnd((M, 4 * C), 12)).to(torch.float16)
    shapes = [(C, 4 * C), (4 * C, C), (C, C), (3 * C, C)]

    def bufs():
        return [torch.zeros(s_, device=tk.DEV) for s_ in shapes], [torch.zeros(s_[0], device=tk.DEV) for s_ in shapes]

    (W2, W1, Wp, Wq), (b2, b1, bp, bq) = bufs()
    probs = [dict(dy=dz, x=u16, dW=W2, dbias=b2, gelu=True), dict(dy=du, x=y, dW=W1, dbias=b1, stats=st2, ln_w=lw2, ln_b=lb2),
             dict(dy=dy, x=o16, dW=Wp, dbias=bp), dict(dy=dqkv, x=x, dW=Wq, dbias=bq, stats=st1, ln_w=lw1, ln_b=lb1)]
    assert ops.linear_wgrad_group(probs)
    (S2, S1, Sp, Sq), (c2, c1, cp, cq) = bufs()
    ops.linear_wgrad(dz, u16, S2, c2)
    ops.linear_wgrad(du, y, S1, c1, stats=st2, ln_w=lw2, ln_b=lb2)
    ops.linear_wgrad(dy, o16, Sp, cp, x_gelu=False)
    ops.linear_wgrad(dqkv, x, Sq, cq, stats=st1, ln_w=lw1, ln_b=lb1)
    for name, g, r in (('fc2', W2, S2), ('fc1', W1, S1), ('proj', Wp, Sp), ('qkv', Wq, Sq), ('b fc2', b2, c2), ('b fc1', b1, c1), ('b proj', bp, cp),
                       ('b qkv', bq, cq)):
        tk.close(g, r, what=name + ' grouped vs single')
    assert ops.linear_wgrad_group(probs)
    tk.close(W1, 2 * S1, what='fc1 accumulates')
    tk.close(bq, 2 * cq, what='qkv bias accumulates')


@pytest.mark.parametrize('C,sizes', [(96, [(4, 32, 40), (4, 16, 20), (4, 8, 10), (4, 32, 40), (4, 16, 20), (4, 8, 10)]),     # the six tower convs of a depth
                                     (128, [(2, 12, 160), (2, 24, 40)]), (48, [(3, 9, 12), (1, 5, 7), (2, 32, 40)])])
def test_conv3x3_group_equals_single_calls(bf16_ops, C, sizes):
    """n problems of one channel geometry in ONE launch (csrc/k_conv3.hip: problem tables; leod_conv3x3_group_fwd / _dgrad / _wgrad,
    leod_bn_silu_*_group) against the same problems launched one by one: forward outputs, BatchNorm statistics and input gradients
    bit-identical (same arithmetic per output), weight gradients to the order of the workers' partial sums; a second round of input
    gradients accumulates."""
    ops = bf16_ops
    n = len(sizes)
    xs = [tk.rnd((B, H, W, C), 10 + k).to(tk.DEV) for k, (B, H, W) in enumerate(sizes)]
    ws = [tk.rnd((C, C, 3, 3), 30 + k, 0.05).to(tk.DEV) for k in range(n)]
    dys = [tk.rnd((B, H, W, C), 50 + k).to(tk.DEV) for k, (B, H, W) in enumerate(sizes)]
    R = 4
    cs_g = [torch.zeros((R, 2, C), dtype=torch.float64, device=tk.DEV) for _ in range(n)]
    cs_s = [torch.zeros((R, 2, C), dtype=torch.float64, device=tk.DEV) for _ in range(n)]
    assert ops.conv3x3_group_ok(xs, ws)
    ys = ops.conv3x3_group_fwd(xs, ws, cs_g)
    assert ys is not None
    for k in range(n):
        y1 = ops.conv_nhwc_fwd(xs[k], ws[k], None, colstats=cs_s[k])
        assert torch.equal(ys[k], y1), f'forward {k}'
        assert torch.allclose(cs_g[k].sum(0), cs_s[k].sum(0), rtol=1e-12, atol=1e-9), f'statistics {k}'
    outs = [torch.empty_like(x) for x in xs]
    assert ops.conv3x3_group_dgrad(dys, ws, [tuple(x.shape) for x in xs], outs, [False] * n)
    singles = [ops.conv_nhwc_dgrad(dys[k], ws[k], tuple(xs[k].shape)) for k in range(n)]
    for k in range(n):
        assert torch.equal(outs[k], singles[k]), f'dgrad {k}'
    assert ops.conv3x3_group_dgrad(dys, ws, [tuple(x.shape) for x in xs], outs, [True] * n)
    for k in range(n):
        tk.close(outs[k], 2 * singles[k], rtol=1e-6, atol=1e-6, what=f'dgrad accumulate {k}')
    dws_g = [torch.zeros_like(w) for w in ws]
    dws_s = [torch.zeros_like(w) for w in ws]
    if ops.conv3x3_group_wgrad(dys, xs, dws_g):
        for k in range(n):
            ops.conv_nhwc_wgrad(dys[k], xs[k], dws_s[k], None)
            tk.close(dws_g[k], dws_s[k], rtol=2e-5, atol=2e-5 * float(dws_s[k].abs().max()), what=f'wgrad {k}')
    else:       # refused as a whole (the direct weight-gradient kernel has no 48 -> 48 / stride-1 form): nothing was written
        assert C == 48 and all(not d.any() for d in dws_g)
    # BatchNorm + SiLU of the same maps, grouped against single launches
    bw = [1 + 0.2 * tk.rnd((C,), 70 + k).to(tk.DEV) for k in range(n)]
    bb = [0.1 * tk.rnd((C,), 80 + k).to(tk.DEV) for k in range(n)]
    counts = [y.numel() // C for y in ys]
    rm = [torch.zeros(C, device=tk.DEV) for _ in range(2 * n)]
    rv = [torch.ones(C, device=tk.DEV) for _ in range(2 * n)]
    res = ops.bn_silu_fwd_group(ys, cs_g, bw, bb, rm[:n], rv[:n], counts, 1e-5, [0.1] * n)
    for k in range(n):
        a, mean, rstd = ops.bn_silu_fwd(ys[k], cs_s[k], bw[k], bb[k], rm[n + k], rv[n + k], counts[k], eps=1e-5, momentum=0.1)
        assert torch.equal(res[k][0], a) and torch.equal(res[k][1], mean) and torch.equal(res[k][2], rstd), f'bn forward {k}'
        assert torch.equal(rm[k], rm[n + k]) and torch.equal(rv[k], rv[n + k])
    sums_g = [torch.zeros((R, 2, C), dtype=torch.float64, device=tk.DEV) for _ in range(n)]
    means, rstds = [r[1] for r in res], [r[2] for r in res]
    ops.bn_silu_bwd_reduce_group(dys, ys, means, rstds, bw, bb, sums_g)
    dgw = [torch.zeros(C, device=tk.DEV) for _ in range(2 * n)]
    dgb = [torch.zeros(C, device=tk.DEV) for _ in range(2 * n)]
    dzs = ops.bn_silu_bwd_apply_group(dys, ys, means, rstds, bw, bb, sums_g, dgw[:n], dgb[:n], counts)
    for k in range(n):
        s1 = torch.zeros((R, 2, C), dtype=torch.float64, device=tk.DEV)
        ops.bn_silu_bwd_reduce(dys[k], ys[k], means[k], rstds[k], bw[k], bb[k], out=s1)
        tk.close(sums_g[k].sum(0).float(), s1.sum(0).float(), rtol=1e-5, atol=1e-5 * float(s1.sum(0).abs().max()), what=f'bn backward sums {k}')
        dz1 = ops.bn_silu_bwd_apply(dys[k], ys[k], means[k], rstds[k], bw[k], bb[k], s1, dgw[n + k], dgb[n + k], counts[k])
        tk.close(dzs[k], dz1, rtol=1e-5, atol=1e-5, what=f'bn backward dz {k}')
        tk.close(dgw[k], dgw[n + k], rtol=1e-5, atol=1e-5 * float(dgw[n + k].abs().max()), what=f'bn dgamma {k}')


@pytest.mark.parametrize('B,H,W,Cin,N,stride', [(5, 64, 80, 48, 96, 2),      # backbone downsampling, stage 2: 48-channel slice, pixel steps dealt to wave pairs
                                                (3, 32, 40, 96, 192, 2),     # stage 3: two output-channel slices
                                                (9, 16, 20, 192, 384, 2),    # stage 4: 4 x 2 slices, one region per image
                                                (4, 32, 40, 96, 96, 2),      # PAFPN bottom-up conv
                                                (2, 16, 20, 192, 192, 2), (3, 16, 20, 192, 192, 1),       # 192 -> 192, both strides
                                                (2, 10, 12, 48, 96, 2), (1, 6, 8, 96, 96, 2), (70, 8, 12, 96, 96, 2),   # ragged regions, many regions per worker
                                                (2, 18, 28, 96, 192, 1),
                                                (2, 24, 320, 64, 128, 2), (2, 48, 160, 128, 256, 2), (3, 24, 80, 256, 512, 2), (2, 48, 160, 128, 128, 2), (2, 24, 80, 256, 256, 2)])  # RVT-B
def test_conv3x3_strided_sliced_direct_bf16(bf16_ops, B, H, W, Cin, N, stride):
    """The direct kernels of csrc/k_conv3.hip for stride 2 and sliced channel planes -- forward (conv3s1_kernel<.., S = 2>), input
    gradient (conv3s2_dgrad_kernel: four parity classes), weight gradient (conv3_wgrad9_kernel / conv3_wgrad_kernel) -- against torch's
    fp32 conv2d on the CPU."""
    import torch.nn.functional as F
    ops = bf16_ops
    from leod_amd import _lib
    assert _lib.lib().leod_conv_nhwc_wgrad_workspace_floats(B, H, W, Cin, N, 3, stride, 1, 0) > 0, 'shape must take the direct kernel'
    x = tk.rnd((B, Cin, H, W), 1).requires_grad_(True)
    w = tk.rnd((N, Cin, 3, 3), 2, 0.05).requires_grad_(True)
    ref = F.conv2d(x, w, None, stride=stride, padding=1)
    dy = tk.rnd(ref.shape, 3)
    ref.backward(dy)
    xn = x.detach().permute(0, 2, 3, 1).contiguous().to(tk.DEV)
    dyn = dy.permute(0, 2, 3, 1).contiguous().to(tk.DEV)
    # forward (stride 2, <= 96 input channels: conv3s1_kernel<.., S = 2>) with BatchNorm statistics, dgrad (stride 2:
    # conv3s2_dgrad_kernel), plain and accumulating
    R = 8
    cs = torch.zeros((R, 2, N), dtype=torch.float64, device=tk.DEV)
    y = ops.conv_nhwc_fwd(xn, w.detach().to(tk.DEV), None, stride=stride, colstats=cs)
    tk.close(y, ref.detach().permute(0, 2, 3, 1), what='conv3x3 fwd (strided)', fwd=True)
    yd = y.double().reshape(-1, N)
    assert torch.allclose(cs.sum(0)[0], yd.sum(0), rtol=1e-4, atol=1e-4 * yd.abs().sum(0).max().item())
    assert torch.allclose(cs.sum(0)[1], (yd * yd).sum(0), rtol=1e-4)
    dx = ops.conv_nhwc_dgrad(dyn, w.detach().to(tk.DEV), xn.shape, stride=stride)
    tk.close(dx, x.grad.permute(0, 2, 3, 1), what='conv3x3 dgrad (strided)')
    dx2 = ops.conv_nhwc_dgrad(dyn, w.detach().to(tk.DEV), xn.shape, stride=stride, out=dx.clone(), accumulate=True)
    tk.close(dx2, 2 * x.grad.permute(0, 2, 3, 1), what='conv3x3 dgrad (strided) accumulate')
    dw = torch.zeros_like(w.detach(), device=tk.DEV)
    ops.conv_nhwc_wgrad(dyn, xn, dw, None, stride=stride)
    tk.close(dw, w.grad, what='conv3x3 wgrad (direct)')
    ops.conv_nhwc_wgrad(dyn, xn, dw, None, stride=stride)
    tk.close(dw, 2 * w.grad, what='conv3x3 wgrad (direct) accumulates')


@pytest.mark.parametrize('B,H,W,C,heads,part', [(3, 16, 20, 48, 2, (8, 10)), (2, 32, 40, 96, 4, (8, 10)), (1, 8, 10, 384, 16, (8, 10)),
                                                # the other instantiations of the bf16-tile kernels: d = 32, one-head workgroups, padded
                                                # partitions (60 of 64, 56 of 64 tokens), 240-token partitions of the 1 Mpx geometry
                                                (2, 16, 20, 64, 2, (8, 10)), (1, 12, 20, 32, 1, (6, 10)), (2, 12, 20, 96, 4, (6, 10)),
                                                (1, 16, 16, 24, 1, (8, 8)), (1, 14, 16, 48, 2, (7, 8)), (1, 24, 40, 64, 2, (12, 20)),
                                                (1, 24, 40, 72, 3, (12, 20))])
@pytest.mark.parametrize('window', [True, False])
def test_partition_attn_bf16_tensors(bf16_ops, B, H, W, C, heads, part, window):
    """qkv handed over as bf16 and dqkv produced as bf16 (precision mode bf16, LDS attention kernels): q, k, v only ever enter bf16
    MFMAs, so the results are those of the fp32-tensor path on the same (rounded) values -- checked against the fp32 CPU attention."""
    ops = bf16_ops
    assert ops.partition_attn_16bit_ok(B, H, W, C, heads, part)
    qkv16 = tk.rnd((B, H, W, 3 * C), 7).to(A16())
    qkv = qkv16.float().requires_grad_(True)
    ref = tk._attn_ref(qkv, heads, part, window)
    dout = tk.rnd(ref.shape, 8)
    ref.backward(dout)
    q = qkv16.to(tk.DEV)
    out, lse = ops.partition_attn_fwd(q, heads, part, window, want_lse=True)
    tk.close(out, ref, what='attn fwd from bf16 qkv', fwd=True)
    dq = ops.partition_attn_bwd(q, dout.to(tk.DEV), lse, heads, part, window)
    assert dq.dtype is torch.bfloat16
    tk.close(dq.float(), qkv.grad, what='bf16 dqkv')
    # the attention output written as bf16 rows / its gradient read as bf16 rows (bit 1 of qkv_bf16): the same values, rounded once
    o16, lse2 = ops.partition_attn_fwd(q, heads, part, window, want_lse=True, out_bf16=True)
    assert o16.dtype is A16() and torch.equal(lse2, lse)
    assert torch.equal(o16, out.to(A16())), '16-bit O = the fp32 O of the same kernel, rounded to nearest even'
    do16 = dout.to(torch.bfloat16)
    qkv2 = qkv16.float().requires_grad_(True)
    tk._attn_ref(qkv2, heads, part, window).backward(do16.float())
    dq2 = ops.partition_attn_bwd(q, do16.to(tk.DEV), lse, heads, part, window)
    tk.close(dq2.float(), qkv2.grad, what='bf16 dqkv from bf16 dO')


@pytest.mark.parametrize('B,H,W,C,heads', [(21, 32, 40, 48, 2), (8, 32, 40, 96, 4), (28, 16, 20, 192, 8)])
def test_attention_block_keeps_o_and_do_as_bf16(bf16_ops, B, H, W, C, heads):
    """Where leod_attn_block_o16_ok holds, the other consumers of O / dO take bf16 rows: proj + LayerScale + residual from bf16 O, the
    dgrad of proj writing bf16 dO, and the proj weight gradient from bf16 O (maxvit.py:185-270) -- against fp32 CPU arithmetic on the
    same rounded values."""
    import torch.nn.functional as F
    ops = bf16_ops
    assert ops.attn_block_o16_ok(B, H, W, C, heads, (8, 10))
    M = B * H * W
    o16 = tk.rnd((M, C), 1).to(A16())
    res, dy = tk.rnd((M, C), 2), tk.rnd((M, C), 3)
    Wp, bp, g = tk.rnd((C, C), 4, 0.2), tk.rnd((C,), 5, 0.2), 0.5 + 0.1 * tk.rnd((C,), 6)
    d = lambda t: t.detach().to(tk.DEV)  # noqa
    y, _ = ops.linear_lsres_fwd(d(o16), d(Wp), d(bp), d(g), d(res), want_t=False, a_gelu=False)
    tk.close(y, res + g * F.linear(o16.float(), Wp, bp), what='proj + LayerScale + residual from bf16 O', fwd=True)
    do = ops.linear_dgrad(d(dy), d(Wp), kscale=d(g), out_bf16=True)
    assert do.dtype is torch.bfloat16
    tk.close(do.float(), (dy * g) @ Wp, what='bf16 dO')
    dW, db = torch.zeros((C, C), device=tk.DEV), torch.zeros((C,), device=tk.DEV)
    ops.linear_wgrad(d(dy), d(o16), dW, db, x_gelu=False)
    tk.close(dW, (dy.double().t() @ o16.double()).float(), what='proj weight gradient from bf16 O')
    tk.close(db, dy.double().sum(0).float(), what='proj bias gradient')


@pytest.mark.parametrize('M,N,K', [(40009, 144, 48), (20011, 288, 96),
                                   (9001, 576, 192), (5003, 1152, 384), (8200, 192, 64), (12000, 96, 32)])   # generic 16-bit epilogue (stages 3-4, RVT-B / T)
def test_ln_qkv_bf16_rows(bf16_ops, M, N, K):
    import torch.nn.functional as F
    ops = bf16_ops
    x, lw, lb = tk.rnd((M, K), 1), 1 + 0.2 * tk.rnd((K,), 2), 0.1 * tk.rnd((K,), 3)
    W, b = tk.rnd((N, K), 4, 0.2), tk.rnd((N,), 5, 0.2)
    ref = F.linear(F.layer_norm(x, (K,), lw, lb, 1e-5), W, b)
    d = lambda t: t.to(tk.DEV)  # noqa
    o16, _, st = ops.ln_linear_fwd(d(x), d(lw), d(lb), d(W), d(b), want_stats=True, out_bf16=True)
    assert o16.dtype is A16() and st is not None
    tk.close(o16.float(), ref, what='bf16 qkv rows', fwd=True)
    mean = x.mean(1)
    tk.close(st[:, 0], mean, rtol=1e-4, atol=1e-5, what='LayerNorm mean')


def test_qkv_bf16_rows_without_layernorm(bf16_ops):
    """First attention block of a stage (norm1 = Identity, maxvit_rnn.py:164-166): the qkv rows still go out as bf16."""
    import torch.nn.functional as F
    ops = bf16_ops
    for M, N, K in ((9001, 576, 192), (20011, 288, 96), (40009, 144, 48)):
        x, W, b = tk.rnd((M, K), 1), tk.rnd((N, K), 4, 0.2), tk.rnd((N,), 5, 0.2)
        o16, _, st = ops.ln_linear_fwd(x.to(tk.DEV), None, None, W.to(tk.DEV), b.to(tk.DEV), out_bf16=True)
        assert o16.dtype is A16() and st is None
        tk.close(o16.float(), F.linear(x, W, b), what=f'bf16 qkv rows without LayerNorm {M}x{N}x{K}', fwd=True)


@pytest.mark.parametrize('M,C', [(40009, 48), (20011, 96), (16384, 64), (9001, 192), (5003, 384), (4100, 128)])
def test_mlp_hidden_stored_once_as_fp16(bf16_ops, M, C):
    """Precision mode bf16 (stages 1-2 on the row-streaming kernels, everything else on the LDS-staged / wide-tile GEMMs with a 16-bit row
    epilogue): norm2 -> fc1 keeps the hidden pre-activation u once, as fp16 (no fp32 u, no gelu(u) copy);
    fc2 + LayerScale + residual, the dgrad through GELU and the fc2 weight gradient evaluate GELU / GELU' while loading it
    (maxvit.py:110-118, 268-269).  Against the fp32 CPU arithmetic of the same chain, ragged row counts."""
    import torch.nn.functional as F
    ops = bf16_ops
    x, res = tk.rnd((M, C), 1), tk.rnd((M, C), 2)
    lw, lb = 1 + 0.2 * tk.rnd((C,), 3), 0.1 * tk.rnd((C,), 4)
    W1, b1 = tk.rnd((4 * C, C), 5, 0.2), tk.rnd((4 * C,), 6, 0.2)
    W2, b2, g = tk.rnd((C, 4 * C), 7, 0.1), tk.rnd((C,), 8, 0.1), 0.5 + 0.1 * tk.rnd((C,), 9)
    u = F.linear(F.layer_norm(x, (C,), lw, lb, 1e-5), W1, b1).requires_grad_(True)
    h = F.gelu(u)
    W2r = W2.clone().requires_grad_(True)
    z = res + g * F.linear(h, W2r, b2)
    dz = tk.rnd((M, C), 10)
    z.backward(dz)
    d = lambda t: t.detach().to(tk.DEV)  # noqa
    u16, hh, st = ops.ln_linear_fwd(d(x), d(lw), d(lb), d(W1), d(b1), want_act=True, want_stats=True)
    assert u16.dtype is torch.float16 and hh is None, 'the row-streaming shapes keep one fp16 tensor'
    tk.close(u16.float(), u, what='u (fp16)', fwd=True)
    zz, _ = ops.linear_lsres_fwd(u16, d(W2), d(b2), d(g), d(res), want_t=False)
    tk.close(zz, z, what='fc2 + LayerScale + residual from the fp16 pre-activation', fwd=True)
    du = ops.linear_dgrad(d(dz), d(W2), kscale=d(g), aux_u=u16)
    assert du.dtype is (torch.bfloat16 if ops.BF16_GRADS else torch.float32)
    tk.close(du.float(), u.grad, what='dgrad through GELU')
    # the bf16 gradient of the hidden feeds the dgrad of fc1 (+ LayerNorm backward) and the fc1 weight gradient as is
    xr = x.clone().requires_grad_(True)
    W1r, b1r = W1.clone().requires_grad_(True), b1.clone().requires_grad_(True)
    lwr, lbr = lw.clone().requires_grad_(True), lb.clone().requires_grad_(True)
    dures = tk.rnd((M, C), 11)
    (F.linear(F.layer_norm(xr, (C,), lwr, lbr, 1e-5), W1r, b1r) * du.float().cpu()).sum().add((xr * dures).sum()).backward()
    dlw, dlb = torch.zeros((C,), device=tk.DEV), torch.zeros((C,), device=tk.DEV)
    dxx = ops.linear_dgrad_ln_bwd(du, d(W1), d(x), st, d(lw), d(dures), dlw, dlb)
    tk.close(dxx, xr.grad, what='dgrad of fc1 + LayerNorm backward from bf16 du')
    tk.close(dlw, lwr.grad, what='LayerNorm weight gradient')
    tk.close(dlb, lbr.grad, what='LayerNorm bias gradient')
    dW1, db1 = torch.zeros((4 * C, C), device=tk.DEV), torch.zeros((4 * C,), device=tk.DEV)
    ops.linear_wgrad(du, d(x), dW1, db1, stats=st, ln_w=d(lw), ln_b=d(lb))
    tk.close(dW1, W1r.grad, what='fc1 weight gradient from bf16 du')
    tk.close(db1, b1r.grad, what='fc1 bias gradient from bf16 du')
    dW, db = torch.zeros((C, 4 * C), device=tk.DEV), torch.zeros((C,), device=tk.DEV)
    ops.linear_wgrad(d(dz) * d(g), u16, dW, db)
    tk.close(dW, W2r.grad, what='fc2 weight gradient')
    tk.close(db, (dz * g).sum(0), what='fc2 bias gradient')


# (tile of wgrad_bf16.hpp, dY format, X mode): every instantiation of the wide weight-gradient kernel, ragged row counts, a second launch
# accumulating on the first, and column counts below the tile width
@pytest.mark.parametrize('M,N,K,dy16,xmode', [
    (8200, 48, 48, False, 'rows'), (40009, 48, 48, False, 'rows'),            # 48 x 48: proj, stage 1
    (8193, 144, 48, True, 'ln'), (40009, 192, 48, True, 'ln'),                # 192 x 48: qkv / fc1, stage 1
    (8200, 48, 192, False, 'gelu16'), (33001, 32, 160, False, 'gelu16'),      # 48 x 192: fc2, stage 1
    (20011, 96, 96, False, 'rows'), (9001, 192, 192, False, 'rows'),          # 96 x 96 tiles: proj
    (20011, 288, 96, True, 'ln'), (8705, 768, 192, True, 'ln'),               # qkv / fc1
    (12000, 96, 384, False, 'gelu16'), (8300, 192, 768, False, 'gelu16'),     # fc2
    (20011, 384, 192, True, 'concat'), (8705, 768, 384, True, 'concat'),      # ConvLSTM 1x1 on [x | h] with bf16 gate gradients
    (20011, 384, 192, False, 'concat'),
    (40009, 48, 48, False, 'rows16'), (20011, 96, 96, False, 'rows16'), (9001, 192, 192, False, 'rows16'),    # proj from the bf16 attention output
    # 192 x 96 / 96 x 192 workgroup tiles (round 4: six tiles per wave along the 16-bit operand), stage-4 and ragged shapes
    (13440, 1536, 384, True, 'ln'), (8705, 1152, 384, True, 'ln'), (8707, 384, 96, True, 'ln'), (13440, 384, 1536, False, 'gelu16'),
    (8705, 1536, 768, True, 'concat'), (8201, 384, 384, False, 'rows16'),
    # the LDS-DMA kernel of the short row ranges (wgrad_dma.hpp, round 6: rows % 64 == 0, <= 60 k, N and K multiples of 96): every operand
    # preparation mode, bf16 operands used in place, one chunk per workgroup up to long chunk streams
    (13440, 384, 384, False, 'rows16'), (13440, 1536, 768, True, 'concat'), (53760, 576, 192, True, 'ln'), (53760, 192, 192, False, 'rows'),
    (8256, 288, 96, True, 'ln'), (53760, 192, 768, False, 'gelu16'), (13440, 768, 384, False, 'concat'), (8192, 96, 96, True, 'rows'),
    (13440, 1152, 384, True, 'rows16'),
    # the same kernel on 128 x 128 tiles (two ring slots) and 64 x 64 tiles: the power-of-two widths of RVT-B, up to 400 k rows
    (21120, 2048, 512, True, 'ln'), (21120, 512, 2048, False, 'gelu16'), (21120, 2048, 1024, True, 'concat'), (21120, 512, 512, False, 'rows16'),
    (84480, 768, 256, True, 'ln'), (84480, 256, 1024, False, 'gelu16'), (337920, 512, 128, True, 'ln'), (8192, 128, 128, True, 'rows'),
    (8256, 192, 64, True, 'ln'), (21120, 64, 320, False, 'gelu16'), (16384, 320, 128, True, 'concat'),
])
def test_wgrad_wide_bf16(bf16_ops, M, N, K, dy16, xmode):
    import torch.nn.functional as F
    ops = bf16_ops
    dy = tk.rnd((M, N), 21)
    if dy16:
        dy = dy.to(torch.bfloat16).float()
    d = lambda t: t.detach().to(tk.DEV)  # noqa
    dW, db = torch.zeros((N, K), device=tk.DEV), torch.zeros((N,), device=tk.DEV)
    dyd = d(dy).to(torch.bfloat16) if dy16 else d(dy)
    if xmode in ('rows', 'rows16'):
        x = tk.rnd((M, K), 22)
        if xmode == 'rows16':
            x = x.to(A16())
        X = x.float()
        xd = d(x)
        call = lambda: ops.linear_wgrad(dyd, xd, dW, db, x_gelu=False)  # noqa
    elif xmode == 'ln':
        x = 0.3 + 1.5 * tk.rnd((M, K), 22)
        lw, lb = 1 + 0.2 * tk.rnd((K,), 23), 0.1 * tk.rnd((K,), 24)
        X = F.layer_norm(x, (K,), lw, lb, 1e-5)
        _, st = ops.layernorm_fwd(d(x), d(lw), d(lb), want_stats=True)
        call = lambda: ops.linear_wgrad(dyd, d(x), dW, db, stats=st, ln_w=d(lw), ln_b=d(lb))  # noqa
    elif xmode == 'gelu16':
        u16 = tk.rnd((M, K), 22).to(torch.float16)
        X = F.gelu(u16.float())
        ud = d(u16)
        call = lambda: ops.linear_wgrad(dyd, ud, dW, db)  # noqa
    else:
        K1 = K // 2
        x, h = tk.rnd((M, K1), 22), tk.rnd((M, K - K1), 25)
        X = torch.cat([x, h], 1)
        call = lambda: ops.linear_wgrad(dyd, d(x), dW, db, x2=d(h))  # noqa
    ref_w, ref_b = dy.double().t() @ X.double(), dy.double().sum(0)
    call()
    tk.close(dW, ref_w.float(), what=f'dW {M}x{N}x{K} {xmode}')
    tk.close(db, ref_b.float(), what='dbias')
    call()
    tk.close(dW, 2 * ref_w.float(), what='dW accumulates')
    tk.close(db, 2 * ref_b.float(), what='dbias accumulates')


# (C = 384 runs the per-timestep kernels in either mode: covered by test_convlstm_bf16 above)
@pytest.mark.parametrize('T,B,H,W,C,state', [(4, 1, 7, 10, 48, True), (3, 2, 8, 10, 32, False), (5, 4, 16, 40, 96, True), (3, 2, 16, 20, 192, True),
                                             (21, 2, 16, 20, 48, True), (21, 1, 8, 10, 192, True),
                                             (5, 2, 8, 10, 384, True), (3, 1, 5, 7, 256, False),    # weights streamed from the packed copy
                                             (4, 2, 6, 10, 512, True), (11, 1, 5, 7, 512, False)])  # RVT-B stage 4: 16 waves per workgroup
def test_convlstm_sequence_bf16(bf16_ops, T, B, H, W, C, state):
    tk.test_convlstm_sequence(bf16_ops, T, B, H, W, C, state)


def test_full_size_training_step_bf16_vs_f32():
    """BASELINE configs[1] (RVT-S, Gen1 240x304, T=21, bs=8) through Module.training_step + FlatAdamW: the bf16 mode against the fp32
    mode of the same build on the same weights and batch (the fp32 mode is pinned to the CPU oracle at this size by
    tests/test_engine_gpu.py::test_full_size_training_step_vs_oracle).  The total loss agrees to 1e-2 relative, its components and the SimOTA
    foreground count to 3 %, the final LSTM cell states to 3e-2 of their magnitude, and the parameter update moves the same way."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import json
    import os
    import numpy as np
    import test_engine_gpu as te
    from oracle.synth import synth_state_dict
    from leod_amd import ops
    from leod_amd.modules.utils.detection import Mode
    from leod_amd.optim import fit_step
    man = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g11_manifest.json')))
    sd = synth_state_dict(man['small_gen1'], 3)
    ev, labels, label_tb = te._full_size_batch(seed=11)
    first = torch.ones(8, dtype=torch.bool, device='cuda')
    res = {}
    prev = ops.get_precision()
    try:
        for mode in ('f32', 'bf16', 'f32_perturbed'):
            mod, opt, lrs = te._full_size_module(0)
            mod.mdl.load_state_dict(sd)
            if mode == 'f32_perturbed':
                # control: the fp32 mode on weights perturbed by rounding noise of bf16 size (2^-9 relative) -- how much of the
                # difference is the sensitivity of this (random-weight) network rather than the arithmetic of the bf16 kernels
                g = torch.Generator(device='cuda').manual_seed(1)
                with torch.no_grad():
                    opt.flat.data.mul_(1 + 2.0 ** -9 * (2 * torch.rand(opt.flat.data.shape, generator=g, device='cuda') - 1))
                    opt.flat.touch()                            # edited through the flat alias: drop cached packed weights
            ops.set_precision('bf16' if mode == 'bf16' else 'f32')
            opt.clip_value = 0.0                                # compare the raw gradients (value clipping saturates many entries at +-1)
            out = fit_step(mod, opt, lrs, te._loader_batch(ev, labels.cpu().numpy(), label_tb, first.clone()))
            states = mod.mode_2_rnn_states[Mode.TRAIN].get_states(0)
            res[mode] = ({k: float(out['log_dict'][f'train/{k}'].detach()) for k in te.KEYS}, [c.cpu().numpy() for _, c in states],
                         opt.flat.data.cpu().numpy(), opt.flat.grad.cpu().numpy())
            del mod, opt, lrs, out, states
            torch.cuda.empty_cache()
    finally:
        ops.set_precision(prev)
    lf, lb = res['f32'][0], res['bf16'][0]
    print('losses f32', lf, 'bf16', lb)
    # a random-init head sits on many near-ties of the SimOTA cost: a 2^-9 operand rounding moves ~1 % of the assignments, which
    # shows in the per-component losses; the total is dominated by the objectness term over all anchors
    # ... and the network is chaotic in that noise: the fp32 mode on weights perturbed by 2^-9 (the control below) already moves the total
    # by 0.64 %, and two bf16 schedules whose LSTM states agree to 2e-6 rms (per-timestep kernels vs csrc/k_lstm.hip; measured in round 3)
    # land at 0.77 % and 1.62 %.  The bound is therefore stated against the control: within 3x its deviation (and never looser than 3 %).
    lc = res['f32_perturbed'][0]
    tol = min(3e-2, max(1e-2, 3.0 * abs(lc['loss'] - lf['loss']) / lf['loss']))
    assert lb['loss'] == pytest.approx(lf['loss'], rel=tol), (lf, lb, lc)
    for k in ('iou_loss', 'conf_loss', 'cls_loss', 'num_fg'):
        assert lb[k] == pytest.approx(lf[k], rel=3e-2), (k, lf[k], lb[k])
    for a, b in zip(res['f32'][1], res['bf16'][1]):
        assert np.abs(a - b).max() <= 3e-2 * np.abs(a).max()
    gf, gb, gp = res['f32'][3], res['bf16'][3], res['f32_perturbed'][3]
    cos = float((gf * gb).sum() / (np.linalg.norm(gf) * np.linalg.norm(gb)))
    cos_ctrl = float((gf * gp).sum() / (np.linalg.norm(gf) * np.linalg.norm(gp)))
    print('gradient cosine bf16 vs f32:', cos, 'norms', float(np.linalg.norm(gf)), float(np.linalg.norm(gb)))
    print('gradient cosine f32(perturbed weights) vs f32:', cos_ctrl, 'losses', res['f32_perturbed'][0])
    mod, opt, _ = te._full_size_module(0)
    worst = []
    for (name, p), off in zip(mod.mdl.named_parameters(), opt.flat.offsets):
        a, b = gf[off:off + p.numel()], gb[off:off + p.numel()]
        na, nb = np.linalg.norm(a), np.linalg.norm(b)
        worst.append((float((a * b).sum() / max(na * nb, 1e-30)), name, float(na), float(nb)))
    for c, name, na, nb in sorted(worst)[:12]:
        print(f'  cos {c:.4f}  |g_f32| {na:.3e}  |g_bf16| {nb:.3e}  {name}')
    # measured on MI355X: 0.914 (bf16) against 0.978 for the fp32 mode on weights perturbed by 2^-9 relative noise -- this
    # random-weight network amplifies rounding noise through 4 stages x 21 timesteps of back-propagation (stage-1 tensors suffer
    # most); every kernel of the bf16 mode is checked on its own above
    assert cos_ctrl > 0.95 and cos > 0.88


@pytest.mark.parametrize('B,H,W,C,heads,part', [(2, 16, 20, 48, 2, (8, 10)), (1, 8, 10, 384, 16, (8, 10)), (3, 32, 40, 96, 4, (8, 10)),
                                                (1, 8, 10, 24, 1, (8, 10)), (2, 16, 20, 96, 3, (8, 10)), (1, 12, 20, 64, 2, (6, 10)),
                                                (1, 16, 16, 32, 1, (8, 8)), (1, 12, 20, 32, 1, (6, 10)), (1, 14, 16, 24, 1, (7, 8))])
@pytest.mark.parametrize('window', [True, False])
def test_partition_attn_bf16(bf16_ops, B, H, W, C, heads, part, window):
    tk.test_partition_attn(bf16_ops, B, H, W, C, heads, part, window)


@pytest.mark.parametrize('u8,B,H,W,Hp,Wp,N', [(True, 2, 60, 90, 64, 96, 48), (True, 2, 60, 88, 64, 96, 48), (True, 1, 240, 304, 256, 320, 48),
                                              (True, 7, 240, 304, 256, 320, 48), (True, 2, 36, 52, 40, 64, 32),   # 560 tiles: the persistent loop of k_stem.hip
                                              (True, 2, 60, 88, 64, 96, 64), (True, 3, 180, 320, 192, 320, 64)])  # RVT-B: 64 channels = two slices of 32 in the weight gradient
def test_stem_conv_bf16(bf16_ops, u8, B, H, W, Hp, Wp, N):
    tk.test_stem_conv(bf16_ops, u8, B, H, W, Hp, Wp, N)


@pytest.mark.parametrize('M,N,K,with_res', [(40009, 192, 48, True), (20011, 144, 48, True), (5000, 192, 48, True), (20000, 288, 96, True)])
def test_linear_dgrad_ln_bwd_bf16(bf16_ops, M, N, K, with_res):
    tk.test_linear_dgrad_ln_bwd(bf16_ops, M, N, K, with_res)


def test_conv_bn_bf16(bf16_ops):
    tk.test_conv_bn_eval_and_train(bf16_ops, 8)


@pytest.mark.parametrize('M,N,K,with_res', [(40009, 192, 48, True), (20011, 144, 48, False),       # stage 1: narrow row-streaming kernel
                                            (20011, 288, 96, True), (16400, 384, 96, True),         # stage 2: its 96-column form
                                            (5000, 288, 96, True)])                                 # two-kernel fallback
def test_linear_dgrad_ln_bwd_from_bf16_rows(bf16_ops, M, N, K, with_res):
    """dgrad of (LayerNorm -> Linear) with the LayerNorm backward in the epilogue, fed by bf16 gradient rows (du / dqkv of precision
    mode bf16): against autograd on the same (rounded) rows."""
    import torch.nn.functional as F
    ops = bf16_ops
    x = tk.rnd((M, K), 1).requires_grad_(True)
    lw, lb = (1 + 0.2 * tk.rnd((K,), 2)).requires_grad_(True), (0.1 * tk.rnd((K,), 3)).requires_grad_(True)
    W = tk.rnd((N, K), 4, 0.2)
    dy16 = tk.rnd((M, N), 5).to(torch.bfloat16)
    dres = tk.rnd((M, K), 6)
    F.linear(F.layer_norm(x, (K,), lw, lb, 1e-5), W).backward(dy16.float())
    d = lambda t: t.detach().to(tk.DEV)  # noqa
    _, st = ops.layernorm_fwd(d(x), d(lw), d(lb), want_stats=True)
    dw, db = torch.zeros(K, device=tk.DEV), torch.zeros(K, device=tk.DEV)
    dx = ops.linear_dgrad_ln_bwd(dy16.to(tk.DEV), W.to(tk.DEV), d(x), st, d(lw), d(dres) if with_res else None, dw, db)
    tk.close(dx, x.grad + (dres if with_res else 0), what='dx')
    tk.close(dw, lw.grad, what='d ln weight')
    tk.close(db, lb.grad, what='d ln bias')


@pytest.mark.parametrize('C', [48, 64])        # stage 1 of RVT-S / -T and of RVT-B
@pytest.mark.parametrize('M,saved', [(16384, True), (20007, True), (70001, False), (16391, False)])
def test_mlp_fwd_fused_bf16(bf16_ops, M, saved, C):
    """The whole stage-1 MLP in one launch (csrc/k_mlp.hip: fc1 evaluated transposed so that its accumulators are fc2's A operands, the
    hidden never leaves the registers): z = y + g * (gelu(LN(y) W1^T + b1) W2^T + b2) against the fp32 CPU arithmetic (maxvit.py:110-118,
    268-269), full and ragged row counts; with ``want_saved`` the fp16 pre-activation and the LayerNorm statistics the backward pass reads
    must be the ones the unfused producer writes."""
    import torch.nn.functional as F
    ops = bf16_ops
    y = tk.rnd((M, C), 21)
    lw, lb = 1 + 0.2 * tk.rnd((C,), 22), 0.1 * tk.rnd((C,), 23)
    W1, b1 = tk.rnd((4 * C, C), 24, 0.2), tk.rnd((4 * C,), 25, 0.2)
    W2, b2, g = tk.rnd((C, 4 * C), 26, 0.1), tk.rnd((C,), 27, 0.1), 0.5 + 0.1 * tk.rnd((C,), 28)
    u = F.linear(F.layer_norm(y, (C,), lw, lb, 1e-5), W1, b1)
    z = y + g * F.linear(F.gelu(u), W2, b2)
    d = lambda t: t.detach().to(tk.DEV)  # noqa
    out = ops.mlp_fwd_fused(d(y), d(lw), d(lb), d(W1), d(b1), d(W2), d(b2), d(g), want_saved=saved)
    assert out is not None, 'K = 48 / H = 192 / M >= 16384 in precision mode bf16 is what the fused kernel covers'
    zz, u16, st = out
    tk.close(zz, z, what='fused MLP output', fwd=True)
    if saved:
        assert u16.dtype is torch.float16 and tuple(u16.shape) == (M, 4 * C) and tuple(st.shape) == (M, 2)
        tk.close(u16.float(), u, what='fp16 pre-activation of the fused MLP', fwd=True)
        u_ref, _, st_ref = ops.ln_linear_fwd(d(y), d(lw), d(lb), d(W1), d(b1), want_act=True, want_stats=True)
        assert float((u16.float() - u_ref.float()).abs().max()) <= 2e-2 * float(u_ref.float().abs().max())
        torch.testing.assert_close(st, st_ref, rtol=1e-5, atol=1e-6)
    else:
        assert u16 is None and st is None
    # shapes outside the fused kernel's coverage are refused (the caller falls back to the two-launch path)
    assert ops.mlp_fwd_fused(d(y)[:1000].contiguous(), d(lw), d(lb), d(W1), d(b1), d(W2), d(b2), d(g)) is None


@pytest.mark.parametrize('C', [48, 64])
@pytest.mark.parametrize('M', [16384, 20007, 70001])
def test_mlp_bwd_dgrad_fused_bf16(bf16_ops, M, C):
    """The activation-path backward of the stage-1 MLP in one launch (csrc/k_mlp.hip): u recomputed from y, du = ((dz g) W2) gelu'(u) as
    bf16 rows, dy = dz + LayerNorm-backward(du W1), norm2's weight / bias gradients -- against fp32 autograd of the same block
    (maxvit.py:110-118, 268-269) and against the two-launch path it replaces."""
    import torch.nn.functional as F
    ops = bf16_ops
    y = tk.rnd((M, C), 31)
    lw, lb = 1 + 0.2 * tk.rnd((C,), 32), 0.1 * tk.rnd((C,), 33)
    W1, b1 = tk.rnd((4 * C, C), 34, 0.2), tk.rnd((4 * C,), 35, 0.2)
    W2, b2, g = tk.rnd((C, 4 * C), 36, 0.1), tk.rnd((C,), 37, 0.1), 0.5 + 0.1 * tk.rnd((C,), 38)
    dz = tk.rnd((M, C), 39)
    yr = y.clone().requires_grad_(True)
    lwr, lbr = lw.clone().requires_grad_(True), lb.clone().requires_grad_(True)
    u = F.linear(F.layer_norm(yr, (C,), lwr, lbr, 1e-5), W1, b1)
    u.retain_grad()
    z = yr + g * F.linear(F.gelu(u), W2, b2)
    z.backward(dz)
    d = lambda t: t.detach().to(tk.DEV)  # noqa
    fwd = ops.mlp_fwd_fused(d(y), d(lw), d(lb), d(W1), d(b1), d(W2), d(b2), d(g), want_saved=True)
    assert fwd is not None
    _, u16, st = fwd
    dlw, dlb = torch.zeros((C,), device=tk.DEV), torch.zeros((C,), device=tk.DEV)
    out = ops.mlp_bwd_dgrad_fused(d(dz), d(y), st, d(lw), d(lb), d(W1), d(b1), d(W2), d(g), dlw, dlb)
    assert out is not None
    dy, du = out
    assert du.dtype is torch.bfloat16 and tuple(du.shape) == (M, 4 * C)
    tk.close(du.float(), u.grad, what='gradient of the hidden pre-activation (bf16 rows)')
    tk.close(dy, yr.grad, what='dy = dz + LayerNorm backward')
    tk.close(dlw, lwr.grad, what='LayerNorm weight gradient')
    tk.close(dlb, lbr.grad, what='LayerNorm bias gradient')
    # and the two launches it replaces (same operands, same roundings up to the fp16 storage of u)
    du2 = ops.linear_dgrad(d(dz), d(W2), kscale=d(g), aux_u=u16)
    dlw2, dlb2 = torch.zeros((C,), device=tk.DEV), torch.zeros((C,), device=tk.DEV)
    dy2 = ops.linear_dgrad_ln_bwd(du2, d(W1), d(y), st, d(lw), d(dz), dlw2, dlb2)
    assert float((dy - dy2).abs().max()) <= 2e-2 * float(dy2.abs().max())
    assert float((dlw - dlw2).abs().max()) <= 2e-2 * float(dlw2.abs().max())


@pytest.mark.parametrize('M,C', [(13440, 384), (53760, 192), (20011, 96), (5003, 384), (9001, 256)])
def test_weight_shadow_is_bit_identical(bf16_ops, M, C):
    """bf16 shadow of a registered flat weight buffer (csrc/k_misc.hip, ops.set_weight_shadow): the Linear forward / dgrad kernels of
    stages 2-4 stage their weight tiles from it instead of rounding fp32 weights on every tile load.  Same rounding -> the results must
    be bit-identical with and without the shadow, for every loader (row-major forward weights, transposed dgrad weights, both GEMM
    kernels), on ragged row counts; a stale shadow (after ``weight_shadow_invalidate`` or an AdamW launch on the buffer) is not read."""
    ops = bf16_ops
    n1, n2 = 4 * C * C, C * 4 * C
    flat = torch.zeros(n1 + n2 + 3 * C * C + C * C, device=tk.DEV)
    shadow = torch.empty(flat.numel(), dtype=torch.bfloat16, device=tk.DEV)
    shadow_f16 = torch.empty(flat.numel(), dtype=torch.float16, device=tk.DEV)     # mode 16f: the forward GEMMs read an fp16 copy
    W1 = flat[:n1].view(4 * C, C)
    W2 = flat[n1:n1 + n2].view(C, 4 * C)
    Wq = flat[n1 + n2:n1 + n2 + 3 * C * C].view(3 * C, C)
    Wp = flat[n1 + n2 + 3 * C * C:].view(C, C)
    for k, w in enumerate((W1, W2, Wq, Wp)):
        w.copy_(tk.rnd(tuple(w.shape), 20 + k, 0.2))
    x, res = tk.rnd((M, C), 1).to(tk.DEV), tk.rnd((M, C), 2).to(tk.DEV)
    lw, lb = (1 + 0.2 * tk.rnd((C,), 3)).to(tk.DEV), (0.1 * tk.rnd((C,), 4)).to(tk.DEV)
    b1, b2, g = tk.rnd((4 * C,), 6, 0.2).to(tk.DEV), tk.rnd((C,), 8, 0.1).to(tk.DEV), (0.5 + 0.1 * tk.rnd((C,), 9)).to(tk.DEV)
    bq = tk.rnd((3 * C,), 12, 0.1).to(tk.DEV)
    dz, dq = tk.rnd((M, C), 10).to(tk.DEV), tk.rnd((M, 3 * C), 11).to(tk.DEV)

    def chain():
        u16, _, st = ops.ln_linear_fwd(x, lw, lb, W1, b1, want_act=True, want_stats=True)          # LN -> fc1 (row-major weights)
        z, _ = ops.linear_lsres_fwd(u16, W2, b2, g, res, want_t=False)                             # fc2 + LayerScale + residual
        q, _, _ = ops.ln_linear_fwd(x, lw, lb, Wq, bq)                                             # LN -> qkv
        p, _ = ops.linear_lsres_fwd(x, Wp, b2, g, res, want_t=False)                               # proj + LayerScale + residual
        du = ops.linear_dgrad(dz, W2, kscale=g, aux_u=u16)                                          # dgrad of fc2 through GELU (transposed weights)
        dn = ops.linear_dgrad(du, W1)                                                               # dgrad of fc1
        dx = ops.linear_dgrad(dq, Wq)                                                               # dgrad of qkv
        do = ops.linear_dgrad(dz, Wp, kscale=g)                                                     # dgrad of proj
        return [t.clone() for t in (u16, z, q, p, du, dn, dx, do)]

    ref = chain()                                          # no shadow registered: fp32 weight tiles rounded by the loaders
    ops.set_weight_shadow(flat, shadow)
    ops.set_weight_shadow_f16(flat, shadow_f16)
    try:
        shadow.fill_(float('nan'))
        shadow_f16.fill_(float('nan'))
        stale = chain()                                    # registered but never refreshed: must not be read
        for a, b in zip(ref, stale):
            assert torch.equal(a, b)
        assert ops.weight_shadow_refresh() == 1 and ops.weight_shadow_refresh() == 0
        assert torch.equal(shadow.float(), flat.to(torch.bfloat16).float())
        if ops.get_precision() == '16f':
            assert torch.equal(shadow_f16.float(), flat.to(torch.float16).float())
        fresh = chain()
        for k, (a, b) in enumerate(zip(ref, fresh)):
            assert torch.equal(a, b), f'output {k} differs with the bf16 weight shadow'
        # an optimiser launch on the buffer makes the shadow stale: poison it, the kernels must be back on the fp32 weights
        gbuf, m, v = torch.zeros_like(flat), torch.zeros_like(flat), torch.zeros_like(flat)
        ops.adamw_clip_step(flat, gbuf, m, v, 0.0, 1)
        shadow.fill_(float('nan'))
        shadow_f16.fill_(float('nan'))
        after = chain()
        for a, b in zip(ref, after):
            assert torch.equal(a, b)
        assert ops.weight_shadow_refresh() == 1
    finally:
        ops.set_weight_shadow(flat, None)
