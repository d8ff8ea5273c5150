"""Precision mode 'bf16' (ops.set_precision; the reference's Lightning precision=16 placement, train.py:236-243): the contraction
tests of tests/test_kernels_gpu.py run again with bf16 MFMA operands against the SAME fp32 CPU references, at the tolerance stated
there (worst element 3e-2 of the tensor's magnitude, rms error 4x tighter).  ``pytest -m gpu``."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import test_kernels_gpu as tk  # noqa: E402


@pytest.fixture()
def bf16_ops():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from leod_amd import ops
    prev = ops.set_precision('bf16')
    tk.MODE['bf16'] = True
    try:
        yield ops
    finally:
        tk.MODE['bf16'] = False
        ops.set_precision(prev)


def test_precision_switch(bf16_ops):
    assert bf16_ops.get_precision() == 'bf16'
    assert bf16_ops.set_precision('f32') == 1 and bf16_ops.get_precision() == 'f32'
    bf16_ops.set_precision('bf16')


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


def test_full_size_training_step_bf16_vs_f32():
    """BASELINE configs[1] (RVT-S, Gen1 240x304, T=21, bs=8) through Module.training_step + FlatAdamW: the bf16 mode against the fp32
    mode of the same build on the same weights and batch (the fp32 mode is pinned to the CPU oracle at this size by
    tests/test_engine_gpu.py::test_full_size_training_step_vs_oracle).  Losses agree to 1e-2 relative, the SimOTA foreground
    count to 2 %, the final LSTM cell states to 3e-2 of their magnitude, and the parameter update moves the same way."""
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
        for mode in ('f32', 'bf16'):
            mod, opt, lrs = te._full_size_module(0)
            mod.mdl.load_state_dict(sd)
            ops.set_precision(mode)
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
    for k in ('loss', 'iou_loss', 'conf_loss', 'cls_loss'):
        assert lb[k] == pytest.approx(lf[k], rel=1e-2), (k, lf[k], lb[k])
    assert lb['num_fg'] == pytest.approx(lf['num_fg'], rel=2e-2)
    for a, b in zip(res['f32'][1], res['bf16'][1]):
        assert np.abs(a - b).max() <= 3e-2 * np.abs(a).max()
    gf, gb = res['f32'][3], res['bf16'][3]
    cos = float((gf * gb).sum() / (np.linalg.norm(gf) * np.linalg.norm(gb)))
    print('clipped-gradient cosine bf16 vs f32:', cos)
    assert cos > 0.98
