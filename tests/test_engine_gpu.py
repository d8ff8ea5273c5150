"""Whole-step parity on the GPU: two consecutive training steps of the HIP engine (eager and hipGraph replay)
against the golden vectors recorded from the reference's own training loop (g12) and the pseudo-label
inference pass against the oracle.  ``pytest -m gpu``."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import postproc as op  # noqa: E402
from oracle import train_step as ot  # noqa: E402
from oracle.synth import synth_state_dict, synth_events, synth_labels  # noqa: E402

DEV = 'cuda'
MICRO = ot.model_cfg(embed_dim=16, dim_head=8, fpn_depth=0.33, partition_size=(2, 3), in_res_hw=(64, 96))
KEYS = ('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss', 'num_fg')


@pytest.fixture(scope='module')
def gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return True


def micro_detector(manifest, seed):
    from leod_amd.config import full_config, dynamically_modify_train_config
    from leod_amd.models.detection.yolox_extension.models.detector import YoloXDetector
    over = dict(model=dict(backbone=dict(embed_dim=16, stage=dict(attention=dict(dim_head=8)))))
    cfg = dynamically_modify_train_config(full_config('gen1', 'small', overrides=over))
    cfg.model.backbone.in_res_hw = (64, 96)
    cfg.model.backbone.stage.attention.partition_size = (2, 3)
    det = YoloXDetector(cfg.model)
    sd = synth_state_dict(manifest['micro'], seed)
    det.load_state_dict(sd)
    return det.to(DEV), sd


def micro_labels(n_frames, seed, hw=(60, 90)):
    labs = synth_labels(n_frames, hw, 2, seed=seed, max_boxes=4)
    for l in labs:
        l[:, 3] = l[:, 3].clamp(max=30)
        l[:, 4] = l[:, 4].clamp(max=24)
        l[:, 1] = torch.minimum(l[:, 1], hw[1] - 1 - l[:, 3])
        l[:, 2] = torch.minimum(l[:, 2], hw[0] - 1 - l[:, 4])
    return labs


def g12_inputs(step, T=5, B=2):
    ev = synth_events(T, B, 20, 60, 90, seed=20 + step, as_uint8=True)
    lab_list = micro_labels(T * B, seed=30 + step)
    label_tb, labs = [], []
    for t in range(T):
        idx = [b for b in range(B) if (t in (2, 4) or (t == 1 and b == 0))]
        label_tb.append(idx)
        labs += [lab_list[t * B + b] for b in idx]
    is_first = torch.tensor([True, True]) if step == 0 else torch.tensor([False, True])
    return ev, op.batched_yolox_labels(labs), label_tb, is_first


def test_two_training_steps_match_reference_golden(gpu, golden_dir, manifest):
    from leod_amd.engine import TrainEngine
    g = np.load(os.path.join(golden_dir, 'g12_trainstep_micro.npz'))
    det, _ = micro_detector(manifest, 9)
    eng = TrainEngine(det, lr=2e-4, total_steps=1000, div_factor=20, final_div_factor=10000)
    params = dict(det.named_parameters())
    for step in range(2):
        ev, labels, label_tb, is_first = g12_inputs(step)
        losses = eng.step(ev.to(DEV), labels.to(DEV), label_tb, is_first.to(DEV))
        got = np.array([float(losses[k]) for k in KEYS])
        np.testing.assert_allclose(got, g[f's{step}_losses'], rtol=1e-4, err_msg=f'losses step {step}')
        keys = [str(k) for k in g[f's{step}_grad_keys']]
        # after the optimiser kernel .grad holds the value-clipped gradient, exactly what the golden run recorded
        np.testing.assert_allclose(np.array([float(params[k].grad.norm()) for k in keys]), g[f's{step}_grad_norms'],
                                   rtol=3e-3, atol=1e-6, err_msg=f'grad norms step {step}')
        np.testing.assert_allclose(np.array([float(params[k].detach().norm()) for k in keys]), g[f's{step}_param_norms'],
                                   rtol=2e-4, err_msg=f'param norms step {step}')
        assert abs(eng.current_lr() - float(g[f's{step}_lr_next'])) < 1e-12
        np.testing.assert_allclose(eng.states[3][1].cpu().numpy(), g[f's{step}_state_c4'], rtol=2e-4, atol=2e-5)


def test_graph_replay_equals_eager(gpu, manifest):
    from leod_amd.engine import TrainEngine
    # same label layout on every step (a captured graph has static shapes)
    T, B = 4, 2
    label_tb = [[], [0], [], [0, 1]]
    res = {}
    for mode in ('eager', 'graph', 'plan'):
        det, _ = micro_detector(manifest, 9)
        eng = TrainEngine(det, lr=2e-4, total_steps=1000)
        out = []
        for step in range(3):
            ev = synth_events(T, B, 20, 60, 90, seed=50 + step, as_uint8=True).to(DEV)
            labels = torch.zeros((3, 4, 7))
            ll = op.batched_yolox_labels(micro_labels(3, seed=60 + step))
            labels[:, :ll.shape[1]] = ll
            labels = labels.to(DEV)
            is_first = torch.tensor([step == 0, True], device=DEV)
            if mode in ('graph', 'plan'):
                if step == 0:                                 # 'plan': the same capture replayed by a launch plan (csrc/k_plan.hip), side stream included
                    eng.capture(ev, labels, label_tb, is_first, plan=(mode == 'plan'))
                losses = eng.step_graph(ev, labels, is_first)
            else:
                losses = eng.step(ev, labels, label_tb, is_first)
            out.append([float(losses[k]) for k in KEYS])
        res[mode] = (np.array(out), eng.flat.data.clone().cpu(), [c.clone().cpu() for _, c in eng.states])
    np.testing.assert_allclose(res['graph'][0], res['eager'][0], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(res['plan'][0], res['eager'][0], rtol=1e-4, atol=1e-5)
    assert float(np.abs(res['plan'][1].numpy() - res['eager'][1].numpy()).max()) < 3.5e-4
    # parameters: Adam turns noise-level gradients (fp32 atomics accumulate in a different order on every run) into
    # +-lr steps, so a handful of elements may differ by up to 2*sum(lr) = 3.4e-4; everything else must agree tightly
    pg, pe = res['graph'][1].numpy(), res['eager'][1].numpy()
    diff = np.abs(pg - pe)
    assert diff.max() < 3.5e-4
    assert (diff > 2e-6 + 1e-4 * np.abs(pe)).mean() < 5e-3
    for a, b in zip(res['graph'][2], res['eager'][2]):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=5e-4, atol=5e-5)


def test_pseudo_label_inference_vs_oracle(gpu, manifest):
    from leod_amd.engine import PseudoLabelEngine
    det, sd = micro_detector(manifest, 5)
    ev = synth_events(4, 2, 20, 60, 90, seed=3, as_uint8=True)
    pl = PseudoLabelEngine(det, 2, conf_thre=0.01, obj_thresh=[0.1, 0.05], cls_thresh=[0.1, 0.05], hflip=True, max_det=126)
    lab, lcnt, dets, cnt = pl.step(ev.to(DEV))
    rdets, _, _ = ot.infer_sequence(sd, MICRO, ev, conf_thre=0.01, hflip=True)
    assert [int(c) for c in cnt.cpu()] == [len(r) for r in rdets]
    for i, r in enumerate(rdets):
        np.testing.assert_allclose(dets[i, :len(r)].cpu().numpy(), r.numpy(), rtol=2e-4, atol=2e-4)
    rl = op.pred2label([r.clone() for r in rdets], [0.1, 0.05], [0.1, 0.05], 'gen1', False)
    assert [int(c) for c in lcnt.cpu()] == [len(r) for r in rl]
    for i, r in enumerate(rl):
        np.testing.assert_allclose(lab[i, :len(r)].cpu().numpy(), r.numpy(), rtol=2e-4, atol=2e-4)


# ---- world-size 2 on ONE GPU (gloo carries the device tensors): SyncBatchNorm + gradient all-reduce end to end -------------
def _world2_worker(rank, port, manifest, q, buckets='1'):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK='0')
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=2)
    from leod_amd.engine import TrainEngine
    det, _ = micro_detector(manifest, 9)
    eng = TrainEngine(det, lr=2e-4, total_steps=1000, grad_buckets=buckets == '1')
    T, B = 4, 4
    ev = synth_events(T, B, 20, 60, 90, seed=70, as_uint8=True)
    labs_all = micro_labels(T * B, seed=71)
    mine = [2 * rank, 2 * rank + 1]                                   # this rank's half of the global batch
    label_tb = [[], [0, 1], [], [0, 1]]
    labs = [labs_all[t * B + mine[b]] for t in range(T) for b in label_tb[t]]
    labels = op.batched_yolox_labels(labs).to(DEV)
    assert (eng.dp.buckets is not None) == (buckets == '1')
    losses = eng.step(ev[:, mine].to(DEV), labels, label_tb, torch.ones(2, dtype=torch.bool, device=DEV))
    if buckets == '1':               # released by the boundary nodes of the backward pass: head first, stage 1 by finish()
        assert eng.dp.buckets.order == [4, 3, 2, 1, 0], eng.dp.buckets.order
    bns = [m.bn for m in det.modules() if hasattr(m, 'bn')]
    q.put((rank, float(losses['loss']), eng.flat.data.detach().cpu().numpy(),
           np.concatenate([b.running_mean.detach().cpu().numpy() for b in bns]),
           np.concatenate([b.running_var.detach().cpu().numpy() for b in bns])))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_syncbn_and_replica_consistency(gpu, manifest):
    """Two ranks (each half of a batch of 4 sequences) through the whole HIP training step with the data-parallel engine:
    the replicas must end bit-identical (same all-reduced gradients, same optimiser step), and the SyncBatchNorm running
    statistics must equal those of ONE process that sees the full batch (train.py:247 sync_batchnorm semantics)."""
    import torch.multiprocessing as mp
    from leod_amd.engine import TrainEngine
    ctx = mp.get_context('spawn')
    runs = {}
    for buckets in ('1', '0'):      # per-stage gradient buckets exchanged under the backward pass vs ONE flat all-reduce after it
        q = ctx.Queue()
        port = 31000 + ((os.getpid() + int(buckets)) % 2000)
        procs = [ctx.Process(target=_world2_worker, args=(r, port, manifest, q, buckets)) for r in range(2)]
        for p in procs:
            p.start()
        runs[buckets] = sorted((q.get(timeout=300) for _ in range(2)), key=lambda r: r[0])
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
    res = runs['1']
    # a sum over two ranks has one rounding whichever way it is chunked: the bucketed and the flat exchange agree up to the
    # atomics' accumulation order inside the weight-gradient kernels (Adam turns noise-level gradients into +-lr steps)
    d = np.abs(runs['1'][0][2] - runs['0'][0][2])
    assert d.max() < 4.5e-4 and (d > 2e-6 + 1e-4 * np.abs(runs['0'][0][2])).mean() < 5e-3
    np.testing.assert_array_equal(res[0][2], res[1][2])               # replicas identical after the step
    np.testing.assert_array_equal(res[0][3], res[1][3])
    np.testing.assert_array_equal(res[0][4], res[1][4])
    # single process, full batch: BatchNorm statistics of the forward pass are the global ones
    det, _ = micro_detector(manifest, 9)
    eng = TrainEngine(det, lr=2e-4, total_steps=1000, grad_buckets=buckets == '1')
    T, B = 4, 4
    ev = synth_events(T, B, 20, 60, 90, seed=70, as_uint8=True)
    labs_all = micro_labels(T * B, seed=71)
    label_tb = [[], [0, 1, 2, 3], [], [0, 1, 2, 3]]
    labs = [labs_all[t * B + b] for t in range(T) for b in label_tb[t]]
    eng.step(ev.to(DEV), op.batched_yolox_labels(labs).to(DEV), label_tb, torch.ones(B, dtype=torch.bool, device=DEV))
    bns = [m.bn for m in det.modules() if hasattr(m, 'bn')]
    rm = np.concatenate([b.running_mean.detach().cpu().numpy() for b in bns])
    rv = np.concatenate([b.running_var.detach().cpu().numpy() for b in bns])
    np.testing.assert_allclose(res[0][3], rm, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(res[0][4], rv, rtol=2e-5, atol=1e-6)


# ---- BASELINE.json configs[1] size (RVT-S, Gen1 240x304, T=21, bs=8): size-independent properties -----------------------
def _full_size_engine(seed=0):
    from leod_amd.config import full_config, dynamically_modify_train_config
    from leod_amd.models.detection.yolox_extension.models.detector import YoloXDetector
    from leod_amd.engine import TrainEngine
    cfg = dynamically_modify_train_config(full_config('gen1', 'small'))
    torch.manual_seed(seed)
    det = YoloXDetector(cfg.model).to(DEV)
    return TrainEngine(det, lr=cfg.training.learning_rate), det


def _full_size_batch(T=21, B=8, seed=3):
    g = torch.Generator().manual_seed(seed)
    mask = torch.rand((T, B, 20, 240, 304), generator=g) < 0.08
    ev = (mask * torch.randint(1, 10, (T, B, 20, 240, 304), generator=g)).to(torch.uint8)
    rng = np.random.RandomState(seed)
    label_tb, rows = [], []
    for t in range(T):
        idx = list(range(B)) if t in (4, 9, 14, 19) else []
        label_tb.append(idx)
        for _ in idx:
            n = rng.randint(1, 6)
            w, h = rng.uniform(10, 90, n), rng.uniform(10, 70, n)
            x, y = rng.uniform(0, 303 - w), rng.uniform(0, 239 - h)
            rows.append(np.stack([rng.randint(0, 2, n), x + w / 2, y + h / 2, w, h, np.ones(n), np.ones(n)], 1).astype(np.float32))
    nmax = max(len(r) for r in rows)
    lab = np.zeros((len(rows), nmax, 7), np.float32)
    for i, r in enumerate(rows):
        lab[i, :len(r)] = r
    return ev.to(DEV), torch.from_numpy(lab).to(DEV), label_tb


def _full_size_module(seed=0, schedule='batched'):
    from leod_amd.config import full_config, dynamically_modify_train_config
    from leod_amd.modules.utils.fetch import fetch_model_module
    cfg = dynamically_modify_train_config(full_config('gen1', 'small'))
    torch.manual_seed(seed)
    mod = fetch_model_module(cfg).to(DEV)
    mod.setup('fit')
    mod.train()
    mod.time_batched = schedule == 'batched'
    oc = mod.configure_optimizers()
    return mod, oc['optimizer'], oc['lr_scheduler']['scheduler']


def _loader_batch(ev, rows, label_tb, is_first, hw=(240, 304)):
    """Device frames + host box labels in the dictionary the reference's loaders emit; ``rows`` are yolox rows
    (cls, cx, cy, w, h, obj, cls_conf) of the labelled frames in (t, b) order."""
    from leod_amd.data.genx_utils.labels import ObjectLabels, SparselyBatchedObjectLabels
    from leod_amd.data.utils.types import DataType
    from leod_amd.modules.utils.detection import WORKER_ID_KEY, DATA_KEY
    T, B = ev.shape[:2]
    it = iter(rows)
    seq = []
    for t in range(T):
        row = [None] * B
        for b in label_tb[t]:
            r = torch.as_tensor(next(it))
            r = r[r.sum(1) > 0]
            row[b] = ObjectLabels(torch.stack([torch.ones(len(r)), r[:, 1] - r[:, 3] / 2, r[:, 2] - r[:, 4] / 2, r[:, 3], r[:, 4],
                                               r[:, 0], r[:, 6], r[:, 5]], 1), hw)
        seq.append(SparselyBatchedObjectLabels(row))
    return {WORKER_ID_KEY: 0, DATA_KEY: {DataType.EV_REPR: [ev[t] for t in range(T)], DataType.OBJLABELS_SEQ: seq,
                                         DataType.IS_FIRST_SAMPLE: is_first}}


def test_full_size_schedule_invariance(gpu):
    """At the benchmark size, through the Module surface: the stage-major time-batched schedule and the reference's
    timestep-major loop over ``forward_backbone`` are the same function -- losses, final LSTM states and the updated
    parameters of one training step (training_step + FlatAdamW.step) agree."""
    from leod_amd.optim import fit_step
    from leod_amd.modules.utils.detection import Mode
    ev, labels, label_tb = _full_size_batch()
    first = torch.ones(8, dtype=torch.bool, device=DEV)
    out = {}
    for sched in ('batched', 'timestep'):
        mod, opt, lrs = _full_size_module(0, sched)
        res = fit_step(mod, opt, lrs, _loader_batch(ev, labels.cpu().numpy(), label_tb, first))
        states = mod.mode_2_rnn_states[Mode.TRAIN].get_states(0)
        out[sched] = (np.array([float(res['log_dict'][f'train/{k}'].detach()) for k in KEYS]), opt.flat.data.detach().cpu().numpy(),
                      [c.detach().cpu().numpy() for _, c in states])
    np.testing.assert_allclose(out['batched'][0], out['timestep'][0], rtol=2e-5, atol=1e-6)
    for a, b in zip(out['batched'][2], out['timestep'][2]):
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-5)
    d = np.abs(out['batched'][1] - out['timestep'][1])
    assert d.max() < 4.5e-4                                   # <= 2 * lr: Adam sign flips on noise-level gradients only
    assert (d > 2e-6 + 1e-4 * np.abs(out['timestep'][1])).mean() < 5e-3


def test_full_size_module_step_equals_engine_step(gpu):
    """The reference-shaped surface IS the fast path: ``Module.training_step`` + ``FlatAdamW.step`` + ``OneCycleLR.step``
    (fetch_model_module -> configure_optimizers, driven like Lightning's automatic optimisation) and the tensor-level
    ``TrainEngine.step`` run the same kernels in the same order -- two steps at the benchmark size, second one with carried
    LSTM state and a partial reset: losses, learning rate and parameters agree."""
    from leod_amd.optim import fit_step
    ev, labels, label_tb = _full_size_batch()
    ev2, labels2, _ = _full_size_batch(seed=4)
    firsts = [torch.ones(8, dtype=torch.bool, device=DEV),
              torch.tensor([False, True, False, False, True, True, True, True], device=DEV)]
    eng, _ = _full_size_engine(0)
    mod, opt, lrs = _full_size_module(0)
    np.testing.assert_array_equal(eng.flat.data.cpu().numpy(), opt.flat.data.cpu().numpy())
    for step, (e, l, f) in enumerate(zip((ev, ev2), (labels, labels2), firsts)):
        le = eng.step(e, l, label_tb, f.clone())
        assert abs(opt.param_groups[0]['lr'] - one_cycle(step)) < 1e-12
        res = fit_step(mod, opt, lrs, _loader_batch(e, l.cpu().numpy(), label_tb, f.clone()), step)
        got = np.array([float(res['log_dict'][f'train/{k}'].detach()) for k in KEYS])
        np.testing.assert_allclose(got, np.array([float(le[k]) for k in KEYS]), rtol=2e-5, atol=1e-6, err_msg=f'step {step}')
        d = np.abs(opt.flat.data.cpu().numpy() - eng.flat.data.cpu().numpy())
        assert d.max() < 4.5e-4 * (step + 1) and (d > 2e-6 + 1e-4 * np.abs(eng.flat.data.cpu().numpy())).mean() < 5e-3


def one_cycle(step):
    from leod_amd.parallel import one_cycle_lr
    return one_cycle_lr(step, 2e-4, 400000, 0.005, 20, 10000)


def test_full_size_training_step_vs_oracle(gpu, manifest):
    """BASELINE configs[1] at full size against the CPU oracle directly: RVT-S, Gen1 240x304, T=21, bs=8, 32 labelled
    frames, synthetic weights; ONE training step through Module.training_step + FlatAdamW (the benchmarked schedule) vs
    ``OracleTrainer.step`` (modules/detection.py:188-298, yolo_head.py:403-597): the six losses to 2e-5 relative (num_fg is
    the SimOTA foreground count / labelled boxes: equal only if the assignment is identical), a sample of post-step
    parameters, the final LSTM cell states."""
    from oracle.synth import synth_state_dict
    from leod_amd.optim import fit_step
    from leod_amd.modules.utils.detection import Mode
    sd = synth_state_dict(manifest['small_gen1'], 3)
    mod, opt, lrs = _full_size_module(0)
    mod.mdl.load_state_dict(sd)
    ev, labels, label_tb = _full_size_batch(seed=11)
    rows = labels.cpu().numpy()
    first = torch.ones(8, dtype=torch.bool)
    batch = _loader_batch(ev, rows, label_tb, first.to(DEV))
    from leod_amd.data.utils.types import DataType
    seq = batch['data'][DataType.OBJLABELS_SEQ]
    ref_labels = [[None if l is None else l.object_labels.clone() for l in seq[t]] for t in range(21)]
    res = fit_step(mod, opt, lrs, batch)
    otr = ot.OracleTrainer(sd, ot.model_cfg(48, 24, 0.33, (8, 10)))
    ref, ref_grads = otr.step(ev.cpu(), ref_labels, first)
    got = {k: float(res['log_dict'][f'train/{k}'].detach()) for k in KEYS}
    assert got['num_fg'] == pytest.approx(ref['num_fg'], rel=1e-6), 'SimOTA foreground count differs'
    for k in ('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss'):
        assert got[k] == pytest.approx(ref[k], rel=2e-5, abs=1e-6), k
    params = dict(mod.mdl.named_parameters())
    rng = np.random.RandomState(0)
    for k in [otr.param_keys[i] for i in rng.choice(len(otr.param_keys), 40, replace=False)]:
        # after the optimiser kernel .grad holds the value-clipped gradient, which is what the oracle returns
        g, gr = params[k].grad.detach().cpu().numpy().ravel(), ref_grads[k].numpy().ravel()
        np.testing.assert_allclose(g, gr, rtol=3e-3, atol=3e-3 * float(np.abs(gr).max()) + 1e-8, err_msg=f'grad {k}')
        # parameters: lr0 = 1e-5, Adam's first step moves every element by ~lr0 * sign(g); an element whose gradient is pure
        # rounding noise (e.g. the key part of a qkv bias: softmax is shift-invariant) may go the other way: <= 2 * lr0 apart
        a, b = params[k].detach().cpu().numpy().ravel(), otr.sd[k].detach().numpy().ravel()
        assert np.abs(a - b).max() < 2.5e-5, k
    states = mod.mode_2_rnn_states[Mode.TRAIN].get_states(0)
    for (h, c), (rh, rc) in zip(states, otr.states):
        np.testing.assert_allclose(c.cpu().numpy(), rc.numpy(), rtol=2e-4, atol=2e-5)


def test_full_size_three_steps_through_plans_vs_oracle(gpu, manifest):
    """The BENCHMARKED executor against the oracle at the benchmarked size: three consecutive optimisation steps of
    BASELINE configs[1] (RVT-S, Gen1 240x304, T=21, bs=8, 32 labelled frames, three different batches, LSTM state carried with a partial
    reset) through ``Module.training_step`` in plan mode -- step 0 eager, step 1 recorded and replayed, step 2 a pure replay of the
    backbone and head plans -- vs three ``OracleTrainer.step`` calls (fp32 mode).  Losses 2e-5 relative on the first step and 2e-4 on the
    two steps that follow an Adam update (an element whose gradient is rounding noise may move by 2 lr the other way), the SimOTA
    foreground count exactly, sampled parameters after the three steps, the final LSTM cell states."""
    from oracle.synth import synth_state_dict
    from leod_amd.optim import fit_step
    from leod_amd.modules.utils.detection import Mode
    from leod_amd.data.utils.types import DataType
    sd = synth_state_dict(manifest['small_gen1'], 3)
    mod, opt, lrs = _full_size_module(0)
    assert mod.plan_mode and mod.time_batched
    mod.mdl.load_state_dict(sd)
    otr = ot.OracleTrainer(sd, ot.model_cfg(48, 24, 0.33, (8, 10)))
    firsts = [torch.ones(8, dtype=torch.bool), torch.tensor([False, True, False, False, True, False, True, True]),
              torch.tensor([False, False, True, False, False, True, False, False])]
    for step in range(3):
        ev, labels, label_tb = _full_size_batch(seed=11 + step)
        batch = _loader_batch(ev, labels.cpu().numpy(), label_tb, firsts[step].to(DEV))
        seq = batch['data'][DataType.OBJLABELS_SEQ]
        ref_labels = [[None if l is None else l.object_labels.clone() for l in seq[t]] for t in range(21)]
        res = fit_step(mod, opt, lrs, batch, step)
        ref, _ = otr.step(ev.cpu(), ref_labels, firsts[step])
        got = {k: float(res['log_dict'][f'train/{k}'].detach()) for k in KEYS}
        print(f'step {step}: HIP {got}\n        oracle { {k: ref[k] for k in KEYS} }')
        assert got['num_fg'] == pytest.approx(ref['num_fg'], rel=1e-6), f'step {step}: SimOTA foreground count differs'
        for k in ('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss'):
            assert got[k] == pytest.approx(ref[k], rel=2e-5 if step == 0 else 2e-4, abs=1e-6), (step, k)
    pl = mod._plans
    assert (pl.captures, pl.head_captures, pl.steps, pl.replays, pl.eager_steps) == (1, 1, 2, 1, 1), pl.info()
    params = dict(mod.mdl.named_parameters())
    rng = np.random.RandomState(0)
    worst = []
    for k in [otr.param_keys[i] for i in rng.choice(len(otr.param_keys), 40, replace=False)]:
        a, b = params[k].detach().cpu().numpy().ravel(), otr.sd[k].detach().numpy().ravel()
        worst.append((float(np.abs(a - b).max()), float((np.abs(a - b) > 2e-6 + 1e-4 * np.abs(b)).mean()), k))
    worst.sort(reverse=True)
    print('largest parameter differences after three steps (max |diff|, fraction of elements apart, key):', worst[:4])
    # three Adam steps at lr ~1e-5 (OneCycle warm-up from 1e-5): an element whose gradient is rounding noise may move by lr the other way on
    # every step (<= 2 lr apart per step); the typical tensor agrees to 1e-6
    assert worst[0][0] < 1.2e-4, worst[:4]
    assert np.median([w[0] for w in worst]) < 2e-5, worst
    states = mod.mode_2_rnn_states[Mode.TRAIN].get_states(0)
    for (h, c), (rh, rc) in zip(states, otr.states):
        d = np.abs(c.cpu().numpy() - rc.numpy())
        print('cell state: max |diff|', float(d.max()), 'of magnitude', float(np.abs(rc.numpy()).max()))
        # the parameters behind them differ by up to ~2.7e-5 after the three updates (noise-level sign flips, see above)
        np.testing.assert_allclose(c.cpu().numpy(), rc.numpy(), rtol=2e-3, atol=3e-4)


def test_full_size_pseudo_label_pass_properties(gpu):
    """Pseudo-label pass at the benchmark resolution (hflip TTA on): (1) permuting the sequences of the batch permutes the
    outputs and changes nothing else (no kernel mixes batch entries: partition index math, time-batched row order, NMS
    per image); (2) a recording processed as two chunks with the LSTM state carried over gives the same detections as one
    long chunk (RNN state hand-over of pseudo_labeler.py:687-722 under the time-batched schedule)."""
    from leod_amd.engine import PseudoLabelEngine
    _, det = _full_size_engine(1)
    T, B = 10, 4
    ev, _, _ = _full_size_batch(T=T, B=B, seed=5)

    def run(chunks, perm=None):
        # random-init heads score every anchor ~1e-4 (bias -log 99): a threshold below that keeps a few hundred boxes/frame
        eng = PseudoLabelEngine(det, 2, conf_thre=9.5e-5, hflip=True, max_det=512)
        outs = []
        for lo, hi in chunks:
            x = ev[lo:hi] if perm is None else ev[lo:hi][:, perm]
            lab, lcnt, dets, cnt = eng.step(x)
            n = hi - lo
            outs.append((dets.view(n, 2 * B, *dets.shape[1:]).cpu().numpy(), cnt.view(n, 2 * B).cpu().numpy()))
        return np.concatenate([o[0] for o in outs], 0), np.concatenate([o[1] for o in outs], 0)

    d0, c0 = run([(0, T)])
    assert int(c0.sum()) > 0
    # (1) batch permutation (applied to the sequences; the hflip copies follow their sources)
    perm = torch.tensor([2, 0, 3, 1], device=DEV)
    d1, c1 = run([(0, T)], perm)
    p = perm.cpu().numpy()
    idx = np.concatenate([p, B + p])
    np.testing.assert_array_equal(c1, c0[:, idx])
    for t in range(T):
        for j, src in enumerate(idx):
            n = c0[t, src]
            np.testing.assert_allclose(d1[t, j, :n], d0[t, src, :n], rtol=1e-5, atol=1e-5)
    # (2) chunked == one pass
    d2, c2 = run([(0, 4), (4, T)])
    np.testing.assert_array_equal(c2, c0)

    def canon(a):                                             # random-init scores tie to ~1e-7: compare as a set of boxes
        return a[np.lexsort((np.round(a[:, 1], 2), np.round(a[:, 0], 2)))]

    for t in range(T):
        for b in range(2 * B):
            np.testing.assert_allclose(canon(d2[t, b, :c0[t, b]]), canon(d0[t, b, :c0[t, b]]), rtol=2e-4, atol=2e-4)


def test_host_feeder_double_buffer(gpu):
    """HostFeeder: batches arrive intact and in order through the two device buffers, for pinned and pageable sources, with a
    consumer kernel still reading buffer k when batch k + 2 is staged."""
    from leod_amd.engine import HostFeeder
    shape = (3, 2, 20, 60, 76)
    g = torch.Generator().manual_seed(1)
    batches = [torch.randint(0, 255, shape, generator=g, dtype=torch.uint8) for _ in range(6)]
    batches = [b.pin_memory() if i % 2 else b for i, b in enumerate(batches)]
    feeder = HostFeeder(shape, DEV)
    feeder.put(batches[0])
    sums = []
    for i in range(6):
        x = feeder.get()
        if i + 1 < 6:
            feeder.put(batches[i + 1])
        y = x.to(torch.float64)
        for _ in range(20):                       # keep the launch stream busy reading x
            y = y * 1.0 + 0.0
        sums.append((y.sum(), x.clone()))
        feeder.done()
    torch.cuda.synchronize()
    for i, (s, x) in enumerate(sums):
        assert torch.equal(x.cpu(), batches[i]) and float(s) == float(batches[i].to(torch.float64).sum())


@pytest.mark.parametrize('mode', ['f32', '16f'])
def test_training_step_with_block_and_depthwise_options_vs_oracle(gpu, mode):
    """A whole training step of a detector built with the options no shipped config enables -- gated swish MLP (SwiGLU), torch-MHA parameter
    layout, depthwise ConvLSTM conv, non-overlapping patch downsample without LayerNorm affine, depthwise PAFPN / head -- through
    ``TrainEngine`` (flat parameter buffer: the composed nodes hand their parameter gradients to autograd, which must add them into the
    flat gradient views) against ``OracleTrainer``: losses and post-step parameters; fp32 mode tight, mode 16f at the 16-bit class."""
    from leod_amd import ops
    from leod_amd.config import full_config, dynamically_modify_train_config
    from leod_amd.engine import TrainEngine
    from leod_amd.models.detection.yolox_extension.models.detector import YoloXDetector
    over = dict(model=dict(backbone=dict(embed_dim=16, stage=dict(attention=dict(dim_head=8, mlp_gated=True, mlp_activation='swish', use_torch_mha=True),
                                                                    lstm=dict(dws_conv=True, dws_conv_only_hidden=True, dws_conv_kernel_size=3),
                                                                    downsample=dict(type='patch', overlap=False, norm_affine=False))),
                           fpn=dict(depthwise=True), head=dict(depthwise=True)))
    cfg = dynamically_modify_train_config(full_config('gen1', 'small', overrides=over))
    cfg.model.backbone.in_res_hw = (64, 96)
    cfg.model.backbone.stage.attention.partition_size = (2, 3)
    prev = ops.set_precision(mode)
    try:
        det = YoloXDetector(cfg.model)
        man = {k: list(v.shape) for k, v in det.state_dict().items()}
        assert any('mha.in_proj_weight' in k for k in man) and any('net.0.proj.weight' in k for k in man) and any('conv3x3_dws' in k for k in man) \
            and any('dconv' in k for k in man) and not any('downsample_cf2cl.norm.weight' in k for k in man)
        sd = synth_state_dict(man, 31)
        for k in sd:                                  # (oracle.synth knows no `in_proj_*` keys and draws them N(0, 1): scaled like a Linear layer here,
            if k.endswith('in_proj_weight'):          # or the attention logits are in the hundreds and the random network is chaotic)
                sd[k] = sd[k] * (1.2 / sd[k].shape[1] ** 0.5)
            elif k.endswith('in_proj_bias'):
                sd[k] = sd[k] * 0.1
        det.load_state_dict(sd)
        det.to(DEV)
        T, B = 3, 2
        ev = synth_events(T, B, 20, 60, 90, seed=11, as_uint8=True)
        labs = micro_labels(B, seed=12)
        labels = [[None] * B for _ in range(T - 1)] + [labs]
        micro = dict(MICRO, mlp_activation='swish')
        otr = ot.OracleTrainer({k: v.clone() for k, v in sd.items()}, micro, total_steps=1000)
        ref_losses, _ = otr.step(ev, labels, torch.ones(B, dtype=torch.bool))
        eng = TrainEngine(det, total_steps=1000)
        ptrs = [p.grad.data_ptr() for p in eng.flat.params]
        losses = eng.step(ev.to(DEV), op.batched_yolox_labels(labs).to(DEV), [[]] * (T - 1) + [list(range(B))],
                          torch.ones(B, dtype=torch.bool, device=DEV))
        assert [p.grad.data_ptr() for p in eng.flat.params] == ptrs, 'a parameter gradient left the flat buffer'
        tol = 1e-4 if mode == 'f32' else 3e-2
        for k in KEYS:
            assert abs(float(losses[k]) - float(ref_losses[k])) <= tol * abs(float(ref_losses[k])) + 1e-5, (k, float(losses[k]), float(ref_losses[k]))
        if mode == 'f32':
            params = dict(det.named_parameters())
            worst = max((float((params[k].detach().cpu() - otr.sd[k].detach()).abs().max()), k) for k in otr.param_keys)
            assert worst[0] < 5e-5, worst
    finally:
        ops.set_precision(prev)
