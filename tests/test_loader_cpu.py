"""on-disk formats + streaming / random / mixed loaders.  The sequence classes are checked sample by sample against
vectors recorded from the REFERENCE's own classes (tests/golden/g17_loader.npz, make_golden.py:g17_loader); the batch-level
machinery (slot streams, worker dealing, padding, pinned batch assembly, mixed batches, rank sharding) by its invariants."""
import os

import numpy as np
import pytest
import torch

from oracle.synth import LOADER_RECORDINGS, loader_cases, synth_dataset_tree

EV_NAME = 'stacked_histogram_dt=50_nbins=10'
BBOX_NAMES = ('t', 'x', 'y', 'w', 'h', 'class_id', 'class_confidence', 'objectness')


@pytest.fixture(scope='module')
def trees(tmp_path_factory):
    root = str(tmp_path_factory.mktemp('genx'))
    return {'gen1': synth_dataset_tree(root, 'gen1', False), 'gen4': synth_dataset_tree(root, 'gen4', True)}


def _check_sample(g, key, sample):
    from leod_amd.data.utils.types import DataType
    if DataType.EV_REPR in sample:
        assert np.array_equal(torch.stack(list(sample[DataType.EV_REPR])).numpy(), g[f'{key}_ev']), key
        assert list(sample[DataType.EV_IDX]) == g[f'{key}_idx'].tolist(), key
        assert list(sample[DataType.IS_PADDED_MASK]) == g[f'{key}_pad'].tolist(), key
        flags = [sample[DataType.IS_FIRST_SAMPLE], sample[DataType.IS_LAST_SAMPLE], sample[DataType.IS_REVERSED]]
        assert flags == g[f'{key}_flags'].tolist(), key
        assert os.path.basename(sample[DataType.PATH]) == str(g[f'{key}_path']), key
    else:
        assert f'{key}_ev' not in g
    for name, k in (('lab', DataType.OBJLABELS_SEQ), ('skip', DataType.SKIPPED_OBJLABELS_SEQ)):
        rows, where = [], []
        for t, l in enumerate(sample[k]):
            if l is not None:
                rows.append(l.object_labels.numpy().astype(np.float32))
                where += [t] * len(l)
                assert tuple(float(v) for v in l.input_size_hw) == tuple(g[f'{key}_{name}hw{t}'].tolist()), (key, t)
        got = np.concatenate(rows) if rows else np.zeros((0, 8), np.float32)
        assert where == g[f'{key}_{name}_t'].tolist(), (key, name)
        assert np.array_equal(got, g[f'{key}_{name}']), (key, name)


@pytest.mark.parametrize('dst', ['gen1', 'gen4'])
def test_sequence_samples_match_reference_golden(trees, golden_dir, dst):
    from pathlib import Path
    from leod_amd.data.genx_utils.sequence_rnd import SequenceForRandomAccess
    from leod_amd.data.genx_utils.sequence_streaming import SequenceForIter
    from leod_amd.data.utils.types import DatasetType
    g = np.load(os.path.join(golden_dir, 'g17_loader.npz'))
    ds2 = dst == 'gen4'
    common = dict(ev_representation_name=EV_NAME, dataset_type=DatasetType.GEN4 if ds2 else DatasetType.GEN1,
                  downsample_by_factor_2=ds2, tflip_offset=-2 if ds2 else -1)
    n_samples = n_err = 0
    for ci, (kind, rec, kw, tf, _) in enumerate(loader_cases()):
        if ds2 and ci % 3:
            continue
        path = Path(trees[dst]) / 'train' / rec
        seq = (SequenceForIter(path=path, **common, **kw) if kind == 'iter'
               else SequenceForRandomAccess(path=path, only_load_end_labels=False, **common, **kw))
        seq.time_flip = tf
        assert len(seq) == int(g[f'{dst}_c{ci}_len']), (ci, kind, rec)
        for i in range(len(seq)):
            key = f'{dst}_c{ci}_s{i}'
            np.random.seed(1000 * ci + i)
            if f'{key}_err' in g:
                with pytest.raises(ValueError):
                    seq[i]
                n_err += 1
                continue
            _check_sample(g, key, seq[i])
            # the same sample produced INTO a caller-owned buffer slot (the pinned-batch path) is identical
            np.random.seed(1000 * ci + i)
            buf = np.full((seq.seq_len,) + seq.frame_shape, 255, np.uint8)
            _check_sample(g, key, seq.sample(i, out=buf))
            assert np.array_equal(buf, g[f'{key}_ev'])
            n_samples += 1
    assert n_samples > (30 if ds2 else 100) and n_err <= 2
    for rec in ('rec_a', 'rec_c', 'rec_e'):
        for L in (3, 5, 9):
            subs = SequenceForIter.get_sequences_with_guaranteed_labels(path=Path(trees[dst]) / 'train' / rec, sequence_length=L, **common)
            got = np.asarray([[s.start_indices[0], s.stop_indices[-1], len(s)] for s in subs], dtype=np.int64)
            assert np.array_equal(got, g[f'{dst}_{rec}_L{L}_ranges']), (rec, L)


def test_worker_dealing_matches_reference_golden(golden_dir):
    from leod_amd.data.utils.stream_sharded_datapipe import ShardedStreamingDataPipe
    g = np.load(os.path.join(golden_dir, 'g17_loader.npz'))

    class DP:
        def __init__(self, n, tag):
            self.n, self.tag = n, tag

        def __len__(self):
            return self.n
    lens = [7, 3, 9, 9, 1, 4, 12, 2, 5, 5, 6]
    pipe = ShardedStreamingDataPipe([DP(n, i) for i, n in enumerate(lens)], batch_size=2)
    assert [p.tag for p in pipe.datapipe_list] == g['shard_sorted'].tolist()
    for total in (1, 2, 3, 4, 8):
        seen = []
        for w in range(total):
            got = [p.tag for p in ShardedStreamingDataPipe.assign_datapipes_to_worker(pipe.datapipe_list, total, w)]
            assert got == g[f'shard_w{total}_{w}'].tolist()
            seen += got
        assert sorted(seen) == list(range(len(lens)))


def _cfg(tree, **over):
    from leod_amd.config import full_config
    cfg = full_config('gen1', 'small', overrides=dict(dataset=dict(path=tree, sequence_length=5, **over)))
    return cfg.dataset


def test_eval_stream_covers_every_frame_once_and_shards_by_rank(trees, monkeypatch):
    """ShardedStreamingDataPipe through the data module: every frame of every recording exactly once across (ranks x workers x
    slots), first/last flags at the recording boundaries, dry slots filled with padding samples, frames of a batch in ONE
    [L,B,C,H,W] buffer; with the time-flip TTA copies a recording and its reversed twin are separate streams."""
    from leod_amd.data.utils.stream_sharded_datapipe import ShardedStreamingDataPipe
    from leod_amd.data.utils.types import DataType
    from leod_amd.modules.data.genx import DataModule
    from leod_amd.modules.utils.detection import DATA_KEY, WORKER_ID_KEY
    want = {name: n for name, _, n, _ in LOADER_RECORDINGS}
    for world in (1, 2):
        seen = {}
        for rank in range(world):
            monkeypatch.setattr(ShardedStreamingDataPipe, 'world', staticmethod(lambda r=rank, w=world: (r, w)))
            dm = DataModule(_cfg(trees['gen1'], data_augmentation=dict(stream=dict(start_from_zero=True))), 2, 1, 4, 2, prefetch=2)
            dm.setup('test')
            last_of_slot = {}
            for batch in dm.test_dataloader():
                data, w = batch[DATA_KEY], batch[WORKER_ID_KEY]
                ev = data[DataType.EV_REPR]
                assert len(ev) == 5 and ev[0].shape == (2, 20, 6, 8) and ev[0].dtype == torch.uint8
                assert ev[1].data_ptr() == ev[0].data_ptr() + ev[0].numel()            # views of one batch buffer
                for b, path in enumerate(data[DataType.PATH]):
                    idx = [int(data[DataType.EV_IDX][t][b]) for t in range(5)]
                    pad = [bool(data[DataType.IS_PADDED_MASK][t][b]) for t in range(5)]
                    if not path:
                        assert idx == [-1] * 5 and all(pad) and not bool(data[DataType.IS_FIRST_SAMPLE][b])
                        assert all(int(ev[t][b].sum()) == 0 for t in range(5))
                        continue
                    name = os.path.basename(path)
                    first = bool(data[DataType.IS_FIRST_SAMPLE][b])
                    assert first == (idx[0] == 0)
                    if not first:
                        assert last_of_slot[(rank, w, b)] == (name, idx[0] - 1)           # the slot continues its recording
                    live = [i for i, p in zip(idx, pad) if not p]
                    assert live == list(range(live[0], live[0] + len(live))) and all(i == -1 for i, p in zip(idx, pad) if p)
                    for t, i in enumerate(idx):
                        if i >= 0:
                            assert int(ev[t][b, 0, 0, 0]) == (i + 1) % 251 and int(ev[t][b, 5, 0, 1]) == 6   # frame / channel tags
                    seen.setdefault(name, []).extend(live)
                    last_of_slot[(rank, w, b)] = (name, live[-1])
                    assert bool(data[DataType.IS_LAST_SAMPLE][b]) == (live[-1] == want[name] - 1)
        assert {k: sorted(v) for k, v in seen.items()} == {k: list(range(n)) for k, n in want.items()}, world


def test_train_streams_mixed_batches_and_augmentation_states(trees):
    """Mixed training batches: stream half + random half (modules/data/genx.py:120-144), merged by the module's
    ``merge_mixed_batches``; the random half always restarts the LSTM state, the stream half only at sub-sequence starts; every
    sample carries labels somewhere and its augmentation state (labels already transformed, frames untouched)."""
    from leod_amd.data.utils.types import DataType, DatasetSamplingMode
    from leod_amd.modules.data.genx import DataModule, MixedLoader
    from leod_amd.modules.utils.detection import DATA_KEY, WORKER_ID_KEY, merge_mixed_batches
    torch.manual_seed(3)
    np.random.seed(3)
    dm = DataModule(_cfg(trees['gen1']), num_workers_train=4, num_workers_eval=1, batch_size_train=4, batch_size_eval=2, prefetch=2)
    dm.setup('fit')
    assert dm.sampling_mode_2_train_batch_size == {DatasetSamplingMode.RANDOM: 2, DatasetSamplingMode.STREAM: 2}
    assert dm.sampling_mode_2_train_workers == {DatasetSamplingMode.RANDOM: 2, DatasetSamplingMode.STREAM: 2}
    loader = dm.train_dataloader()
    assert isinstance(loader, MixedLoader)
    n = 0
    cursor = {}
    for batch in loader:
        assert set(batch) == {DatasetSamplingMode.RANDOM, DatasetSamplingMode.STREAM}
        wid = batch[DatasetSamplingMode.STREAM][WORKER_ID_KEY]
        merged = merge_mixed_batches(batch)
        data = merged[DATA_KEY]
        assert merged[WORKER_ID_KEY] == wid
        first = data[DataType.IS_FIRST_SAMPLE]
        assert first.shape == (4,) and bool(first[2]) and bool(first[3])               # random half: always a fresh state
        assert len(data[DataType.AUGM_STATE]) == 4 and len(data[DataType.PATH]) == 4
        assert data[DataType.EV_REPR][0].shape == (4, 20, 6, 8)
        for b in range(4):
            assert any(data[DataType.OBJLABELS_SEQ][t][b] is not None for t in range(5))   # guaranteed labels
        for b in range(2):                                                             # stream half: consecutive chunks
            idx = [int(data[DataType.EV_IDX][t][b]) for t in range(5)]
            rev = bool(data[DataType.IS_REVERSED][b])
            key = (wid, b)
            if not bool(first[b]):
                path, nxt, prev_rev = cursor[key]
                assert path == data[DataType.PATH][b] and rev == prev_rev
                assert [i for i in idx if i >= 0][0] == nxt
            live = [i for i in idx if i >= 0]
            cursor[key] = (data[DataType.PATH][b], (live[-1] - 1) if rev else (live[-1] + 1), rev)
        n += 1
    assert n >= 4


def test_pseudo_label_dataset_round_trip(trees, tmp_path):
    """EventSeqData.save writes the reference's dataset layout (pseudo_labeler.py:335-397); the loaders read it back: the next
    self-training round trains on exactly what the pseudo-labelling round wrote."""
    from pathlib import Path
    from leod_amd.config.dictconfig import DictConfig
    from leod_amd.data.genx_utils.labels import ObjectLabels
    from leod_amd.data.genx_utils.sequence_streaming import SequenceForIter
    from leod_amd.data.utils import misc
    from leod_amd.data.utils.types import DatasetType, DataType
    from leod_amd.modules.pseudo_labeler import EventSeqData
    src = os.path.join(trees['gen1'], 'train', 'rec_a')
    esd = EventSeqData(path=src, scale_ratio=1, filter_config=DictConfig(dict(min_track_len=0, track_method='forward', inpaint=False, ignore_label=1024)),
                       postproc_cfg=DictConfig(dict(confidence_threshold=0.01, nms_threshold=0.45)))
    assert esd.eoe is False
    rows = {3: [[0., 10., 20., 30., 40., 1., 0.8, 0.9]], 7: [[0., 50., 60., 20., 25., 0., 0.7, 0.6], [0., 5., 6., 70., 25., 1., 0.5, 0.95]],
            22: [[1150000., 100., 100., 40., 40., 0., 1., 1.]]}
    frames = sorted(rows)
    labels = [ObjectLabels(torch.tensor(rows[f]), (240, 304)) for f in frames]
    esd.update(labels=labels, ev_idx=frames, is_last_sample=True, is_padded_mask=[False] * 3, is_hflip=False, is_tflip=False, tflip_offset=-1)
    assert esd.eoe is True and esd.aug is False
    save_dir = str(tmp_path / 'gen1_pse' / 'train')
    os.makedirs(save_dir)
    new_seq = esd.save(save_dir, 'gen1')
    ev_dir = misc.get_ev_dir(new_seq)
    assert os.path.islink(os.path.join(ev_dir, 'event_representations.npy')) and os.path.islink(os.path.join(ev_dir, 'event_representations.h5'))
    assert os.path.islink(str(tmp_path / 'gen1_pse' / 'val')) and os.path.islink(str(tmp_path / 'gen1_pse' / 'test'))
    assert os.path.samefile(os.readlink(str(tmp_path / 'gen1_pse' / 'val')), os.path.join(trees['gen1'], 'val'))
    assert misc.read_objframe_idx_2_repr_idx(new_seq).tolist() == frames
    lab, starts = misc.read_npz_labels(new_seq)
    assert starts.tolist() == [0, 1, 3] and lab.dtype.names == BBOX_NAMES and lab['t'].tolist() == [0, 0, 0, 1150000]
    with pytest.raises(FileExistsError):
        esd.save(save_dir, 'gen1')
    seq = SequenceForIter(path=Path(new_seq), ev_representation_name=EV_NAME, sequence_length=5, dataset_type=DatasetType.GEN1,
                          downsample_by_factor_2=False, start_from_zero=True)
    got = {}
    for i in range(len(seq)):
        s = seq[i]
        for t, l in enumerate(s[DataType.OBJLABELS_SEQ]):
            if l is not None:
                got[s[DataType.EV_IDX][t]] = l.object_labels.numpy()
    assert sorted(got) == frames
    for f in frames:
        np.testing.assert_array_equal(got[f], np.asarray(rows[f], np.float32))
    src_seq = SequenceForIter(path=Path(src), ev_representation_name=EV_NAME, sequence_length=5, dataset_type=DatasetType.GEN1,
                              downsample_by_factor_2=False, start_from_zero=True)
    assert torch.equal(torch.stack(seq[1][DataType.EV_REPR]), torch.stack(src_seq[1][DataType.EV_REPR]))       # linked frames


# ---- the pseudo-label pass sharded over ranks (gloo, world size 2, CPU) ----------------------------------------
class _RecordingModule(torch.nn.Module):
    """Stands in for PseudoLabeler where no GPU is available: same driver-facing surface (setup, transfer_batch_to_device,
    predict_step, ev_path_2_ev_data, ev_cnt, evaluator buffer), but ``predict_step`` only records what it was fed."""

    def __init__(self):
        super().__init__()
        from leod_amd.modules.utils.detection import Mode
        from leod_amd.utils.evaluation.prophesee.evaluator import PropheseeEvaluator
        self.w = torch.nn.Parameter(torch.zeros(1))
        self.ev_path_2_ev_data, self.ev_cnt, self.save_dir, self.dst_name = {}, 0, '', 'gen1'
        self.mode_2_psee_evaluator = {Mode.TEST: PropheseeEvaluator(dataset='gen1', downsample_by_2=False)}
        self.mode_2_hw, self.mode_2_batch_size = {Mode.TEST: (240, 304)}, {Mode.TEST: 2}
        self.seen = {}

    def setup(self, stage):
        assert stage == 'predict'

    def transfer_batch_to_device(self, batch, device, dataloader_idx=0):
        return batch

    def predict_step(self, batch, batch_idx=0):
        from leod_amd.data.utils.types import DataType
        from leod_amd.modules.utils.detection import DATA_KEY, Mode
        from leod_amd.utils.evaluation.prophesee.io.box_loading import to_prophesee
        data = batch[DATA_KEY]
        for b, path in enumerate(data[DataType.PATH]):
            if not path:
                continue
            key = (path, bool(data[DataType.IS_REVERSED][b]))
            self.seen.setdefault(key, []).extend(int(data[DataType.EV_IDX][t][b]) for t in range(len(data[DataType.EV_IDX]))
                                                 if int(data[DataType.EV_IDX][t][b]) >= 0)
            if path not in self.ev_path_2_ev_data:
                self.ev_path_2_ev_data[path] = type('E', (), {'eoe': True})()
                self.ev_cnt += 1
            for t in range(len(data[DataType.OBJLABELS_SEQ])):
                lab = data[DataType.OBJLABELS_SEQ][t][b]
                if lab is not None and not key[1]:                   # "predict" the GT itself on the plain view
                    l, p = to_prophesee([lab], [lab])
                    self.mode_2_psee_evaluator[Mode.TEST].add_labels(l)
                    self.mode_2_psee_evaluator[Mode.TEST].add_predictions(p)


def _predict_worker(rank, world, port, tree, q):
    import torch.distributed as dist
    if world > 1:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from leod_amd.config import full_config, dynamically_modify_train_config
    from leod_amd.modules.data.genx import DataModule
    from leod_amd.predict import run_pseudo_labeling
    cfg = dynamically_modify_train_config(full_config('gen1', 'small', model='pseudo_labeler', is_train=False, overrides=dict(
        tta=dict(enable=True, hflip=True, tflip=True),
        dataset=dict(path=tree, sequence_length=5, data_augmentation=dict(stream=dict(start_from_zero=True))))))
    dm = DataModule(cfg.dataset, 2, 1, 4, 2, prefetch=2)
    mod = _RecordingModule()
    out = run_pseudo_labeling(cfg, mod, dm, device=torch.device('cpu'), save=False)
    q.put((rank, {k: sorted(v) for k, v in mod.seen.items()}, out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_pseudo_label_pass_sharded_over_two_ranks_gloo(trees):
    """run_pseudo_labeling with world size 2 (gloo): the union of what the ranks processed is exactly what one rank processes
    (every frame of every recording once per TTA view), a recording and its time-reversed copy meet on ONE rank, nothing is
    processed twice, and the gathered quality KPIs equal the single-rank ones."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    res = {}
    for world in (1, 2):
        q = ctx.Queue()
        port = 35000 + (os.getpid() % 2000)
        procs = [ctx.Process(target=_predict_worker, args=(r, world, port, trees['gen1'], q)) for r in range(world)]
        for p in procs:
            p.start()
        res[world] = sorted((q.get(timeout=240) for _ in range(world)), key=lambda r: r[0])
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
    single = res[1][0]
    assert len(single[1]) == 2 * len(LOADER_RECORDINGS)                            # plain + reversed view of every recording
    assert single[2]['num_sequences'] == len(LOADER_RECORDINGS) and single[2]['metrics'] is not None
    union = {}
    for rank, seen, out in res[2]:
        assert out['num_sequences'] == len(LOADER_RECORDINGS) and sum(out['num_sequences_rank']) == len(LOADER_RECORDINGS)
        assert out['metrics'] == single[2]['metrics']
        paths = {p for p, _ in seen}
        assert {(p, False) for p in paths} | {(p, True) for p in paths} == set(seen)       # both views on this rank
        for k, v in seen.items():
            assert k not in union
            union[k] = v
    assert union == single[1]
    assert min(res[2][0][2]['num_sequences_rank']) >= 1


def _flatten(o, out, prefix=''):
    from leod_amd.data.genx_utils.labels import ObjectLabels, SparselyBatchedObjectLabels
    if torch.is_tensor(o):
        out.append((prefix, o.clone().numpy()))
    elif isinstance(o, ObjectLabels):
        out.append((prefix + '/ol', o.object_labels.clone().numpy())); out.append((prefix + '/hw', np.asarray(o.input_size_hw)))
    elif isinstance(o, SparselyBatchedObjectLabels):
        for i, l in enumerate(o.sparse_object_labels_batch):
            _flatten(l, out, f'{prefix}/sb{i}')
    elif isinstance(o, dict):
        for k, v in o.items():
            _flatten(v, out, f'{prefix}/{k}')
    elif isinstance(o, (list, tuple)):
        for i, v in enumerate(o):
            _flatten(v, out, f'{prefix}/{i}')
    elif o is None or isinstance(o, (str, int, float, bool)):
        out.append((prefix, o))
    else:                                                       # augmentation states: compare by repr of their fields
        out.append((prefix, repr(getattr(o, '__dict__', o))))


@pytest.mark.parametrize('stage', ['fit-stream', 'fit-random', 'fit-mixed', 'test'])
def test_worker_process_loader_yields_the_in_process_batches(trees, stage):
    """modules/data/process_loader.py: batch assembly in a forked worker, frames through the shared ring, labels as numpy over a
    pipe -- the batches must be the ones the in-process loader yields (same RNG state at the start of the epoch), frames of a
    batch still views of ONE [L,B,C,H,W] buffer, and a slot must not be recycled while the consumer may still read it."""
    from leod_amd.data.utils.types import DataType
    from leod_amd.modules.data.genx import DataModule
    from leod_amd.modules.data.process_loader import ProcessLoader

    def run(worker_process):
        torch.manual_seed(11); np.random.seed(11)
        import random; random.seed(11)
        over = dict(train=dict(sampling=stage[4:])) if stage.startswith('fit') else {}
        dm = DataModule(_cfg(trees['gen1'], **over), num_workers_train=4, num_workers_eval=2, batch_size_train=4, batch_size_eval=2,
                        prefetch=2, io_threads=1, worker_process=worker_process, ring_slots=8)
        dm.setup(stage[:3] if stage.startswith('fit') else stage)
        loader = dm.train_dataloader() if stage.startswith('fit') else dm.test_dataloader()
        assert isinstance(loader, ProcessLoader) == worker_process
        if not worker_process:                  # what ProcessLoader.__iter__ / _worker do: one parent draw, worker seeded with it
            from leod_amd.modules.data.process_loader import draw_base_seed, seed_worker
            seed_worker(draw_base_seed(), 0)
        batches, live = [], []
        for batch in loader:
            flat = []
            _flatten(batch, flat)
            batches.append(flat)
            sub = [batch] if 'data' in batch else list(batch.values())
            for s in sub:
                ev = s['data'][DataType.EV_REPR]
                assert ev[1].data_ptr() == ev[0].data_ptr() + ev[0].numel() and ev[0]._base is not None
                assert tuple(ev[0]._base.shape) == (len(ev),) + tuple(ev[0].shape)      # what Module._stack_frames recognises
            live.append((batch, flat))
            if len(live) > 2:
                live.pop(0)
            for b, f in live:                                   # the previous batches' frames have not been overwritten
                g = []
                _flatten(b, g)
                assert all(np.array_equal(x[1], y[1]) if isinstance(x[1], np.ndarray) else x[1] == y[1] for x, y in zip(f, g))
        return batches

    ref, got = run(False), run(True)
    assert len(ref) == len(got) and len(ref) >= 4
    for fr, fg in zip(ref, got):
        if stage == 'fit-mixed':            # two producer threads share the global RNG: the draws interleave differently from run to
            assert [(k, getattr(a, 'shape', None)) for k, a in fr if 'EV_REPR' in k] == \
                   [(k, getattr(a, 'shape', None)) for k, a in fg if 'EV_REPR' in k]       # run also in process; structure only
            continue
        assert [k for k, _ in fr] == [k for k, _ in fg]
        for (k, a), (_, b) in zip(fr, fg):
            if isinstance(a, np.ndarray):
                assert a.dtype == b.dtype and np.array_equal(a, b), k
            else:
                assert a == b, k


def test_worker_process_loader_surfaces_worker_errors():
    from leod_amd.modules.data.process_loader import ProcessLoader

    def broken():
        yield {'data': {}, 'worker_id': 0}
        raise ValueError('recording 7 is truncated')

    it = iter(ProcessLoader(broken, slot_bytes=1024, n_slots=2))
    assert next(it) == {'data': {}, 'worker_id': 0}
    with pytest.raises(RuntimeError, match='recording 7 is truncated'):
        next(it)


def test_worker_process_loader_reseeds_every_epoch():
    """a forked worker inherits the parent's RNG state; without a per-epoch base seed every epoch replayed the same
    shuffles / augmentation draws.  Epochs must differ, a re-seeded parent must reproduce them, and ranks must differ."""
    from leod_amd.modules.data.process_loader import ProcessLoader

    def gen():
        yield {'perm': torch.randperm(10), 'np': np.random.randint(0, 1 << 30, 4), 'worker_id': 0}

    def epochs(rank):
        torch.manual_seed(5)
        loader = ProcessLoader(gen, slot_bytes=1024, n_slots=2, rank=rank)
        return [next(iter(loader)) for _ in range(3)]

    a, b, c = epochs(0), epochs(0), epochs(1)
    perms = [tuple(e['perm'].tolist()) for e in a]
    assert len(set(perms)) == 3, 'every epoch replayed the same draws'
    assert len({tuple(e['np'].tolist()) for e in a}) == 3
    assert perms == [tuple(e['perm'].tolist()) for e in b], 'a seeded parent must reproduce its epochs'
    assert perms[0] != tuple(c[0]['perm'].tolist()), 'ranks must not draw the same stream'


def test_training_loaders_are_rank_aware(trees):
    """under N > 1 ranks the random-access loader must partition ONE shared order (DistributedSampler semantics: no
    sample twice per epoch across ranks, every rank the same count) and the streaming loader must not replay the same shuffle on
    every rank when all ranks were seeded identically (bench.py does torch.manual_seed(0) everywhere)."""
    from leod_amd.data.utils.types import DataType
    from leod_amd.modules.data.genx import DataModule, RandomLoader

    def build(rank, world, sampling):
        torch.manual_seed(0); np.random.seed(0)                 # identical user seed on every rank
        dm = DataModule(_cfg(trees['gen1'], train=dict(sampling=sampling)), num_workers_train=2, num_workers_eval=1,
                        batch_size_train=2, batch_size_eval=1, prefetch=0, io_threads=1, rank=rank, world_size=world)
        dm._seed_shared = 1234                                  # what shared_seed() broadcasts from rank 0
        dm.setup('fit')
        return dm.train_dataloader()

    # random access: the two ranks' index orders are disjoint halves of one permutation
    loaders = [build(r, 2, 'random') for r in range(2)]
    assert all(isinstance(l, RandomLoader) for l in loaders)
    orders = [l._order() for l in loaders]
    n = len(loaders[0].dataset)
    assert len(orders[0]) == len(orders[1]) == -(-n // 2)
    both = orders[0] + orders[1]
    assert set(both) == set(range(n)) and len(both) - n in (0, 1)            # padded by at most one wrapped sample
    assert len(loaders[0]) == len(orders[0]) // 2
    nxt = [l._order() for l in loaders]                          # next epoch: another shared permutation
    assert nxt[0] != orders[0] and set(nxt[0] + nxt[1]) == set(range(n))
    one = build(0, 1, 'random')
    assert len(one._order()) == n                                # single rank: everything, global RNG

    # streaming: identically seeded ranks draw different shuffles (the process RNG is re-seeded base + rank)
    def first_paths(rank):
        loader = build(rank, 2, 'stream')
        out = []
        for i, batch in enumerate(loader):
            out.append((tuple(batch['data'][DataType.PATH]), tuple(int(e) for e in batch['data'][DataType.EV_IDX][0])))
            if i == 5:
                break
        return out
    a, b, a2 = first_paths(0), first_paths(1), first_paths(0)
    assert a == a2, 'a seeded rank must reproduce its stream'
    assert a != b, 'two identically seeded ranks replayed the same stream order'


def test_random_loader_is_deterministic_with_a_thread_pool_and_frame_stores_are_bounded(trees):
    """(1) samples of one recording share an augmentor; with the reads on a thread pool the drawn state must be
    private to its sample -- the batches of a seeded run with 8 I/O threads equal those with one thread; (2) sequence objects do
    not own file descriptors: many of them over the same files keep at most ``FRAME_STORES.capacity`` stores open."""
    from leod_amd.data.utils import misc
    from leod_amd.data.utils.types import DataType
    from leod_amd.modules.data.genx import DataModule

    def run(io_threads):
        torch.manual_seed(3); np.random.seed(3)
        dm = DataModule(_cfg(trees['gen1'], train=dict(sampling='random')), num_workers_train=2, num_workers_eval=1,
                        batch_size_train=4, batch_size_eval=1, prefetch=0, io_threads=io_threads)
        dm.setup('fit')
        out = []
        for i, batch in enumerate(dm.train_dataloader()):
            flat = []
            _flatten(batch, flat)
            out.append(flat)
            if i == 3:
                break
        return out

    a, b = run(1), run(8)
    assert len(a) == len(b) == 4
    for fa, fb in zip(a, b):
        assert [k for k, _ in fa] == [k for k, _ in fb]
        for (k, x), (_, y) in zip(fa, fb):
            assert (np.array_equal(x, y) if isinstance(x, np.ndarray) else x == y), k
    assert any('AUGM_STATE' in k for k, _ in a[0])

    old = misc.FRAME_STORES
    try:
        misc.FRAME_STORES = misc.FrameStoreCache(capacity=2)
        dm = DataModule(_cfg(trees['gen1'], train=dict(sampling='stream')), num_workers_train=2, num_workers_eval=1,
                        batch_size_train=4, batch_size_eval=1, prefetch=0, io_threads=1)
        dm.setup('fit')
        n_seq = len(dm.sampling_mode_2_dataset[next(iter(dm.sampling_mode_2_dataset))].datapipe_list)
        assert n_seq > 2 and len(misc.FRAME_STORES) == 0         # building the sequences opened nothing for keeps
        for i, batch in enumerate(dm.train_dataloader()):
            if i == 6:
                break
        assert 1 <= len(misc.FRAME_STORES) <= 2
    finally:
        misc.FRAME_STORES = old


def test_random_loader_epoch_survives_recreation():
    """under N > 1 ranks the shuffle of epoch e is seeded seed + e, but a loader that is re-created every epoch (the
    forked worker of ProcessLoader builds its loaders anew per epoch; a trainer may call train_dataloader() per epoch) started at
    epoch 0 each time -- every epoch replayed one permutation.  The creator now hands the epoch in."""
    import inspect
    from leod_amd.modules.data import genx as G
    from leod_amd.modules.data import process_loader as PL

    class DS:                                            # the two attributes RandomLoader.__init__ / _order touch
        class _S:
            class sequence:
                seq_len, frame_shape = 2, (20, 4, 4)
        datasets = [_S]
        def __len__(self):
            return 64

    def order(epoch, recreate):
        ld = G.RandomLoader(DS(), batch_size=4, rank=1, world_size=2, seed=5, epoch=epoch if recreate else 0, pin_memory=False)
        if not recreate:
            for _ in range(epoch):
                ld._order()
        return ld._order()
    e0, e1, e2 = order(0, True), order(1, True), order(2, True)
    assert e0 != e1 != e2 and e0 != e2                   # a re-created loader no longer replays epoch 0
    assert e1 == order(1, False) and e2 == order(2, False)      # and matches what a persistent loader does in that epoch
    # the plumbing: ProcessLoader counts its iterations and its worker passes the count to loader_fn(epoch=...)
    seen = []
    assert PL._call_loader_fn(lambda epoch=0: seen.append(epoch) or [], 3) == [] and seen == [3]
    assert PL._call_loader_fn(lambda: ['x'], 7) == ['x']
    assert 'epoch' in inspect.signature(G.DataModule._train_loaders).parameters
    src = inspect.getsource(PL.ProcessLoader.__iter__)
    assert 'self.epoch + 1' in src and 'epoch)' in src


def test_read_helpers_of_a_recording(tmp_path):
    """``misc.read_ev_repr`` / ``read_labels_as_list`` (data/utils/misc.py:28-46,83-88 of the reference) on a synthetic recording: all frames as one
    array; labels placed at their frame positions inside a window, None elsewhere."""
    from types import SimpleNamespace
    from oracle.synth import LOADER_RECORDINGS, synth_dataset_tree
    from leod_amd.data.utils import misc
    tree = synth_dataset_tree(str(tmp_path / 'src'), 'gen1', False, frame_hw=(6, 8))
    name, _, n_frames, lab = LOADER_RECORDINGS[0]
    seq = os.path.join(tree, 'train', name)
    ev = misc.read_ev_repr(seq)
    assert ev.dtype == np.uint8 and ev.shape == (n_frames, 20, 6, 8)
    o2r = misc.read_objframe_idx_2_repr_idx(seq).tolist()
    cfg = SimpleNamespace(ev_repr_hw=(240, 304), downsample_by_factor_2=False)
    full = misc.read_labels_as_list(seq, cfg, L=n_frames)
    assert len(full) == n_frames and [i for i, l in enumerate(full) if l is not None] == o2r
    labels, starts = misc.read_npz_labels(seq)
    assert sum(len(l) for l in full if l is not None) == len(labels)
    lo = o2r[1]
    window = misc.read_labels_as_list(seq, cfg, L=3, start_idx=lo)
    assert window[0] is not None and len(window) == 3 and [i for i, l in enumerate(window) if l is not None] == [r - lo for r in o2r if lo <= r < lo + 3]
    np.testing.assert_allclose(window[0].x.numpy() if hasattr(window[0].x, 'numpy') else window[0].x, full[lo].x)


class _FakeH5Dataset:
    """The slice of h5py.Dataset that H5Frames uses (shape, __getitem__, read_direct) over an in-memory array."""

    def __init__(self, arr):
        self._a = arr
        self.shape, self.dtype = arr.shape, arr.dtype
        self.reads = 0

    def __getitem__(self, s):
        self.reads += 1
        return self._a[s].copy()

    def read_direct(self, dest, source_sel=None):
        self.reads += 1
        np.copyto(dest, self._a[source_sel])


def test_h5_frames_agree_with_raw_frames_through_a_stand_in_h5py(tmp_path, monkeypatch):
    """``misc.H5Frames`` (the reader of the reference's container, sequence_base.py:184-193: ``h5py.File(fn, 'r')['data'][a:b]``) executed through
    a stand-in ``h5py`` module (h5py is not in this image): every read path of H5Frames and RawFrames returns the same bytes, the frame-store
    cache and ``open_ev_repr`` / ``read_frame_header`` / ``read_ev_repr`` pick the HDF5 file when no raw twin exists, and a streaming sequence
    built over the HDF5 file yields the samples of the one built over the twin."""
    import sys
    import types
    from leod_amd.data.utils import misc
    tree = synth_dataset_tree(str(tmp_path / 'src'), 'gen1', False, frame_hw=(6, 8))
    name, _, n_frames, _ = LOADER_RECORDINGS[0]
    seq = os.path.join(tree, 'train', name)
    raw_fn = misc.get_ev_raw_fn(seq)
    h5_fn = misc.get_ev_h5_fn(seq)
    frames = np.load(raw_fn)
    opened = []

    class File:
        def __init__(self, fn, mode='r'):
            assert mode == 'r' and str(fn).endswith('.h5') and os.path.exists(fn)
            self.ds = _FakeH5Dataset(frames)
            opened.append(self)
            self.closed = False

        def __getitem__(self, key):
            assert key == 'data'
            return self.ds

        def close(self):
            self.closed = True

        def __enter__(self):
            return self

        def __exit__(self, *a):
            self.close()

    fake = types.ModuleType('h5py')
    fake.File = File
    monkeypatch.setitem(sys.modules, 'h5py', fake)
    raw, h5 = misc.RawFrames(raw_fn), misc.H5Frames(h5_fn)
    assert h5.shape == raw.shape == frames.shape and len(h5) == len(raw) == n_frames
    for a, b in ((0, 1), (2, 7), (0, n_frames), (n_frames - 1, n_frames)):
        assert np.array_equal(h5.read(a, b), raw.read(a, b))
        out_h, out_r = np.empty((b - a,) + frames.shape[1:], np.uint8), np.empty((b - a,) + frames.shape[1:], np.uint8)
        assert h5.read(a, b, out_h) is out_h and np.array_equal(out_h, raw.read(a, b, out_r))
    h5.close()
    assert opened[0].closed
    # dispatch: with the twin present the raw file wins; without it the HDF5 file is opened
    assert isinstance(misc.open_ev_repr(seq), misc.RawFrames)
    os.rename(raw_fn, raw_fn + '.away')
    try:
        st = misc.open_ev_repr(seq)
        assert isinstance(st, misc.H5Frames) and np.array_equal(st.read(1, 4), frames[1:4])
        st.close()
        assert misc.read_frame_header(h5_fn) == (n_frames, frames.shape[1:])
        assert np.array_equal(misc.read_ev_repr(seq), frames)
        misc.FRAME_STORES.clear()
        assert isinstance(misc.FRAME_STORES.get(h5_fn), misc.H5Frames)
        misc.FRAME_STORES.clear()
    finally:
        os.rename(raw_fn + '.away', raw_fn)


@pytest.mark.parametrize('compression', ['blosc', 'blosc-zlib', 'gzip', 'shuffle-gzip', None, 'contiguous'])
def test_h5lite_reads_the_reference_container(tmp_path, compression):
    """``leod_amd.data.utils.h5lite`` -- the package's own reader of the reference's container (classic HDF5 layout, dataset ``data``
    [N,20,H,W] uint8 chunked frame by frame, blosc-zstd filter 32001: sequence_base.py:184-193, utils/preprocessing.py:4-15) -- on files built
    by ``tests/h5_writer.py`` from the HDF5 / c-blosc specifications (no libhdf5 here): every read path, a multi-level chunk B-tree, missing
    chunks, other dtypes and filters; then ``misc.H5Frames`` / ``open_ev_repr`` / a streaming sequence over such a file without a raw twin."""
    from h5_writer import write_h5, blosc_frame
    from leod_amd.data.utils import h5lite, misc
    rng = np.random.RandomState(3)
    frames = ((rng.rand(150, 20, 6, 8) < 0.08) * rng.randint(1, 10, (150, 20, 6, 8))).astype(np.uint8)
    fn = str(tmp_path / 'event_representations.h5')
    if compression == 'contiguous':
        write_h5(fn, frames, chunks=None)
    else:
        write_h5(fn, frames, chunks=(1, 20, 6, 8), compression=compression, istore_k=4)      # 150 chunks, 8 per node: a three-level index
    with h5lite.H5File(fn) as f:
        assert f.keys() == ['data'] and 'data' in f
        d = f['data']
        assert d.shape == frames.shape and d.dtype == np.uint8 and len(d) == 150
        assert (d.chunks == (1, 20, 6, 8)) == (compression != 'contiguous')
        assert np.array_equal(d[:], frames) and np.array_equal(d[17:43], frames[17:43]) and np.array_equal(d[149], frames[149])
        assert d[5:5].shape == (0, 20, 6, 8)
        out = np.empty((9, 20, 6, 8), np.uint8)
        d.read_direct(out, np.s_[100:109])
        assert np.array_equal(out, frames[100:109])
    # the frame-store interface of the loaders on top of it (h5py is absent: H5Frames takes h5lite)
    st = misc.H5Frames(fn)
    assert st.shape == frames.shape and np.array_equal(st.read(3, 11), frames[3:11])
    buf = np.empty((4, 20, 6, 8), np.uint8)
    assert st.read(60, 64, buf) is buf and np.array_equal(buf, frames[60:64])
    st.close()
    assert misc.read_frame_header(fn) == (150, (20, 6, 8))


def test_h5lite_other_dtypes_missing_chunks_and_refusals(tmp_path):
    from h5_writer import write_h5, blosc_frame
    from leod_amd.data.utils import h5lite
    rng = np.random.RandomState(4)
    b = rng.randn(33, 4, 5).astype(np.float32)
    fn = write_h5(str(tmp_path / 'f.h5'), b, chunks=(4, 4, 5), compression='blosc', missing=(2,))
    with h5lite.H5File(fn) as f:
        want = b.copy()
        want[8:12] = 0                                             # an unallocated chunk reads as the fill value
        assert f['data'].dtype == np.float32 and np.array_equal(f['data'][:], want) and np.array_equal(f['data'][7:13], want[7:13])
    i16 = rng.randint(-3000, 3000, (10, 7)).astype(np.int16)
    with h5lite.H5File(write_h5(str(tmp_path / 'i.h5'), i16, chunks=(3, 7), compression='shuffle-gzip')) as f:
        assert f['data'].dtype == np.int16 and np.array_equal(f['data'][:], i16)
    # blosc frames: split streams with byte shuffle (typesize 4), stored blocks, several blocks
    raw = b.tobytes() * 20
    for kw in (dict(split=True, blocksize=4096), dict(split=False, blocksize=1000), dict(codec='zlib')):
        assert h5lite.blosc_decompress(blosc_frame(raw, 4, kw.pop('codec', 'zstd'), True, **kw)) == raw
    noise = rng.bytes(5000)
    assert h5lite.blosc_decompress(blosc_frame(noise, 1)) == noise                 # incompressible: stored streams
    with pytest.raises(KeyError):
        h5lite.H5File(fn)['nope']
    bad = tmp_path / 'bad.h5'
    bad.write_bytes(b'not an hdf5 file' * 64)
    with pytest.raises(IOError):
        h5lite.H5File(str(bad))
    v2 = bytearray(open(fn, 'rb').read())
    v2[8] = 2                                                      # superblock version 2 (libver='latest'): refused by name
    (tmp_path / 'v2.h5').write_bytes(bytes(v2))
    with pytest.raises(NotImplementedError, match='superblock version 2'):
        h5lite.H5File(str(tmp_path / 'v2.h5'))
