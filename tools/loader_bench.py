#!/usr/bin/env python
"""SURVEY 8(f1) acceptance: the training step fed by the real loader path (memory-mapped recordings on disk -> pinned
[L,B,20,H,W] batches -> one PCIe copy on a side stream -> Module.training_step) against the HBM-resident rate of bench.py.

    python tools/loader_bench.py [--root /dev/shm/leod_synth] [--steps 20]

Writes a synthetic Gen1-sized dataset (8 recordings x 260 frames of 20x240x304 uint8, sparse counts, labels at 4 Hz-like
spacing) unless it exists, then times: (a) the loader alone, (b) training with loader + DevicePrefetcher, (c) training with a
resident batch."""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch


def make_dataset(root, n_rec=8, n_frames=260):
    from oracle.synth import BBOX_DTYPE
    base = os.path.join(root, 'gen1')
    if os.path.exists(os.path.join(base, 'train', f'rec_{n_rec - 1}', 'labels_v2', 'labels.npz')):
        return base
    rng = np.random.RandomState(0)
    for split, n in (('train', n_rec), ('val', 1), ('test', 1)):
        for r in range(n):
            seq = os.path.join(base, split, f'rec_{r}')
            ev_dir = os.path.join(seq, 'event_representations_v2', 'stacked_histogram_dt=50_nbins=10')
            os.makedirs(ev_dir, exist_ok=True); os.makedirs(os.path.join(seq, 'labels_v2'), exist_ok=True)
            nf = n_frames if split == 'train' else 42
            fr = np.lib.format.open_memmap(os.path.join(ev_dir, 'event_representations.npy'), mode='w+', dtype=np.uint8, shape=(nf, 20, 240, 304))
            for lo in range(0, nf, 20):
                blk = fr[lo:lo + 20]
                blk[:] = (rng.rand(*blk.shape) < 0.08) * rng.randint(1, 10, size=blk.shape)
            fr.flush(); del fr
            labelled = list(range(4, nf, 5))
            rows, starts = [], []
            for f in labelled:
                starts.append(len(rows))
                for _ in range(rng.randint(1, 7)):
                    w, h = rng.uniform(10, 90), rng.uniform(10, 70)
                    rows.append(((f + 1) * 50000, rng.uniform(0, 303 - w), rng.uniform(0, 239 - h), w, h, rng.randint(0, 2), 1.0, 1.0))
            np.save(os.path.join(ev_dir, 'objframe_idx_2_repr_idx.npy'), np.asarray(labelled, dtype=np.int64))
            np.savez(os.path.join(seq, 'labels_v2', 'labels.npz'), labels=np.array(rows, dtype=BBOX_DTYPE),
                     objframe_idx_2_label_idx=np.asarray(starts, dtype=np.int64))
    return base


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--root', default='/dev/shm/leod_synth')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--io-threads', type=int, default=8)
    ap.add_argument('--loader-only', action='store_true')
    args = ap.parse_args()
    t0 = time.time()
    base = make_dataset(args.root)
    print(f'dataset at {base} ({time.time() - t0:.1f} s)')
    from leod_amd.config import full_config, dynamically_modify_train_config
    from leod_amd.modules.data.genx import DataModule
    from leod_amd.modules.data.prefetch import DevicePrefetcher
    from leod_amd.modules.utils.fetch import fetch_model_module
    from leod_amd.optim import fit_step
    cfg = dynamically_modify_train_config(full_config('gen1', 'small', overrides=dict(dataset=dict(path=base))))
    dm = DataModule(cfg.dataset, num_workers_train=cfg.hardware.num_workers.train, num_workers_eval=2,
                    batch_size_train=cfg.batch_size.train, batch_size_eval=cfg.batch_size.eval, prefetch=4, io_threads=args.io_threads)
    dm.setup('fit')
    T, B = cfg.dataset.sequence_length, cfg.batch_size.train
    # (a) loader alone
    n, t0 = 0, time.perf_counter()
    for batch in dm.train_dataloader():
        n += 1
        if n == args.steps:
            break
    dt = time.perf_counter() - t0
    print(f'loader alone : {n * T * B / dt:9.1f} event-frames/s ({1e3 * dt / n:.1f} ms/batch, {n * T * B * 20 * 240 * 304 / dt / 1e9:.2f} GB/s of voxels)')
    dmh = DataModule(cfg.dataset, num_workers_train=cfg.hardware.num_workers.train, num_workers_eval=2,
                     batch_size_train=cfg.batch_size.train, batch_size_eval=cfg.batch_size.eval, prefetch=4, io_threads=args.io_threads,
                     worker_process=True)
    dmh.setup('fit')
    n, t0 = 0, None
    for batch in dmh.train_dataloader():
        if n == 2:
            t0 = time.perf_counter()
        n += 1
        if n == args.steps + 2:
            break
    dt = time.perf_counter() - t0
    print(f'loader in a worker process, alone : {(n - 2) * T * B / dt:9.1f} event-frames/s ({1e3 * dt / (n - 2):.1f} ms/batch)')
    if args.loader_only:
        return
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    module = fetch_model_module(cfg).to(dev)
    module.setup('fit'); module.train()
    oc = module.configure_optimizers()
    opt, sched = oc['optimizer'], oc['lr_scheduler']['scheduler']

    def timed(batches):
        k, t0 = 0, None
        for batch in batches:
            if k == args.warmup:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            fit_step(module, opt, sched, batch, k)
            k += 1
            if k == args.warmup + args.steps:
                break
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (k - args.warmup)

    def epochs(loader_fn):
        while True:
            yield from loader_fn()

    ms_loader = 1e3 * timed(DevicePrefetcher(epochs(dm.train_dataloader), module, dev))
    print(f'loader -> pinned -> side-stream copy -> training_step: {T * B / ms_loader * 1e3:9.1f} event-frames/s ({ms_loader:.2f} ms/step)')
    dmp = DataModule(cfg.dataset, num_workers_train=cfg.hardware.num_workers.train, num_workers_eval=2,
                     batch_size_train=cfg.batch_size.train, batch_size_eval=cfg.batch_size.eval, prefetch=4, io_threads=args.io_threads,
                     worker_process=True, device=dev)
    dmp.setup('fit')
    ms_proc = 1e3 * timed(epochs(dmp.train_dataloader))
    print(f'worker process -> registered shared ring -> side-stream copy -> training_step: {T * B / ms_proc * 1e3:9.1f} event-frames/s ({ms_proc:.2f} ms/step)')
    ms_inline = 1e3 * timed(module.transfer_batch_to_device(b, dev, 0) for b in epochs(dm.train_dataloader))
    print(f'loader -> pinned -> copy on the launch stream        : {T * B / ms_inline * 1e3:9.1f} event-frames/s ({ms_inline:.2f} ms/step)')
    it = iter(dm.train_dataloader())
    resident = module.transfer_batch_to_device(next(it), dev, 0)
    import copy

    def same():
        while True:
            b = dict(resident)
            yield copy.copy(resident)
    # labels are consumed (augmented in place) by a step: rebuild the label containers from a pristine copy every step
    from leod_amd.data.utils.types import DataType, DatasetSamplingMode
    pristine = copy.deepcopy({k: {kk: (vv if kk == 'worker_id' else {a: b for a, b in vv.items() if a != DataType.EV_REPR}) for kk, vv in v.items()}
                              for k, v in resident.items()})

    def resident_batches():
        while True:
            out = {}
            for k, v in resident.items():
                data = dict(copy.deepcopy(pristine[k]['data']))
                data[DataType.EV_REPR] = v['data'][DataType.EV_REPR]
                out[k] = {'data': data, 'worker_id': v['worker_id']}
            yield out
    ms_res = 1e3 * timed(resident_batches())
    print(f'HBM-resident batch (same module, same step)           : {T * B / ms_res * 1e3:9.1f} event-frames/s ({ms_res:.2f} ms/step)')
    print(f'loader-fed / resident = {ms_res / ms_loader:.3f} (loader thread in the training process), {ms_res / ms_proc:.3f} (loader in a worker process)')


if __name__ == '__main__':
    main()
