"""SURVEY 8(f1) on the device: the worker-process loader (modules/data/process_loader.py) hands batches over through a shared ring
registered with HIP; the frames it delivers in HBM must be the frames the in-process loader reads, and Module.training_step must
accept the batches as they come."""
import numpy as np
import pytest
import torch

from oracle.synth import synth_dataset_tree

pytestmark = pytest.mark.gpu


def _dm(tree, **kw):
    from leod_amd.config import full_config
    from leod_amd.modules.data.genx import DataModule
    cfg = full_config('gen1', 'small', overrides=dict(dataset=dict(path=tree, sequence_length=5, train=dict(sampling='stream'))))
    return DataModule(cfg.dataset, num_workers_train=2, num_workers_eval=2, batch_size_train=4, batch_size_eval=2, prefetch=2,
                      io_threads=1, **kw)


def test_worker_process_loader_delivers_device_frames(tmp_path):
    from leod_amd.data.utils.types import DataType
    from leod_amd.modules.data.process_loader import ProcessLoader
    tree = synth_dataset_tree(str(tmp_path), 'gen1', False)
    dev = torch.device('cuda', 0)

    def run(**kw):
        torch.manual_seed(5); np.random.seed(5)
        dm = _dm(tree, **kw)
        dm.setup('fit')
        out = []
        loader = dm.train_dataloader()
        if not isinstance(loader, ProcessLoader):              # what ProcessLoader does per epoch: one parent draw seeds the worker
            from leod_amd.modules.data.process_loader import draw_base_seed, seed_worker
            seed_worker(draw_base_seed(), 0)
        for batch in loader:
            ev = batch['data'][DataType.EV_REPR]
            out.append((torch.stack([e.cpu() for e in ev]), [bool(x) for x in batch['data'][DataType.IS_FIRST_SAMPLE].cpu()], ev[0].device.type))
        return out, loader

    ref, _ = run()
    got, loader = run(worker_process=True, device=dev, ring_slots=4)
    assert isinstance(loader, ProcessLoader) and loader.registered            # the ring is pinned: the copies are DMA, not staged
    assert len(ref) == len(got) >= 4
    for (a, fa, da), (b, fb, db) in zip(ref, got):
        assert da == 'cpu' and db == 'cuda'
        assert fa == fb and torch.equal(a, b)
