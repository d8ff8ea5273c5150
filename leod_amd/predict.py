"""Pseudo-label generation driver (what predict.py:167-246 of the reference does with ``pl.Trainer.predict``), for 1..N ranks.

    module = fetch_model_module(config)            # PseudoLabeler
    summary = run_pseudo_labeling(config, module, data_module)

The reference asserts a single GPU (predict.py:167-169).  Recordings are independent, so N ranks simply shard them -- one
process per GPU, no collective on the data path: ``ShardedStreamingDataPipe`` keeps a recording and its time-reversed TTA copy on
one rank, every rank runs ``PseudoLabeler.predict_step`` over its own stream and writes its own recordings
(``EventSeqData.save``).  Only the end of the run exchanges anything: the per-rank counts and the detection records of the frames
whose GT was withheld (pseudo_labeler.py:756-763) are gathered with ``all_gather_object`` and rank 0 computes the quality KPIs."""
import os
from typing import Any, Dict, Optional

import torch
import torch.distributed as dist

from leod_amd.modules.utils.detection import Mode


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def run_pseudo_labeling(config, module, data_module, device: Optional[torch.device] = None, save: bool = True,
                        process_group=None, prefetch_to_device: bool = True) -> Dict[str, Any]:
    """Returns (on every rank) {'num_sequences': total, 'num_sequences_rank': [...], 'metrics': KPI dict | None,
    'saved': [paths written by THIS rank]}."""
    rank, world = _world(process_group)
    if device is None:
        device = next(module.parameters()).device
    module.eval()
    module.setup('predict')
    data_module.setup('predict')
    loader = data_module.predict_dataloader()
    if prefetch_to_device and torch.device(device).type == 'cuda':
        from leod_amd.modules.data.prefetch import DevicePrefetcher
        batches = DevicePrefetcher(loader, module, device)
    else:
        batches = (module.transfer_batch_to_device(b, device, 0) for b in loader)
    pipelined = hasattr(module, 'flush_predictions') and torch.device(device).type == 'cuda'
    if pipelined:
        module.pipelined = True                # host bookkeeping of chunk i - 1 under the device work of chunk i
    with torch.no_grad():
        for i, batch in enumerate(batches):
            module.predict_step(batch, i)
        if pipelined:
            module.flush_predictions()
            module.pipelined = False
    saved = []
    if save and module.save_dir:
        if rank == 0:
            os.makedirs(module.save_dir, exist_ok=True)
        if world > 1:
            dist.barrier(group=process_group)
        for ev_data in module.ev_path_2_ev_data.values():
            assert ev_data.eoe, 'some data are not evaluated in full sequence'
            saved.append(ev_data.save(save_dir=module.save_dir, dst_name=module.dst_name))
    # ---- end of run: one small gather ----------------------------------------------------------------------------------
    evaluator = module.mode_2_psee_evaluator.get(Mode.TEST)
    mine = dict(ev_cnt=module.ev_cnt, paths=sorted(module.ev_path_2_ev_data),
                labels=evaluator._buffer[evaluator.LABELS] if evaluator is not None else [],
                predictions=evaluator._buffer[evaluator.PREDICTIONS] if evaluator is not None else [],
                hw=module.mode_2_hw[Mode.TEST], batch_size=module.mode_2_batch_size[Mode.TEST])
    if world > 1:
        everyone = [None] * world
        dist.all_gather_object(everyone, mine, group=process_group)
    else:
        everyone = [mine]
    all_paths = [p for r in everyone for p in r['paths']]
    assert len(all_paths) == len(set(all_paths)), 'a recording was processed by more than one rank'
    metrics = None
    if rank == 0 and evaluator is not None and any(r['labels'] for r in everyone):
        evaluator.reset_buffer()
        for r in everyone:
            if r['labels']:
                evaluator.add_labels(r['labels'])
                evaluator.add_predictions(r['predictions'])
        hw = next(r['hw'] for r in everyone if r['hw'] is not None)
        metrics = evaluator.evaluate_buffer(img_height=hw[0], img_width=hw[1])
    if world > 1:
        box = [metrics]
        dist.broadcast_object_list(box, src=0, group=process_group)
        metrics = box[0]
    return {'num_sequences': sum(r['ev_cnt'] for r in everyone), 'num_sequences_rank': [r['ev_cnt'] for r in everyone],
            'metrics': metrics, 'saved': saved}
