"""Training and evaluation drivers without Lightning, for 1..N ranks (what train.py:131-133,221-250 / val.py:83-96 of the reference do with
``pl.Trainer.fit`` / ``pl.Trainer.test``; pytorch_lightning is not part of the MI355X image; INTEGRATION.md section 4 shows the three lines a Lightning user changes).

    module = fetch_model_module(config); data_module = fetch_data_module(config)
    history = fit(config, module, data_module)                    # trains for training.max_steps, validates every validation.val_check_interval
    metrics = run_evaluation(config, module, data_module, 'test')  # val.py

One process per GPU.  ``fit``: every rank iterates its own shard of the training loaders (the data module shards by rank), ``FlatAdamW.step``
all-reduces the flat gradient buffer in buckets (parallel.py), SyncBatchNorm exchanges run inside the head -- nothing here issues a
collective of its own.  ``run_evaluation``: recordings are independent, every rank streams its shard through ``validation_step`` /
``test_step``; the Prophesee records are gathered once at the end and rank 0 computes the KPIs (the same single gather as
``leod_amd.predict.run_pseudo_labeling``)."""
import os
import time
from typing import Any, Callable, Dict, Optional

import torch
import torch.distributed as dist

from leod_amd.modules.utils.detection import Mode
from leod_amd.optim import fit_step

_MODES = {'val': Mode.VAL, 'validate': Mode.VAL, 'test': Mode.TEST}


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _device_batches(loader, module, device):
    if torch.device(device).type == 'cuda':
        from leod_amd.modules.data.prefetch import DevicePrefetcher
        return DevicePrefetcher(loader, module, device)
    return (module.transfer_batch_to_device(b, device, 0) for b in loader)


def run_evaluation(config, module, data_module, mode: str = 'test', device: Optional[torch.device] = None, process_group=None,
                   limit_batches: Optional[int] = None, do_setup: bool = True) -> Optional[Dict[str, float]]:
    """``trainer.validate`` / ``trainer.test`` of the reference (val.py:83-96, train.py validation loop): every batch of the split through
    ``validation_step`` / ``test_step`` (detections buffered as Prophesee records), KPIs over everything buffered at the end.  Returns the
    KPI dictionary (``{'test/AP': ..}``) on every rank; None when the split produced no labelled frame."""
    m = _MODES[mode]
    rank, world = _world(process_group)
    if device is None:
        device = next(module.parameters()).device
    was_training = module.training
    module.eval()
    if do_setup:                                       # (inside ``fit`` the 'fit' stage has set the validation loader up: the TEST split, genx.py:166-170)
        stage = 'validate' if m == Mode.VAL else 'test'
        module.setup(stage)
        data_module.setup(stage)
    loader = data_module.val_dataloader() if m == Mode.VAL else data_module.test_dataloader()
    step = module.validation_step if m == Mode.VAL else module.test_step
    started = module.started_training
    module.started_training = True                     # (the reference skips the sanity-check validation before training has started)
    with torch.no_grad():
        for i, batch in enumerate(_device_batches(loader, module, device)):
            if limit_batches is not None and i >= limit_batches:
                break
            step(batch, i)
    evaluator = module.mode_2_psee_evaluator.get(m)
    mine = dict(labels=evaluator._buffer[evaluator.LABELS] if evaluator is not None else [],
                predictions=evaluator._buffer[evaluator.PREDICTIONS] if evaluator is not None else [],
                hw=module.mode_2_hw[m], batch_size=module.mode_2_batch_size[m])
    if world > 1:
        everyone = [None] * world
        dist.all_gather_object(everyone, mine, group=process_group)
    else:
        everyone = [mine]
    metrics = None
    if rank == 0 and evaluator is not None and any(r['labels'] for r in everyone):
        evaluator.reset_buffer()
        for r in everyone:
            if r['labels']:
                evaluator.add_labels(r['labels'])
                evaluator.add_predictions(r['predictions'])
        hw = next(r['hw'] for r in everyone if r['hw'] is not None)
        raw = evaluator.evaluate_buffer(img_height=hw[0], img_width=hw[1])
        metrics = {f'{mode if mode != "validate" else "val"}/{k}': float(v) for k, v in raw.items()}
    if evaluator is not None:
        evaluator.reset_buffer()
    if world > 1:
        box = [metrics]
        dist.broadcast_object_list(box, src=0, group=process_group)
        metrics = box[0]
    module.started_training = started
    module.train(was_training)
    return metrics


def save_checkpoint(path: str, module, optimizer, scheduler, global_step: int, epoch: int) -> str:
    """A Lightning-shaped checkpoint (``state_dict`` with the ``mdl.`` prefix ``Module.load_weight`` and the reference's own checkpoints
    use, optimiser and scheduler states, ``global_step``, ``epoch``)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)) or '.', exist_ok=True)
    ckpt = {'state_dict': {k: v.detach().cpu() for k, v in module.state_dict().items()},
            'optimizer_states': [optimizer.state_dict()], 'lr_schedulers': [scheduler.state_dict()] if scheduler is not None else [],
            'global_step': int(global_step), 'epoch': int(epoch)}
    torch.save(ckpt, path)
    return path


def fit(config, module, data_module, max_steps: Optional[int] = None, val_check_interval: Optional[int] = None,
        log_every_n_steps: int = 100, ckpt_path: Optional[str] = None, resume_from: Optional[str] = None,
        on_step: Optional[Callable[[int, Dict[str, Any]], None]] = None, device: Optional[torch.device] = None, process_group=None,
        limit_val_batches: Optional[int] = None) -> Dict[str, Any]:
    """``pl.Trainer.fit`` of train.py:221-250 restated as a loop: ``fit_step`` (zero_grad, ``training_step``, backward, value clipping +
    AdamW, OneCycle) per batch until ``max_steps`` (default ``training.max_steps``) or ``training.max_epochs``; validation every
    ``val_check_interval`` steps (default ``validation.val_check_interval``) and at the end; a checkpoint at the end (and after every
    validation) when ``ckpt_path`` is given.  -> {'global_step', 'epochs', 'loss' (per logged step), 'val' [(step, KPIs)], 'step_ms'}."""
    rank, world = _world(process_group)
    if device is None:
        device = next(module.parameters()).device
    max_steps = int(max_steps if max_steps is not None else config.training.max_steps)
    max_epochs = int(config.training.get('max_epochs', 10000))
    if val_check_interval is None:
        val_check_interval = config.validation.get('val_check_interval', None)
    module.setup('fit')
    data_module.setup('fit')
    module.train()
    oc = module.configure_optimizers()
    opt, sched = (oc['optimizer'], oc['lr_scheduler']['scheduler']) if isinstance(oc, dict) else (oc, None)
    step, epoch = 0, 0
    if resume_from:
        # tensors, numbers, strings and containers only (what save_checkpoint writes): no unpickling of arbitrary objects from a path
        ck = torch.load(resume_from, map_location='cpu', weights_only=True)
        module.load_state_dict(ck['state_dict'])
        opt.load_state_dict(ck['optimizer_states'][0])
        if sched is not None and ck.get('lr_schedulers'):
            sched.load_state_dict(ck['lr_schedulers'][0])
        step, epoch = int(ck['global_step']), int(ck['epoch'])
    hist: Dict[str, Any] = {'loss': [], 'val': [], 'step_ms': []}

    def validate():
        kp = run_evaluation(config, module, data_module, 'val', device=device, process_group=process_group, limit_batches=limit_val_batches,
                            do_setup=False)
        hist['val'].append((step, kp))
        module.train()
        if ckpt_path and rank == 0:
            save_checkpoint(ckpt_path, module, opt, sched, step, epoch)
        return kp

    t_last = time.perf_counter()
    done = step >= max_steps
    while not done and epoch < max_epochs:
        loader = data_module.train_dataloader()
        if hasattr(loader, 'set_epoch'):
            loader.set_epoch(epoch)
        n_in_epoch = 0
        # N > 1: every step issues collectives in lock-step (gradient exchange, SyncBatchNorm, validation), so the ranks must take the SAME
        # number of steps per epoch although their shards may hold different batch counts: a sized loader is cut to the smallest length of
        # any rank (one all-reduce per epoch); an unsized (streaming) one agrees per step on "every rank still has a batch"
        cap, per_step_flag = None, False
        if world > 1:
            try:
                n_mine = len(loader)
            except TypeError:
                n_mine = -1
            agree = torch.tensor([n_mine, -n_mine], dtype=torch.int64, device=device if dist.get_backend(process_group) == 'nccl' else 'cpu')
            dist.all_reduce(agree, op=dist.ReduceOp.MIN, group=process_group)     # (min length, -max length)
            if int(agree[0]) >= 0:
                cap = int(agree[0])
            else:
                per_step_flag = True
        batches = iter(_device_batches(loader, module, device))
        while True:
            if cap is not None and n_in_epoch >= cap:
                break
            batch = next(batches, None)
            if per_step_flag:
                flag = torch.tensor([0 if batch is None else 1], dtype=torch.int32, device=device if dist.get_backend(process_group) == 'nccl' else 'cpu')
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=process_group)
                if int(flag) == 0:
                    break
            elif batch is None:
                break
            out = fit_step(module, opt, sched, batch, step)
            step += 1
            n_in_epoch += 1
            if on_step is not None:
                on_step(step, out)
            if log_every_n_steps and step % log_every_n_steps == 0:
                loss = float(out['loss'].detach())                   # the only read-back of the loop, once per logging interval
                now = time.perf_counter()
                hist['loss'].append((step, loss))
                hist['step_ms'].append((step, 1e3 * (now - t_last) / log_every_n_steps))
                t_last = now
            if val_check_interval and step % int(val_check_interval) == 0:
                validate()
            if step >= max_steps:
                done = True
                break
        if hasattr(module, 'on_train_epoch_end'):
            module.on_train_epoch_end()
        epoch += 1
        if n_in_epoch == 0:
            break                                                     # an empty training split
    if not hist['val'] or hist['val'][-1][0] != step:
        validate()
    elif ckpt_path and rank == 0:
        save_checkpoint(ckpt_path, module, opt, sched, step, epoch)
    if world > 1:
        dist.barrier(group=process_group)
    hist.update(global_step=step, epochs=epoch)
    return hist
