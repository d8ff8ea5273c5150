"""The optimiser object ``Module.configure_optimizers`` hands to the trainer (reference: modules/detection.py:485-518
returns ``torch.optim.AdamW`` + ``OneCycleLR``; train.py:236-237 clips gradients by value 1.0 before the step).

``FlatAdamW`` is a ``torch.optim.Optimizer`` -- Lightning (or any hand-written loop) drives it through the usual
``zero_grad() / step(closure)`` calls and LR schedulers see ``param_groups[0]['lr']`` -- but its state is the flat
buffers of ``leod_amd.parallel.FlatParams``:

* ``zero_grad``  one memset of the flat gradient buffer (the wgrad kernels accumulate into views of it; the views are
  never replaced by ``None``),
* ``step``       join the weight-gradient side stream -> complete the gradient sum over RCCL when the job has more than one rank
  (five per-stage buckets whose all-reduces were started DURING the backward pass, ``leod_amd.parallel.GradBuckets``) -> ONE
  ``leod_adamw_clip_step`` launch (value-clip + 1/world scaling + AdamW fused).

The data-parallel exchange lives here, not in a DistributedDataParallel wrapper: the autograd Functions of this package
write parameter gradients straight into the flat buffer (21 timesteps accumulate in place) and return ``None`` to
autograd, so DDP's per-parameter hooks would never fire.  See INTEGRATION.md ("N > 1").
"""
from typing import Optional

import torch

from . import ops
from .functions import WgradSide, flush_bn_counters
from .parallel import DataParallel, FlatParams


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, module: torch.nn.Module, lr=2e-4, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8,
                 clip_value: Optional[float] = None, process_group=None, sync_bn: bool = True,
                 flat: Optional[FlatParams] = None, grad_buckets: bool = True):
        self.module = module
        self.flat = flat if flat is not None else FlatParams(module)
        self.dp = DataParallel(self.flat, process_group, sync_bn=sync_bn)
        self.dp.broadcast_parameters()
        self.dp.make_buckets(module, bucketed=grad_buckets)      # per-stage gradient buckets when the job has more than one rank
        self.clip_value = clip_value
        super().__init__(self.flat.params, dict(lr=lr, weight_decay=weight_decay, betas=betas, eps=eps))

    @property
    def world_size(self) -> int:
        return self.dp.world_size

    def zero_grad(self, set_to_none: bool = False) -> None:      # noqa: ARG002 -- the .grad views must survive
        self.flat.zero_grad()
        self.dp.begin_step()                                     # the backward pass that follows releases the gradient buckets

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        WgradSide.active = False
        WgradSide.join()                                         # parameter gradients are complete on the launch stream
        scale = self.dp.all_reduce_gradients()
        g = self.param_groups[0]
        self.flat.adamw_step(float(g['lr']), g['weight_decay'], self.clip_value or 0.0, grad_scale=scale,
                             betas=tuple(g['betas']), eps=g['eps'])
        ops.StatArena.end_step()
        flush_bn_counters(self.module)
        return loss

    # ---- checkpointing (Lightning stores optimizer.state_dict() in the .ckpt) ----------------------------------------------
    def state_dict(self):
        groups = [{k: v for k, v in g.items() if k != 'params'} for g in self.param_groups]
        return {'state': {'step': self.flat.step_count, 'exp_avg': self.flat.exp_avg.clone(),
                          'exp_avg_sq': self.flat.exp_avg_sq.clone()}, 'param_groups': groups}

    def load_state_dict(self, state_dict) -> None:
        st = state_dict['state']
        self.flat.step_count = int(st['step'])
        self.flat.exp_avg.copy_(st['exp_avg'])
        self.flat.exp_avg_sq.copy_(st['exp_avg_sq'])
        for g, saved in zip(self.param_groups, state_dict['param_groups']):
            g.update({k: v for k, v in saved.items() if k != 'params'})


def fit_step(module, optimizer, scheduler, batch, batch_idx: int = 0):
    """What Lightning's automatic optimisation does with one batch (optimizer.step(closure); closure = zero_grad ->
    training_step -> backward; then the per-step LR scheduler) -- the Lightning-free driver used by bench.py and tests."""
    out = {}

    def closure():
        optimizer.zero_grad()
        out.update(module.training_step(batch, batch_idx))
        module.backward(out['loss'])
        return out['loss']

    optimizer.step(closure)
    if scheduler is not None:
        scheduler.step()
    return out
