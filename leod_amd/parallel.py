"""Data-parallel plumbing for the LEOD training step on MI355X: one process per GPU, RCCL over xGMI.

* ``FlatParams``: every trainable parameter becomes a 16-byte-aligned view into ONE fp32 buffer and every
  ``.grad`` a view into ONE gradient buffer.  The wgrad kernels accumulate straight into it, the
  optimiser is one ``leod_adamw_clip_step`` launch, zeroing is one memset.
* ``GradBuckets``: the gradient exchange.  The training step runs STAGE-major (``RNNDetector.forward_sequence``), so
  the backward pass finishes head / PAFPN -> stage 4 -> 3 -> 2 -> 1 and the parameters of a stage are final as soon
  as that stage's backward (and its side-stream weight-gradient kernels) are done -- parameter counts scale with C^2,
  so ~98 % of the 39.5 MB (RVT-S) are final before the backward pass of stage 1, the longest, even starts.  The flat
  buffer is laid out in that module order, so each of the five buckets is ONE contiguous slice: it is all-reduced on a
  communication stream the moment its boundary node fires in the backward pass (``functions.BucketBoundaryFn``), under
  the rest of the backward pass; the optimiser waits for the five handles.  ``make_buckets(bucketed=False)`` (``FlatAdamW(..., grad_buckets=False)``) falls back to one
  flat all-reduce after the backward pass (needed for gradient accumulation: the buckets assume ONE backward pass per optimiser
  step); ``LEOD_DP_WIRE=bf16`` all-reduces a bf16 copy of each bucket -- half the bytes, but the cross-rank sum itself is then carried
  out and rounded in bf16 at every ring step (log2(world) bits of the gradient sum are lost; off by default).
* SyncBatchNorm statistics go through ``functions.set_sync_batchnorm`` (reference: train.py:247).
* Pseudo-labelling shards whole recordings over ranks with no collective on the data path
  (``shard_sequences``; reference: data/utils/stream_sharded_datapipe.py:40-57,88-105).
"""
import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from . import ops
from . import functions as Fn
from .comm import NativeComm

ALIGN = 4   # floats (16 bytes): weight rows are read with 16-byte loads


class FlatParams:
    import weakref as _weakref
    live = _weakref.WeakSet()          # instances with a registered weight shadow (see ``ensure_all_shadows``)

    def __init__(self, module: torch.nn.Module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        dev = self.params[0].device
        offs, n = [], 0
        for p in self.params:
            offs.append(n)
            n += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.numel = n
        self.data = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, offs):
            k = p.numel()
            self.data[o:o + k].copy_(p.detach().reshape(-1))
            p.data = self.data[o:o + k].view(p.shape)
            p.grad = self.grad[o:o + k].view(p.shape)
        self.offsets = offs
        self.step_count = 0
        # bf16 shadow of the parameters for the Linear kernels of precision mode bf16 (csrc/k_misc.hip): registered with the library for
        # as long as this object lives; made fresh by ``ensure_shadow`` at the start of a step, stale by the AdamW kernel / ``touch``
        self.shadow = None
        self._param_versions = None
        if dev.type == 'cuda':
            import weakref
            self.shadow = torch.empty(n, dtype=torch.bfloat16, device=dev)
            self.shadow_f16 = torch.empty(n, dtype=torch.float16, device=dev)     # written / read in precision mode 16f only (forward GEMMs)
            ops.set_weight_shadow(self.data, self.shadow)
            ops.set_weight_shadow_f16(self.data, self.shadow_f16)
            FlatParams.live.add(self)
            weakref.finalize(self, ops.unset_weight_shadow_ptr, self.data.data_ptr())

    def zero_grad(self):
        self.grad.zero_()

    @classmethod
    def ensure_all_shadows(cls) -> None:
        """Every forward entry point of a model whose parameters may live in a flat buffer calls this (``RNNDetector.forward`` /
        ``forward_sequence``): a shadow marked fresh must not outlive an in-place edit of the weights made outside the optimiser
        (``load_state_dict``, ``p.copy_``, EMA) on a path that does not go through ``Module._run_sequence`` -- ``PseudoLabeler`` drives the
        backbone itself, and so does any direct ``mdl(...)`` call.  Not while a step is being recorded (the recording starts with a forced
        refresh and pins the shadows)."""
        if not cls.live or torch.cuda.is_current_stream_capturing():
            return
        for fp in list(cls.live):
            fp.ensure_shadow()

    def adamw_step(self, lr, weight_decay=0.0, clip_value=1.0, grad_scale=1.0, betas=(0.9, 0.999), eps=1e-8, hp_dev=None):
        """value-clip + AdamW (reference: train.py:236-237 gradient_clip_val=1.0 by value; detection.py:485-488).
        With ``hp_dev`` (device float[4], see ``step_scalars``) the per-step scalars come from device memory."""
        if hp_dev is None:
            self.step_count += 1
        ops.adamw_clip_step(self.data, self.grad, self.exp_avg, self.exp_avg_sq, lr, max(self.step_count, 1), betas=betas,
                            eps=eps, weight_decay=weight_decay, clip_value=clip_value, grad_scale=grad_scale, hp_dev=hp_dev)
        ops.PackCache.invalidate()                     # the kernel rewrote every parameter: packed weight copies are stale
        # (the library marked the bf16 shadow stale itself: leod_adamw_clip_step knows the buffer it rewrote)

    def touch(self):
        """Call after editing ``self.data`` (or any parameter through another alias than the parameter itself) outside the optimiser:
        cached derived copies of the weights (``ops.PackCache``, the bf16 shadow) are dropped."""
        ops.PackCache.invalidate()
        if self.shadow is not None:
            ops.weight_shadow_invalidate()

    def ensure_shadow(self, force: bool = False) -> None:
        """Start of a step: the bf16 shadow is rounded from the parameters if it is stale -- after an optimiser step or ``touch()``, or when
        torch's version counters say a parameter was edited in place since the last look (``load_state_dict``, ``p.copy_``).  ``force``:
        always (a step that is being recorded starts with the refresh, so a replay never depends on these flags)."""
        if self.shadow is None:
            return
        ver = self.data._version
        for p in self.params:
            ver += p._version
        if ver != self._param_versions:
            self._param_versions = ver
            ops.weight_shadow_invalidate()
        ops.weight_shadow_refresh(force)

    def step_scalars(self, lr, grad_scale=1.0, betas=(0.9, 0.999)):
        """Advance the step counter and return [lr, 1-b1^t, sqrt(1-b2^t), grad_scale] for the graph-replayed optimiser."""
        self.step_count += 1
        t = self.step_count
        return [float(lr), 1.0 - betas[0] ** t, (1.0 - betas[1] ** t) ** 0.5, float(grad_scale)]


def one_cycle_lr(step, max_lr, total_steps, pct_start=0.005, div_factor=20, final_div_factor=10000):
    """OneCycleLR(linear, no momentum cycling) with the reference's convention final_lr = max_lr/final_div_factor
    (modules/detection.py:498-511)."""
    initial_lr = max_lr / div_factor
    min_lr = max_lr / final_div_factor
    end1 = float(pct_start * total_steps) - 1
    end2 = total_steps - 1
    if step <= end1:
        return (max_lr - initial_lr) * (step / end1) + initial_lr
    return (min_lr - max_lr) * ((step - end1) / (end2 - end1)) + max_lr


class DataParallel:
    """Gradient averaging across ranks for a FlatParams buffer (torch.distributed 'nccl' == RCCL on ROCm,
    'gloo' in the CPU tests)."""

    def __init__(self, flat: Optional[FlatParams], process_group=None, sync_bn: bool = True):
        self.flat = flat
        self.group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        # LEOD_FORCE_COLLECTIVES=1: issue every collective of the N > 1 path even with one rank (exercises RCCL --
        # communicator, all-reduce of the flat gradient, SyncBN exchanges, stream ordering -- on a single-GPU box)
        self.force = os.environ.get('LEOD_FORCE_COLLECTIVES') == '1' and dist.is_available() and dist.is_initialized()
        if (self.world_size > 1 or self.force) and flat is not None and flat.data.is_cuda:
            NativeComm.setup(process_group)                    # collective; every rank ends up on the same path (comm.py)
        if sync_bn and (self.world_size > 1 or self.force):
            Fn.set_sync_batchnorm(process_group, self.world_size)
            Fn._SYNC_BN['force'] = self.force

    def broadcast_parameters(self, src=0):
        if self.world_size > 1 or self.force:
            dist.broadcast(self.flat.data, src=src, group=self.group)
            self.flat.touch()                                  # parameter memory was rewritten behind the parameters' backs: drop packed copies

    def make_buckets(self, module: torch.nn.Module, bucketed: bool = True) -> Optional['GradBuckets']:
        """Per-stage buckets for ``module`` (None: one rank, or ``bucketed=False`` -> one flat all-reduce after the backward pass)."""
        self.buckets = None
        if (self.world_size > 1 or self.force) and bucketed:
            self.buckets = GradBuckets(self.flat, module, self)
        return self.buckets

    def begin_step(self):
        b = getattr(self, 'buckets', None)
        if b is not None:
            b.begin_step()

    def all_reduce_gradients(self) -> float:
        """Complete the gradient sum over ranks (bucketed exchanges already in flight, or one flat all-reduce); returns the scale
        (1/world) the optimiser kernel applies."""
        b = getattr(self, 'buckets', None)
        if b is not None and GradBuckets.current is b:
            b.finish()
        elif self.world_size > 1 or self.force:
            if NativeComm.usable(self.flat.grad, self.group):
                NativeComm.all_reduce(self.flat.grad)
            else:
                dist.all_reduce(self.flat.grad, group=self.group)
        return 1.0 / self.world_size


class GradBuckets:
    """Contiguous per-stage slices of a ``FlatParams`` gradient buffer, all-reduced as the backward pass releases them.

    Bucket k < n_stages = parameters of ``backbone.stages.k``; the last bucket = everything else (PAFPN + head).  A bucket is
    released by ``ready(k)`` -- called from the backward of the boundary node in front of stage k (``functions.bucket_boundary``) or
    of the features that enter the PAFPN -- and by ``finish()`` for whatever is left (stage 1 has no differentiable input, hence
    no boundary).  All ranks build the same autograd graph, so they release the buckets in the same order."""
    current: Optional['GradBuckets'] = None       # the instance the boundary nodes of the running step report to

    def __init__(self, flat: FlatParams, module: torch.nn.Module, dp: 'DataParallel'):
        import re
        self.flat, self.dp = flat, dp
        names = [n for n, p in module.named_parameters() if p.requires_grad]
        assert len(names) == len(flat.params)
        ids = []
        for n in names:
            m = re.search(r'backbone\.stages\.(\d+)\.', n)
            ids.append(int(m.group(1)) if m else -1)
        self.n_stages = max(ids) + 1 if ids and max(ids) >= 0 else 0
        ids = [i if i >= 0 else self.n_stages for i in ids]
        assert all(a <= b for a, b in zip(ids, ids[1:])), 'parameters are not ordered stage by stage: buckets would not be contiguous'
        ends = flat.offsets[1:] + [flat.numel]
        self.ranges = []
        for k in range(self.n_stages + 1):
            idx = [j for j, b in enumerate(ids) if b == k]
            self.ranges.append((flat.offsets[idx[0]], ends[idx[-1]]) if idx else None)
        self.head = self.n_stages
        self.wire_bf16 = os.environ.get('LEOD_DP_WIRE', 'f32').lower() in ('bf16', 'bfloat16')
        self.comm = torch.cuda.Stream(device=flat.grad.device) if flat.grad.is_cuda else None
        self.works, self.done, self.order = [], set(), []

    def begin_step(self):
        self.works, self.done, self.order = [], set(), []
        GradBuckets.current = self

    def ready(self, k: int, closing: bool = False):
        """Bucket k is final on the launch stream and on the weight-gradient side stream(s): all-reduce it on the comm stream.
        One backward pass per optimiser step: a boundary that fires a second time (gradient accumulation, two losses) would add local
        gradients to a slice that is already summed over ranks or still in flight -- refused loudly (use the flat exchange there: ``grad_buckets=False``)."""
        if k in self.done and not closing and self.ranges[k] is not None:
            GradBuckets.current = None
            raise RuntimeError('GradBuckets: a second backward pass reached the boundary of gradient bucket %d before the optimiser step; '
                               'gradient accumulation needs the flat exchange (FlatAdamW(grad_buckets=False))' % k)
        if k in self.done or self.ranges[k] is None:
            self.done.add(k)
            return
        self.done.add(k)
        self.order.append(k)
        lo, hi = self.ranges[k]
        g = self.flat.grad[lo:hi]
        if self.comm is None:                                  # CPU tensors (gloo tests)
            self.works.append((dist.all_reduce(g, group=self.dp.group, async_op=True), None, None))
            return
        main = torch.cuda.current_stream()
        self.comm.wait_stream(main)
        for key in Fn.WgradSide.used:                          # this stage's weight gradients were enqueued there
            self.comm.wait_stream(Fn.WgradSide.streams[key])
        native = NativeComm.usable(g, self.dp.group)          # enqueued on the comm stream itself: nothing to wait for but the stream
        with torch.cuda.stream(self.comm):
            if self.wire_bf16:
                wire = g.to(torch.bfloat16)
                if native:
                    NativeComm.all_reduce(wire, self.comm)
                self.works.append((None if native else dist.all_reduce(wire, group=self.dp.group, async_op=True), wire, g))
            elif native:
                NativeComm.all_reduce(g, self.comm)
                self.works.append((None, None, None))
            else:
                self.works.append((dist.all_reduce(g, group=self.dp.group, async_op=True), None, None))

    def finish(self):
        """Release what no boundary released (stage 1), then make the launch stream wait for every exchange."""
        try:
            for k in reversed(range(len(self.ranges))):
                self.ready(k, closing=True)
            for work, wire, g in self.works:
                if self.comm is not None:
                    with torch.cuda.stream(self.comm):
                        if work is not None:
                            work.wait()
                        if wire is not None:
                            g.copy_(wire)
                else:
                    work.wait()
            if self.comm is not None:
                torch.cuda.current_stream().wait_stream(self.comm)
        finally:
            self.works = []
            GradBuckets.current = None                          # never left set behind a step that raised


def init_distributed(backend: Optional[str] = None, local_device: Optional[int] = None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torchrun contract).  ``local_device`` overrides LOCAL_RANK
    (functional tests of the N > 1 path on one GPU: every rank on device 0 over gloo)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if local_device is not None:
        local = int(local_device)
    if (world > 1 or os.environ.get('LEOD_FORCE_COLLECTIVES') == '1') and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get('LEOD_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard_sequences(lengths: Sequence[int], world_size: int, rank: int, keys: Optional[Sequence] = None) -> List[int]:
    """Deal whole recordings to ranks: sort by length (descending) and assign in 'pyramid' (boustrophedon) order so
    every rank gets a similar number of frames -- the order the reference uses for dataloader workers
    (stream_sharded_datapipe.py:40-57).  ``keys`` (e.g. the recording path) groups entries that must stay together:
    a recording and its time-flipped copy share a key and land on the same rank."""
    n = len(lengths)
    if keys is None:
        keys = list(range(n))
    groups = {}
    for i, k in enumerate(keys):
        groups.setdefault(k, []).append(i)
    items = sorted(groups.items(), key=lambda kv: (-sum(lengths[i] for i in kv[1]), str(kv[0])))
    mine: List[int] = []
    for pos, (_, idxs) in enumerate(items):
        rnd, slot = divmod(pos, world_size)
        owner = slot if rnd % 2 == 0 else world_size - 1 - slot
        if owner == rank:
            mine.extend(idxs)
    return sorted(mine)
