"""Launch plans for ``Module.training_step``: the host's answer to launch overhead on the product path.

The reference's answer is ``torch.compile(mode='reduce-overhead')`` = CUDA graphs on the backbone (config/model/maxvit_yolox/
default.yaml:8-11, modules/detection.py:43-44; static shapes asserted at :176-179,196-199).  Here the unit is the whole training step
of one static shape: the SECOND time ``training_step`` sees a batch geometry (event tensor shape, number of labelled frames, padded
label count) its forward pass -- LSTM-row reset, stage-major backbone over the L frames, PAFPN + head + SimOTA + losses -- and its
backward pass are stream-captured ONCE (two captures sharing a private memory pool; nothing executes during capture), turned into
two ``ops.LaunchPlan`` objects (csrc/k_plan.hip: the captured kernels replayed as plain stream launches from one C loop, ~3 us each, the
weight-gradient fork / join on the plan's own side stream), and from then on a step of that geometry is

    copy-in of (frames, labels, row indices, is_first)  ->  forward plan  ->  [loss.backward()]  backward plan  ->  optimiser

with ~10 Python-level launches instead of ~600.  The first occurrence of a geometry runs eagerly (it is the warm-up a capture needs
and a real training step), any geometry that is never repeated stays eager, and so does everything a capture cannot hold: more
than one rank with SyncBatchNorm (collectives between the kernels), or a graph node the plan cannot replay (reported once).

Measured (RVT-S Gen1 bs 8 L 21, bf16 mode, profiles/r04_*): hipGraphLaunch of the same captures costs 8 us of host time per node and
loses the side stream; the plans make the step independent of the host (16.5 ms on a host where the eager step is 17.7 ms).
"""
import os
import warnings
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch as th
from torch.autograd import Function

from leod_amd import functions as Fn
from leod_amd import ops
from leod_amd._lib import LeodHipError

LOSS_KEYS = ('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss', 'num_fg')
NMAX_PAD = 8          # label rows are padded to a multiple of this (all-zero rows are "no box" to SimOTA / the loss, yolo_head.py:437-441)


def _flat(states) -> List[th.Tensor]:
    return [t for pair in states for t in pair]


class PlanRecorder:
    """Stream capture of a step in SEGMENTS.  Whatever cannot live inside a capture -- a collective between two kernels (the SyncBatchNorm
    statistic exchanges, the release of a gradient bucket to the communication stream) -- calls ``PlanRecorder.split(callback)``: the
    capture in progress is closed and becomes one launch plan, the callback is remembered, a new capture begins in the same memory
    pool.  A replay is  plan 0 -> callback 0 -> plan 1 -> callback 1 ... : the callbacks run eagerly on tensors of the pool (static
    addresses), exactly where the eager step would have issued them.  Captures use the relaxed error mode: the backward pass runs on
    autograd's device thread, which therefore ends / begins captures that the calling thread started."""
    current: Optional['PlanRecorder'] = None

    def __init__(self, max_lanes: int, pool=None):
        self.max_lanes, self.pool = max_lanes, pool
        self.segments: List[Tuple[Optional[ops.LaunchPlan], Any]] = []
        self.graph = None

    def begin(self):
        self.graph = th.cuda.CUDAGraph(keep_graph=True)
        if self.pool is None:
            self.pool = th.cuda.graph_pool_handle()          # one private pool for every segment of the step (forward and backward)
        self.graph.capture_begin(pool=self.pool, capture_error_mode='relaxed')

    def end(self, callback=None):
        Fn.WgradSide.join()                                  # every forked stream back on the capturing stream before the capture ends
        self.graph.capture_end()
        try:
            plan = ops.LaunchPlan(self.graph, self.max_lanes)
        except LeodHipError as e:
            if 'empty graph' not in str(e):
                raise
            plan = None                                      # two callbacks back to back: nothing was launched in between
        self.segments.append((plan, callback))
        self.graph = None

    @classmethod
    def split(cls, callback) -> bool:
        """Called where the step needs the host: True = a recording is in progress and took the callback (do NOT run it now)."""
        rec = cls.current
        if rec is None or rec.graph is None:
            return False
        rec.end(callback)
        rec.begin()
        return True

    def replay(self, join_last: bool = True):
        """``join_last=False``: the last segment's side lanes are not joined into the current stream (``join()`` does it later)."""
        last = len(self.segments) - 1
        for k, (plan, callback) in enumerate(self.segments):
            if plan is not None:
                plan.launch(join=join_last or k < last or callback is not None)
            if callback is not None:
                callback()

    def rebase_input(self, captured: th.Tensor, new: th.Tensor) -> int:
        """-> number of input-reading kernels over all segments now reading ``new`` instead of ``captured``"""
        return sum(p.rebase_input(captured, new) for p, _ in self.segments if p is not None)

    def join(self):
        plan = self.segments[-1][0] if self.segments else None
        if plan is not None:
            plan.join()

    def info(self):
        k = sum(p.info['kernels'] for p, _ in self.segments if p is not None)
        return {'kernels': k, 'memsets': sum(p.info['memsets'] for p, _ in self.segments if p is not None),
                'memcpys': sum(p.info['memcpys'] for p, _ in self.segments if p is not None),
                'lanes': max([p.info['lanes'] for p, _ in self.segments if p is not None] or [0]),
                'waits': sum(p.info['waits'] for p, _ in self.segments if p is not None),
                'segments': len(self.segments), 'callbacks': sum(1 for _, c in self.segments if c is not None),
                'collectives': sum(p.info.get('collectives', 0) for p, _ in self.segments if p is not None)}

    def close(self):
        for p, _ in self.segments:
            if p is not None:
                p.close()
        self.segments = []


class PlanLossFn(Function):
    """The loss of a replayed step as an autograd leaf-to-loss edge: ``forward`` has already happened (the forward plans), ``backward``
    hands the seed gradient to the captured backward passes and launches them.  Parameter gradients land in the flat gradient buffer as
    in the eager step (the captured weight-gradient kernels accumulate into the same addresses).  The plans hold ONE step's activations:
    a second forward of the same geometry before this backward (two losses summed, ``retain_graph`` replays) would differentiate the first
    loss with the second step's activations -- refused loudly."""

    @staticmethod
    def forward(ctx, bb, hd, anchor, loss):
        ctx.bb, ctx.hd, ctx.gen = bb, hd, bb.uses
        return loss.view_as(loss)

    @staticmethod
    def backward(ctx, g):
        bb, hd = ctx.bb, ctx.hd
        if bb.uses != ctx.gen or bb.consumed == ctx.gen or hd.fwd is None or bb.fwd is None:
            raise RuntimeError('leod_amd launch plans: backward() of a planned training step whose activations are gone (another '
                               'training_step of the same geometry ran before this backward, backward ran twice, or the plan was evicted); '
                               'run such steps with Module.plan_mode = False')
        bb.consumed = ctx.gen
        hd.run_backward(g)
        bb.run_backward()
        hd.bwd.join()
        return None, None, None, None


class EagerHeadGate(Function):
    """Planned backbone, EAGER head (``module.plan_head_eager``, an option for N > 1): the outputs are the labelled frames' stage features gathered from the backbone
    plan's static outputs; everything after them -- PAFPN, head, SimOTA, losses and their backward, with the SyncBatchNorm exchanges and the
    head's gradient buckets issued between the kernels as in an eager step -- is ordinary autograd.  ``backward`` receives the gradient of the
    gathered rows, puts it into the backbone plan's static slots and launches the captured backbone backward.  (A captured head pays for every
    collective with a plan segment: a join of the side lanes and a relaunch from the host; the eager head's ~170 launches are enqueued while
    the GPU is still running the backbone's forward plan.)"""

    @staticmethod
    def forward(ctx, bb, anchor, rows, *stages):
        ctx.bb, ctx.gen, ctx.n, ctx.stages = bb, bb.uses, int(rows.shape[0]), stages
        return tuple(bb.x_out[k].index_select(0, rows) for k in stages)

    @staticmethod
    def backward(ctx, *grads):
        bb = ctx.bb
        if bb.uses != ctx.gen or bb.consumed == ctx.gen or bb.fwd is None:
            raise RuntimeError('leod_amd launch plans: backward() of a planned training step whose activations are gone (another '
                               'training_step of the same geometry ran before this backward, backward ran twice, or the plan was evicted); '
                               'run such steps with Module.plan_mode = False')
        bb.consumed = ctx.gen
        dst, src = [], []
        for k, g in zip(ctx.stages, grads):
            if g is None:
                bb.gsel[k][:ctx.n].zero_()
            else:
                dst.append(bb.gsel[k][:ctx.n])
                src.append(g.contiguous())
        if dst:
            ops.copy_multi(dst, src)
        bb.run_backward()
        return (None, None, None) + (None,) * len(ctx.stages)


class HeadPlan:
    """PAFPN + head + SimOTA + losses and their backward for ONE labelled-frame count B' (and padded label width): the data-dependent part
    of the step (modules/detection.py:209-224 -- the reference gathers the labelled frames' features per step; its static-shape unit is
    the backbone, config/model/maxvit_yolox/default.yaml:8-11).  Reads the backbone plan's static stage outputs through the static row-index
    buffer, writes the gradient of the gathered rows into the backbone plan's static gradient slots.  Cheap to hold (one small private
    pool shared by all head plans of a backbone plan; they never run concurrently and keep nothing across steps)."""

    def __init__(self, bb: 'BackbonePlan', n_frames: int, nmax: int):
        dev = bb.ev.device
        self.bb, self.n_frames = bb, n_frames
        self.labels = th.zeros((n_frames, nmax, 7), dtype=th.float32, device=dev)
        self.seed = th.ones((), dtype=th.float32, device=dev)
        self.arena_high = 0
        self.bn_incs: List[Tuple[Any, int]] = []
        self.losses6: Optional[th.Tensor] = None
        self.fwd: Optional['PlanRecorder'] = None
        self.bwd: Optional['PlanRecorder'] = None
        self.pin: Optional[th.Tensor] = None
        self.pin_event = None

    def capture(self, module, wgrad_side: bool, max_lanes: int):
        bb, mdl = self.bb, module.mdl
        in_features = tuple(mdl.fpn.in_features)
        mods = [m for m in mdl.modules() if hasattr(m, 'bn_calls_pending')]
        pending0 = [m.bn_calls_pending for m in mods]
        saved_arena = ops.StatArena.swap(bb.head_arena)
        from leod_amd.parallel import GradBuckets
        saved_buckets = GradBuckets.current
        saved_side = Fn.WgradSide.active
        ops.PackCache.invalidate()                         # the capture must contain the weight packs of the head
        capture_stream = th.cuda.Stream(device=bb.ev.device)
        capture_stream.wait_stream(th.cuda.current_stream())
        th.cuda.synchronize()
        fwd = PlanRecorder(max_lanes, pool=bb.head_pool)
        bwd = None
        try:
            with th.enable_grad(), th.cuda.stream(capture_stream):
                PlanRecorder.current = fwd
                GradBuckets.current = saved_buckets
                fwd.begin()
                bb.head_pool = fwd.pool
                ops.StatArena.begin_step(bb.ev.device)
                rows = bb.rows[:self.n_frames]
                sel = {k: bb.x_out[k].index_select(0, rows).requires_grad_() for k in in_features}
                _, losses = mdl.forward_detect(backbone_features={k: Fn.as_nchw(v) for k, v in sel.items()}, targets=self.labels)
                self.losses6 = mdl.yolox_head.last_losses6
                fwd.end()
                bwd = PlanRecorder(max_lanes, pool=fwd.pool)
                PlanRecorder.current = bwd
                bwd.begin()
                Fn.WgradSide.active = wgrad_side
                try:
                    losses['loss'].backward(gradient=self.seed)
                finally:
                    Fn.WgradSide.active = False
                # the gradient of the gathered rows -> the backbone plan's static slots (first B' frames; its backward adds them into the stage
                # outputs' gradients through the static row-index buffer)
                ops.copy_multi([bb.gsel[k][:self.n_frames] for k in in_features], [sel[k].grad for k in in_features])
                bwd.end()
            self.arena_high = ops.StatArena.high
            self.fwd, self.bwd = fwd, bwd
        except BaseException:
            _abort_captures(capture_stream, fwd, bwd)
            raise
        finally:
            PlanRecorder.current = None
            th.cuda.current_stream().wait_stream(capture_stream)
            ops.StatArena.swap(*saved_arena)
            GradBuckets.current = saved_buckets
            Fn.WgradSide.active = saved_side
            ops.PackCache.invalidate()                     # nothing was executed: the pack buffers do not hold what the cache believes
            self.bn_incs = [(m, m.bn_calls_pending - p0) for m, p0 in zip(mods, pending0) if m.bn_calls_pending != p0]
            for m, p0 in zip(mods, pending0):
                m.bn_calls_pending = p0
        return self

    def stage_labels(self, labels_host_or_dev: th.Tensor):
        n, nmax = labels_host_or_dev.shape[0], labels_host_or_dev.shape[1]
        if labels_host_or_dev.is_cuda:
            if nmax != self.labels.shape[1]:
                self.labels.zero_()
            self.labels[:, :nmax].copy_(labels_host_or_dev)
        else:
            # host labels: padded on the host, ONE asynchronous copy from a pinned staging buffer (reused once its copy has completed)
            if self.pin is None:
                self.pin = th.zeros(self.labels.shape, dtype=th.float32).pin_memory()
            elif self.pin_event is not None:
                self.pin_event.synchronize()
            if nmax != self.labels.shape[1]:
                self.pin.zero_()
            self.pin[:, :nmax].copy_(labels_host_or_dev)
            self.labels.copy_(self.pin, non_blocking=True)
            self.pin_event = th.cuda.Event()
            self.pin_event.record()

    def run_forward(self):
        if self.arena_high:
            self.bb.head_arena[:self.arena_high].zero_()   # BatchNorm statistic accumulators of the captured head pass
        self.fwd.replay()
        for m, inc in self.bn_incs:
            m.bn_calls_pending += inc

    def run_backward(self, g: th.Tensor):
        # the head's weight gradients (side lane) feed nothing the backbone's backward pass reads: the current stream goes on without waiting
        # for them, PlanLossFn joins them after the backbone plan has been launched
        self.seed.copy_(g.reshape(()), non_blocking=True)
        self.bwd.replay(join_last=False)

    def close(self):
        for p in (self.fwd, self.bwd):
            if p is not None:
                p.close()
        self.fwd = self.bwd = None


def _abort_captures(capture_stream, *recs):
    with th.cuda.stream(capture_stream):                  # leave no stream in capture mode behind (a capturing graph aborts in its destructor)
        for rec in recs:
            if rec is not None and rec.graph is not None:
                try:
                    Fn.WgradSide.join()
                    rec.graph.capture_end()
                except Exception:                             # noqa: BLE001
                    pass
            if rec is not None:
                rec.close()


class BackbonePlan:
    """The static-shape unit of the step, keyed by the event tensor's shape alone (the reference's unit too: ``torch.compile`` of the
    backbone, shapes asserted at modules/detection.py:176-179,196-199): LSTM-row reset + stage-major backbone over the L frames, and its
    backward pass.  The data-dependent part -- which frames carry labels, how many -- enters through two static buffers: ``rows`` (frame
    indices t * B + b of the labelled frames, padded with -1 to T * B entries) read by the head plans' gathers and by this plan's backward
    (``ops.rows_index_add`` skips negative indices), and ``gsel`` (the gradient of the gathered rows per PAFPN input stage, written by the
    head plan's backward).  One captured backbone serves every step of its geometry, whatever B' is."""

    def __init__(self, key, module, ev: th.Tensor, states_like):
        dev = ev.device
        self.key = key
        self.ev = th.empty_like(ev)                        # the buffer the plans are captured with
        self.ev_now = self.ev                              # the tensor the input-reading kernels read on the next replay
        self.rebase_ok, self.rebase_count = False, 0
        self.n_max = int(ev.shape[0] * ev.shape[1])
        self.rows = th.full((self.n_max,), -1, dtype=th.long, device=dev)
        self.is_first = th.ones((ev.shape[1],), dtype=th.bool, device=dev)
        self.states = [(th.zeros_like(h), th.zeros_like(c)) for h, c in states_like]      # strides preserved (NCHW views of NHWC rows)
        self.arena = th.zeros(ops.StatArena.SIZE, dtype=th.uint8, device=dev)             # private scratch arena of this plan
        self.head_arena = th.zeros(ops.StatArena.SIZE, dtype=th.uint8, device=dev)        # ... and of its head plans (one runs at a time)
        self.head_pool = None
        self.arena_high = 0
        self.x_out: Dict[int, th.Tensor] = {}              # stage -> its output rows [T*B, H, W, C] inside this plan's pool (static)
        self.gsel: Dict[int, th.Tensor] = {}               # stage -> gradient of the gathered rows (first B' frames live)
        self.tokens: List[th.Tensor] = []
        self.fwd: Optional['PlanRecorder'] = None
        self.bwd: Optional['PlanRecorder'] = None
        self.heads: Dict[Any, HeadPlan] = {}
        self.max_heads = 64
        self.owner = None                      # worker id whose LSTM state currently lives in ``self.states``
        self.uses = 0
        self.consumed = -1
        self.rows_host: Optional[tuple] = None

    # called from RNNDetector.forward_sequence at every PAFPN input stage while the plan is being recorded
    def fork(self, stage: int, x_nchw: th.Tensor) -> th.Tensor:
        xp, tok = Fn.ForkInjectFn.apply(Fn.to_nhwc(x_nchw), self, stage)
        self.tokens.append(tok)
        return Fn.as_nchw(xp)

    def capture(self, module, wgrad_side: bool, max_lanes: int):
        mdl = module.mdl
        in_features = tuple(mdl.fpn.in_features)
        saved_arena = ops.StatArena.swap(self.arena)
        from leod_amd.parallel import GradBuckets
        saved_buckets = GradBuckets.current
        saved_side = Fn.WgradSide.active
        ops.PackCache.invalidate()                         # the capture must contain the weight packs of a step
        flat = getattr(module, '_flat', None)              # FlatParams of configure_optimizers: the recorded step starts by refreshing
        if flat is not None:                               # its 16-bit weight shadows, and the recorded GEMMs read them
            ops.weight_shadow_pin(True)
        capture_stream = th.cuda.Stream(device=self.ev.device)
        capture_stream.wait_stream(th.cuda.current_stream())
        th.cuda.synchronize()
        fwd = PlanRecorder(max_lanes)
        bwd = None
        try:
            with th.enable_grad(), th.cuda.stream(capture_stream):
                PlanRecorder.current = fwd
                GradBuckets.current = saved_buckets           # boundary nodes reach the buckets through PlanRecorder.split
                fwd.begin()
                if flat is not None:
                    flat.ensure_shadow(force=True)
                ops.StatArena.begin_step(self.ev.device)
                ops.rows_masked_zero(_flat(self.states), self.is_first)                       # RNNStates.reset (detection.py:95-157)
                self.tokens = []
                _, new_states = mdl.backbone.forward_sequence(self.ev, self.states, select_stages=in_features, inject=self)
                ops.copy_multi(_flat(self.states), [t.detach() for t in _flat(new_states)])      # state hand-over to the next step
                fwd.end()
                if set(self.x_out) != set(in_features):
                    raise LeodHipError('backbone plan: the stage outputs the PAFPN reads were not exposed by forward_sequence')
                for k in in_features:                          # static gradient slots (no capture is in progress here: ordinary memory)
                    self.gsel[k] = th.zeros(self.x_out[k].shape, dtype=th.float32, device=self.ev.device)
                bwd = PlanRecorder(max_lanes, pool=fwd.pool)
                PlanRecorder.current = bwd
                bwd.begin()
                Fn.WgradSide.active = wgrad_side
                try:
                    th.autograd.backward(self.tokens, [th.ones_like(t) for t in self.tokens])
                finally:
                    Fn.WgradSide.active = False
                bwd.end()
            self.tokens = []
            self.arena_high = ops.StatArena.high
            self.fwd, self.bwd = fwd, bwd
            # how many kernels of the two plans read the event tensor through a re-pointable argument: the stem convolution in the forward plan and
            # its weight gradient in the backward plan, or none (float events take the generic convolution: the batch is then copied in)
            nf, nb = fwd.rebase_input(self.ev, self.ev), bwd.rebase_input(self.ev, self.ev)
            self.rebase_ok, self.rebase_count = (nf >= 1 and nb >= 1), nf + nb
        except BaseException:
            _abort_captures(capture_stream, fwd, bwd)
            raise
        finally:
            if flat is not None:
                ops.weight_shadow_pin(False)
            PlanRecorder.current = None
            th.cuda.current_stream().wait_stream(capture_stream)
            ops.StatArena.swap(*saved_arena)
            GradBuckets.current = saved_buckets
            Fn.WgradSide.active = saved_side
            ops.PackCache.invalidate()                     # nothing was executed: the pack buffers do not hold what the cache believes
        return self

    # ---- replay -------------------------------------------------------------------------------------------------------------
    def stage_inputs(self, ev: th.Tensor, rows_host: tuple, rows_dev_padded: th.Tensor, is_first: th.Tensor):
        if ev.data_ptr() != self.ev_now.data_ptr():
            # the batch's own event tensor is read in place: the stem convolution (forward plan) and its weight gradient (backward plan) are
            # re-pointed at it -- 245 MB less to copy per step at the benchmark size.  The tensor stays referenced here until the next batch
            # replaces it (the backward plan reads it last).  Plans without a registered input kernel (a float event tensor takes the generic
            # convolution) keep the copy into the captured buffer.
            if self.rebase_ok and ev.dtype == self.ev.dtype and ev.shape == self.ev.shape and ev.is_contiguous():
                n = self.fwd.rebase_input(self.ev, ev) + self.bwd.rebase_input(self.ev, ev)
                assert n == self.rebase_count, (n, self.rebase_count)
                self.ev_now = ev
            else:
                if self.ev_now is not self.ev and self.rebase_ok:
                    self.fwd.rebase_input(self.ev, self.ev)
                    self.bwd.rebase_input(self.ev, self.ev)
                self.ev.copy_(ev, non_blocking=True)
                self.ev_now = self.ev
        if rows_host != self.rows_host:                    # the labelled frames moved: one small copy (frame indices, -1 beyond B')
            self.rows.copy_(rows_dev_padded, non_blocking=True)
            self.rows_host = rows_host
        self.is_first.copy_(is_first, non_blocking=True)

    def load_states(self, rnn, worker_id):
        """The LSTM state of ``worker_id`` into the static state buffers (RNNStates semantics: one state set per loader worker)."""
        if self.owner is not None and self.owner != worker_id:
            held = rnn.get_states(self.owner)
            if held is not None and _flat(held)[0].data_ptr() == _flat(self.states)[0].data_ptr():
                spill = [(th.empty_like(h), th.empty_like(c)) for h, c in self.states]      # the previous owner keeps its state
                ops.copy_multi(_flat(spill), _flat(self.states))
                rnn.save_states_and_detach(self.owner, spill)
        saved = rnn.get_states(worker_id)
        if saved is None:
            for t in _flat(self.states):
                t.zero_()
        elif _flat(saved)[0].data_ptr() != _flat(self.states)[0].data_ptr():
            src, dst = _flat(saved), _flat(self.states)
            if ops.multi_ok(src) and ops.multi_ok(dst) and all(a.stride() == b.stride() and a.dtype is b.dtype for a, b in zip(src, dst)):
                ops.copy_multi(dst, src)
            else:
                for a, b in zip(dst, src):
                    a.copy_(b)
        self.owner = worker_id

    def run_forward(self):
        if self.arena_high:
            self.arena[:self.arena_high].zero_()           # LayerScale scratch of the captured backward pass
        self.fwd.replay()
        self.uses += 1

    def run_backward(self):
        self.bwd.replay()
        ops.PackCache.invalidate()                         # the replayed step re-packed conv / LSTM weights behind the cache's back

    def head(self, key):
        """-> HeadPlan | 'eager' (its capture failed before) | None"""
        h = self.heads.get(key)
        if h is not None:
            self.heads[key] = self.heads.pop(key)          # most recently used last
        return h

    def add_head(self, key, h):
        while len(self.heads) >= self.max_heads:
            old = self.heads.pop(next(iter(self.heads)))
            if isinstance(old, HeadPlan):
                old.close()
        self.heads[key] = h

    def close(self):
        for h in self.heads.values():
            if isinstance(h, HeadPlan):
                h.close()
        self.heads.clear()
        for p in (self.fwd, self.bwd):
            if p is not None:
                p.close()
        self.fwd = self.bwd = None


class TrainStepPlans:
    """Per-module cache of launch plans.  {event-tensor geometry: BackbonePlan | 'seen' | 'eager'} with a small LRU bound (each backbone
    plan owns the activation pool of its step: ~13 GB for RVT-S bs 8 L 21 of the 288 GB); every backbone plan holds its head plans
    {(B', padded label width): HeadPlan} (up to LEOD_PLAN_MAX_HEADS = 64, small).  A backbone geometry is captured the second time it is
    seen (the first, eager, step is the warm-up a capture needs); a head geometry is captured the first time it is seen under a captured
    backbone -- the labelled-frame count is data dependent and every new count would otherwise cost an eager step.  Counters:
    ``steps`` planned-path steps, ``replays`` of them executed entirely from plans captured EARLIER, ``captures`` / ``head_captures``."""

    def __init__(self, max_plans: Optional[int] = None, max_lanes: Optional[int] = None):
        self.entries: Dict[Any, Any] = {}
        self.max_plans = 4 if max_plans is None else max_plans
        self.max_lanes = 2 if max_lanes is None else max_lanes
        self.anchor = None
        self.captures = 0
        self.head_captures = 0
        self.replays = 0
        self.steps = 0
        self.eager_steps = 0

    plan_dist = True        # N > 1: steps with collectives are recorded in segments (False: such steps stay eager)

    @staticmethod
    def allowed() -> bool:
        """Not while a step is being captured.  (Collectives between kernels -- SyncBatchNorm, gradient buckets -- split the capture into
        segments, see ``PlanRecorder``; ``TrainStepPlans.plan_dist = False`` keeps steps with collectives eager.)"""
        if th.cuda.is_current_stream_capturing():
            return False
        return not Fn._sync_bn_on() or TrainStepPlans.plan_dist

    @staticmethod
    def key_of(ev: th.Tensor):
        return (tuple(ev.shape), ev.dtype, str(ev.device), ops.get_precision())

    @staticmethod
    def head_key_of(n_frames: int, nmax: int):
        nmax_pad = max(NMAX_PAD, -(-nmax // NMAX_PAD) * NMAX_PAD)
        return (n_frames, nmax_pad), nmax_pad

    def hit_rate(self) -> float:
        """fraction of the training steps so far that ran entirely from previously captured plans"""
        n = self.steps + self.eager_steps
        return self.replays / n if n else 0.0

    def lookup(self, key):
        """-> BackbonePlan to replay | 'capture' (second occurrence) | None (run eagerly)."""
        e = self.entries.get(key)
        if isinstance(e, BackbonePlan):
            self.entries[key] = self.entries.pop(key)       # most recently used last
            return e
        if e is None:
            if len(self.entries) > 64:                      # geometries seen once and never again do not accumulate; 'eager' markers
                for k in [k for k, v in self.entries.items() if v == 'seen']:      # (failed / evicted captures) stay, bounded below
                    del self.entries[k]
                eager = [k for k, v in self.entries.items() if v == 'eager']
                for k in eager[:max(0, len(eager) - 256)]:
                    del self.entries[k]
            self.entries[key] = 'seen'
            return None
        if e == 'seen':
            return 'capture'
        return None                                           # 'eager': a capture of this geometry failed before

    def build(self, key, module, ev, states_like, wgrad_side) -> Optional[BackbonePlan]:
        plans = [k for k, v in self.entries.items() if isinstance(v, BackbonePlan)]
        while len(plans) >= self.max_plans:
            old = plans.pop(0)
            self.entries.pop(old).close()
            self.entries[old] = 'eager'                     # an evicted geometry is not captured again (no capture / evict thrash)
        entry = BackbonePlan(key, module, ev, states_like)
        try:
            entry.capture(module, wgrad_side, self.max_lanes)
        except (LeodHipError, RuntimeError) as e:
            if os.environ.get('LEOD_PLAN_DEBUG'):
                import traceback
                traceback.print_exc()
            warnings.warn(f'leod_amd: the backbone of geometry {key} could not be turned into a launch plan ({e}); its steps stay eager')
            entry.close()
            self.entries[key] = 'eager'
            return None
        self.entries[key] = entry
        self.captures += 1
        return entry

    def build_head(self, bb: BackbonePlan, hkey, module, n_frames, nmax_pad, wgrad_side) -> Optional[HeadPlan]:
        h = HeadPlan(bb, n_frames, nmax_pad)
        try:
            h.capture(module, wgrad_side, self.max_lanes)
        except (LeodHipError, RuntimeError) as e:
            if os.environ.get('LEOD_PLAN_DEBUG'):
                import traceback
                traceback.print_exc()
            warnings.warn(f'leod_amd: the head pass for {n_frames} labelled frames could not be turned into a launch plan ({e}); such steps stay eager')
            h.close()
            bb.add_head(hkey, 'eager')
            return None
        bb.add_head(hkey, h)
        self.head_captures += 1
        return h

    def info(self) -> Optional[Dict[str, Any]]:
        """What the most recently used (backbone, head) plan pair launches per step, and the cache's counters."""
        bbs = [v for v in self.entries.values() if isinstance(v, BackbonePlan)]
        if not bbs:
            return None
        bb = bbs[-1]
        hds = [h for h in bb.heads.values() if isinstance(h, HeadPlan)]
        if not hds:
            if bb.fwd is None or bb.bwd is None:
                return None
            # planned backbone, eager head (EagerHeadGate: the N > 1 configuration)
            return {'forward': bb.fwd.info(), 'backward': bb.bwd.info(), 'backbone_forward_kernels': bb.fwd.info()['kernels'],
                    'head_forward_kernels': None, 'head_backward_kernels': None, 'backbone_backward_kernels': bb.bwd.info()['kernels'],
                    'head': 'eager', 'captures': self.captures, 'head_captures': 0, 'head_plans': 0, 'planned_steps': self.steps,
                    'replays': self.replays, 'eager_steps': self.eager_steps, 'plan_hit_rate': round(self.hit_rate(), 4)}
        hd = hds[-1]

        def both(a, b):
            return {k: (max(a[k], b[k]) if k == 'lanes' else a[k] + b[k]) for k in a}
        return {'forward': both(bb.fwd.info(), hd.fwd.info()), 'backward': both(hd.bwd.info(), bb.bwd.info()),
                'backbone_forward_kernels': bb.fwd.info()['kernels'], 'head_forward_kernels': hd.fwd.info()['kernels'],
                'head_backward_kernels': hd.bwd.info()['kernels'], 'backbone_backward_kernels': bb.bwd.info()['kernels'],
                'captures': self.captures, 'head_captures': self.head_captures, 'head_plans': len(hds), 'planned_steps': self.steps,
                'replays': self.replays, 'eager_steps': self.eager_steps, 'plan_hit_rate': round(self.hit_rate(), 4)}

    def clear(self):
        for v in self.entries.values():
            if isinstance(v, BackbonePlan):
                v.close()
        self.entries.clear()
