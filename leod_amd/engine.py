"""Tensor-level drivers of the LEOD hot path on MI355X, below the reference-shaped API of ``leod_amd.modules``
(``Module.training_step`` + ``leod_amd.optim.FlatAdamW`` is the product training path; it runs the same schedule and kernels).

``TrainEngine.step(ev_seq, labels, label_tb, is_first)`` = one full training step on device tensors (no label containers,
no loader dictionaries): reset LSTM rows -> stage-major time-batched backbone over T frames -> head + SimOTA + loss on the
labelled frames -> backward (HIP wgrad kernels accumulate into the flat gradient buffer) -> RCCL all-reduce -> fused
value-clip + AdamW -> OneCycle LR -> detach states.  No host synchronisation inside a step, so the whole step can also be
captured into ONE hipGraph (``capture`` / ``step_graph``).  Used by the kernel-level parity tests and the graph replay.
"""
import os
from typing import Dict, List, Optional, Sequence

import torch

from . import ops
from .functions import WgradSide, flush_bn_counters
from .parallel import FlatParams, DataParallel, one_cycle_lr
from .models.detection.yolox.utils.boxes import postprocess_padded
from .modules.utils.ssod import pred2label_padded
from .utils.host import bound_host_threads


class TrainEngine:
    def __init__(self, detector: torch.nn.Module, lr=2e-4, weight_decay=0.0, total_steps=400000, pct_start=0.005,
                 div_factor=20, final_div_factor=10000, clip_value=1.0, process_group=None, sync_bn=True, grad_buckets=True):
        bound_host_threads()
        self.det = detector
        self.det.train()
        self.flat = FlatParams(detector)
        self.dp = DataParallel(self.flat, process_group, sync_bn=sync_bn)
        self.dp.broadcast_parameters()
        self.dp.make_buckets(detector, bucketed=grad_buckets)
        self.hp = dict(lr=lr, weight_decay=weight_decay, total_steps=total_steps, pct_start=pct_start,
                       div_factor=div_factor, final_div_factor=final_div_factor, clip_value=clip_value)
        self.global_step = 0
        self.states = None
        self.last_losses = None
        self._idx_cache = {}
        self._graph = None
        self._capturing = False
        # weight-gradient kernels on a side HIP stream (eager launches): they feed nothing until the optimiser, so they
        # fill the device while the sequential parts of the backward pass (BPTT of the ConvLSTMs: 2 small launches per
        # timestep) run on the launch stream.  45.4 -> 43.6 ms per step; not used inside hipGraph capture.
        self.wgrad_side = True
        self.graph_side = False      # set for the duration of a capture that a launch plan replays: the weight-gradient fork / join is captured

    def current_lr(self):
        h = self.hp
        return one_cycle_lr(self.global_step, h['lr'], h['total_steps'], h['pct_start'], h['div_factor'], h['final_div_factor'])

    def _reset_rows(self, states, is_first):
        """RNNStates.reset: zero the LSTM rows of samples that start a new sequence (in place, sync-free:
        a masked multiply instead of boolean indexing so the step can be captured in a hipGraph)."""
        if states is None or is_first is None:
            return
        keep = (~is_first).to(torch.float32).view(-1, 1, 1, 1)
        for h, c in states:
            h.mul_(keep)
            c.mul_(keep)

    def forward_loss(self, ev_seq: torch.Tensor, labels: torch.Tensor, label_tb: Sequence[Sequence[int]],
                     is_first: Optional[torch.Tensor] = None, states=None):
        """ev_seq [T,B,C,H,W] uint8/fp32 on the device; ``label_tb[t]`` = batch indices with labels at timestep t (host
        lists, static); labels [B',N,7] yolox targets in (t, b) order, already on the device."""
        self._reset_rows(states, is_first)
        # Stage-major, time-batched schedule: every stage processes all T timesteps of the batch in one go
        # (``RNNDetector.forward_sequence``) -- the per-frame layers see T*B samples per launch, only the ConvLSTM
        # recurrence is unrolled over t.  Same arithmetic as the per-timestep loop of modules/detection.py:188-226 (no
        # layer in front of the LSTM mixes samples), ~6x fewer and ~21x larger kernel launches.
        T, B = ev_seq.shape[:2]
        rows = tuple(t * B + b for t in range(T) for b in label_tb[t])
        idx = self._index(rows, ev_seq.device)
        _, states, feats = self.det.backbone.forward_sequence(ev_seq, states, select_rows=idx, select_stages=tuple(self.det.fpn.in_features))
        preds, losses = self.det.forward_detect(feats, targets=labels)
        return preds, losses, [(h.detach(), c.detach()) for h, c in states]

    def _index(self, idx, device):
        key = tuple(idx)
        if key not in self._idx_cache:
            self._idx_cache[key] = torch.tensor(list(idx), dtype=torch.long, device=device)
        return self._idx_cache[key]

    def _step_body(self, ev_seq, labels, label_tb, is_first, states, hp_dev=None, lr=None, scale=1.0):
        self.flat.ensure_shadow(force=torch.cuda.is_current_stream_capturing())   # bf16 copy of the weights for the Linear kernels
        self.flat.zero_grad()
        if not self._capturing and not torch.cuda.is_current_stream_capturing():
            self.dp.begin_step()                         # per-stage gradient buckets are exchanged under the backward pass
        ops.StatArena.begin_step(ev_seq.device)          # one memset for all BatchNorm statistic accumulators of the step
        _, losses, new_states = self.forward_loss(ev_seq, labels, label_tb, is_first, states)
        WgradSide.active = self.wgrad_side and (self.graph_side or (not self._capturing and not torch.cuda.is_current_stream_capturing()))
        try:
            losses['loss'].backward()
        finally:
            WgradSide.active = False
            WgradSide.join()            # parameter gradients are complete on the launch stream from here on
        if hp_dev is None:
            scale = self.dp.all_reduce_gradients()
            self.flat.adamw_step(lr, self.hp['weight_decay'], self.hp['clip_value'], grad_scale=scale)
        else:
            self.dp.all_reduce_gradients()
            self.flat.adamw_step(0.0, self.hp['weight_decay'], self.hp['clip_value'], hp_dev=hp_dev)
        ops.StatArena.end_step()
        flush_bn_counters(self.det)
        return losses, new_states

    def step(self, ev_seq, labels, label_tb, is_first=None):
        """Eager step (one launch per kernel from Python)."""
        losses, self.states = self._step_body(ev_seq, labels, label_tb, is_first, self.states, lr=self.current_lr())
        self.global_step += 1
        self.last_losses = losses
        return losses

    # ---- hipGraph replay ---------------------------------------------------------------------------------
    def capture(self, ev_seq, labels, label_tb, is_first, plan: bool = False, max_lanes: int = 8):
        """Capture the WHOLE training step (zero-grad, 21 timesteps, head/loss, backward, all-reduce, AdamW, state
        hand-over) into one hipGraph.  Shapes are static (reference asserts constant B and HxW, detection.py:176-199);
        per-step scalars (lr, bias corrections) live in device memory.  ~4400 kernel launches per step then cost one
        graph launch on the host."""
        dev = ev_seq.device
        self._g_ev, self._g_labels = ev_seq.clone(), labels.clone()
        self._g_first = torch.ones_like(is_first)
        self._g_label_tb = [list(x) for x in label_tb]
        self._g_hp = torch.zeros(4, dtype=torch.float32, device=dev)
        # static LSTM state buffers (channels-last memory, NCHW logical), produced by one eager forward
        with torch.no_grad():
            _, st = self.det.forward_backbone(ev_seq[0], None)
        self._g_states = [(torch.zeros_like(h), torch.zeros_like(c)) for h, c in st]
        if self.states is not None:
            for (gh, gc), (h, c) in zip(self._g_states, self.states):
                gh.copy_(h)
                gc.copy_(c)
        ops.set_scalars4(self._g_hp, 1e-12, 1.0, 1.0, 1.0)           # capture-time scalars (lr ~ 0: weights barely move)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        saved = (self.flat.data.clone(), self.flat.exp_avg.clone(), self.flat.exp_avg_sq.clone(),
                 [(h.clone(), c.clone()) for h, c in self._g_states],
                 [m.bn.running_mean.clone() for m in self.det.modules() if hasattr(m, 'bn')],
                 [m.bn.running_var.clone() for m in self.det.modules() if hasattr(m, 'bn')])
        self._capturing = True                                          # single-stream schedule from here on
        with torch.cuda.stream(side):                                   # warm-up on a side stream, as torch requires
            self._graph_body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # plan=True: the capture is replayed by a launch plan (ops.LaunchPlan: plain stream launches from one C loop, parallel branches
        # on the plan's side streams) instead of hipGraphLaunch -- the fork / join of the weight-gradient side stream and of the head's
        # level streams is then worth capturing
        self._graph = torch.cuda.CUDAGraph(keep_graph=True) if plan else torch.cuda.CUDAGraph()
        side_in_graph, self.graph_side = self.graph_side, self.graph_side or plan
        ops.PackCache.invalidate()                                      # the capture must contain the weight packs of a step
        ops.weight_shadow_pin(True)                                     # the recorded GEMMs read the shadow the recorded refresh writes
        try:
            with torch.cuda.graph(self._graph):
                self._g_losses = self._graph_body()
        finally:
            ops.weight_shadow_pin(False)
        self.graph_side = side_in_graph
        self._capturing = False
        self._plan = ops.LaunchPlan(self._graph, max_lanes) if plan else None
        # undo the side effects of the warm-up + capture passes (capture itself does not execute)
        self.flat.data.copy_(saved[0]); self.flat.exp_avg.copy_(saved[1]); self.flat.exp_avg_sq.copy_(saved[2])
        self.flat.touch()
        for (gh, gc), (h, c) in zip(self._g_states, saved[3]):
            gh.copy_(h); gc.copy_(c)
        bns = [m.bn for m in self.det.modules() if hasattr(m, 'bn')]
        for bn, rm, rv in zip(bns, saved[4], saved[5]):
            bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
        torch.cuda.synchronize()

    def _graph_body(self):
        losses, new_states = self._step_body(self._g_ev, self._g_labels, self._g_label_tb, self._g_first, self._g_states,
                                             hp_dev=self._g_hp)
        # hand the states over to the next replay: one copy kernel (a tensor.copy_ would be captured as a 1-D memcpy node, which a launch
        # plan cannot read back from the graph)
        dst = [t for pair in self._g_states for t in pair]
        src = [t for pair in new_states for t in pair]
        if ops.multi_ok(dst) and ops.multi_ok(src) and all(d.stride() == s_.stride() for d, s_ in zip(dst, src)):
            ops.copy_multi(dst, src)
        else:
            for d, s_ in zip(dst, src):
                d.copy_(s_)
        return torch.stack([losses[k] for k in ('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss', 'num_fg')])

    def step_graph(self, ev_seq=None, labels=None, is_first=None):
        """Replay the captured step.  New inputs (same shapes) are copied into the static buffers first."""
        if ev_seq is not None and ev_seq.data_ptr() != self._g_ev.data_ptr():
            self._g_ev.copy_(ev_seq, non_blocking=True)
        if labels is not None and labels.data_ptr() != self._g_labels.data_ptr():
            self._g_labels.copy_(labels, non_blocking=True)
        if is_first is not None:
            self._g_first.copy_(is_first, non_blocking=True)
        ops.set_scalars4(self._g_hp, *self.flat.step_scalars(self.current_lr(), 1.0 / self.dp.world_size))
        if getattr(self, '_plan', None) is not None:
            self._plan.launch()
        else:
            self._graph.replay()
        ops.PackCache.invalidate()                          # the replayed AdamW kernel rewrote the parameters
        self.global_step += 1
        self.states = self._g_states
        self.last_losses = dict(zip(('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss', 'num_fg'), self._g_losses))
        return self.last_losses


class PseudoLabelEngine:
    """Inference loop of modules/pseudo_labeler.py:622-770 for sequences already on the device: optional hflip
    copy on the batch dim, backbone over T, one batched head pass over every (t, b) frame, batched NMS and the
    pseudo-label filters -- all device-side, no host sync until the caller reads the counts."""

    def __init__(self, detector: torch.nn.Module, num_classes: int, conf_thre=0.01, nms_thre=0.45, obj_thresh=(0.6, 0.3),
                 cls_thresh=(0.6, 0.3), dataset_name='gen1', downsampled_by_2=False, hflip=True, max_det=256):
        bound_host_threads()
        self.det = detector.eval()
        self.nc, self.conf, self.nms = num_classes, conf_thre, nms_thre
        self.obj_thresh, self.cls_thresh = list(obj_thresh), list(cls_thresh)
        self.dataset_name, self.ds2, self.hflip, self.max_det = dataset_name, downsampled_by_2, hflip, max_det
        self.states = None

    @torch.no_grad()
    def step(self, ev_seq: torch.Tensor, is_first: Optional[torch.Tensor] = None):
        """ev_seq [T,B,C,H,W] -> (labels [T*B', max_det, 8], counts [T*B'], detections, det counts); B' = 2B with hflip."""
        if self.hflip:
            ev_seq = torch.cat([ev_seq, torch.flip(ev_seq, dims=[-1])], dim=1)
            if is_first is not None:
                is_first = torch.cat([is_first, is_first])
        T = ev_seq.shape[0]
        if self.states is not None and is_first is not None:
            for h, c in self.states:
                h[is_first] = 0
                c[is_first] = 0
        # stage-major: every stage sees all T*B' frames per launch, only the ConvLSTM walks over t (same values as the
        # per-timestep loop of pseudo_labeler.py:687-722, see RNNDetector.forward_sequence)
        feats_all, states = self.det.backbone.forward_sequence(ev_seq, self.states)
        feats = {k: feats_all[k] for k in self.det.fpn.in_features}
        self.states = states
        preds, _ = self.det.forward_detect(feats)
        det, cnt = postprocess_padded(preds, self.nc, self.conf, self.nms, max_det=self.max_det)
        lab, lcnt = pred2label_padded(det, cnt, self.obj_thresh, self.cls_thresh, self.dataset_name, self.ds2)
        return lab, lcnt, det, cnt


class HostFeeder:
    """Double-buffered host -> HBM feed of the uint8 event batches on a copy stream, overlapped with the previous step.

    The reference's loaders hand over *host* tensors (Lightning moves them with a blocking ``.to(device)``, and the module
    then materialises an 8x larger fp32 copy, detection.py:132-135).  Here the batch stays uint8 end to end; ``put`` stages it
    in one of two pinned buffers and enqueues the PCIe copy on its own HIP stream, ``get`` makes the launch stream wait for
    that copy only.  A RVT-S Gen1 batch (21 x 8 x 20 x 240 x 304) is 245 MB: ~5 ms of PCIe Gen5 that disappears behind
    the ~44 ms step that is running meanwhile."""

    def __init__(self, shape, device, depth: int = 2):
        self.device = torch.device(device)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.host = [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(depth)]
        self.dev = [torch.empty(shape, dtype=torch.uint8, device=self.device) for _ in range(depth)]
        self.ready = [torch.cuda.Event() for _ in range(depth)]
        self.free = [None] * depth                 # event recorded on the launch stream once the buffer's consumer is enqueued
        self.n_put = self.n_get = 0

    def put(self, batch: torch.Tensor) -> None:
        """batch: host uint8 tensor [T,B,C,H,W].  A pinned batch (DataLoader(pin_memory=True)) is copied straight from where it
        lies and must stay untouched until the matching ``get``; a pageable one is staged through a pinned buffer first (a
        245 MB memcpy on the calling thread -- keep it off the thread that launches kernels)."""
        k = self.n_put % len(self.host)
        if self.free[k] is not None:
            self.copy_stream.wait_event(self.free[k])          # the step that read this buffer has been enqueued before
        src = batch
        if not batch.is_pinned():
            self.ready[k].synchronize()                         # the pinned staging buffer is reusable once its copy finished
            self.host[k].copy_(batch)
            src = self.host[k]
        with torch.cuda.stream(self.copy_stream):
            self.dev[k].copy_(src, non_blocking=True)
            self.ready[k].record(self.copy_stream)
        self.n_put += 1

    def get(self) -> torch.Tensor:
        """The oldest staged batch, valid on the current stream; call ``done`` after enqueuing its consumer."""
        assert self.n_get < self.n_put, 'get() without a staged batch'
        k = self.n_get % len(self.host)
        torch.cuda.current_stream().wait_event(self.ready[k])
        return self.dev[k]

    def done(self) -> None:
        k = self.n_get % len(self.host)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.free[k] = ev
        self.n_get += 1
