"""hipGraph "cell" scheduler for the recurrent backbone of the LEOD training step.

The RVT backbone is a T x 4 grid of *cells* (stage s at timestep t).  Cell (s, t) only depends on cell (s-1, t)
(its input map) and cell (s, t-1) (its LSTM state), so the grid is a wavefront; a single cell is ~45 small kernels
forward and ~75 backward, none of which fills 256 CUs on its own.  Two things therefore bound a step:

  * the host (~15 us of Python per launch x ~4500 launches) when every kernel is launched eagerly, and
  * the device when the whole step is ONE single-stream hipGraph (no kernel-level overlap; ROCm 7.2 cannot capture
    the multi-stream backward that autograd produces).

This scheduler removes both: the forward and the backward of every cell are captured ONCE into their own small
single-stream hipGraphs (``2 * T * 4`` graphs, private memory pool per stage), and a step replays them on one HIP
stream per stage, chained with events -- the same wavefront as ``TrainEngine._backbone_wavefront``, forward and
mirrored backward, at ~170 graph launches of host work.  Autograd is only used at capture time, cell-locally (one
``torch.autograd.backward`` per cell onto its input leaves); gradients travel between cells through static buffers:

    d h(s,t)  =  d x_in(s+1,t)  +  d h_prev(s,t+1)  +  d feat(s,t)      (the sum is the first node of the cell's
    d c(s,t)  =  d c_prev(s,t+1)                                         backward graph)

``d feat`` is the gradient the detection head sends into the feature map (stages 2-4, labelled frames only; a
zeroed static arena otherwise), so the backbone graphs do not depend on which frames carry labels.  The head
(PAFPN + YOLOX head + SimOTA + losses) runs eagerly on leaf views of the static cell outputs, or as one more
captured graph when the label layout is static.  Parameter gradients are accumulated by the wgrad kernels straight
into the flat gradient buffer, exactly as in the eager engine.

Reference control flow being scheduled: modules/detection.py:170-330 (training_step: per-timestep backbone loop,
RNN state hand-over, selected labelled frames -> head loss).
"""
from typing import List, Optional, Sequence

import torch

from . import ops
from .engine import TrainEngine
from .functions import flush_bn_counters


class CellGraphEngine(TrainEngine):
    """``build(ev, labels, label_tb, is_first)`` once (static shapes), then ``step_cells(ev, labels, is_first)``."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._built = False

    # ------------------------------------------------------------------------------------------------------------
    def build(self, ev_seq: torch.Tensor, labels: torch.Tensor, label_tb: Sequence[Sequence[int]], is_first: torch.Tensor,
              capture_head: Optional[bool] = None):
        det = self.det
        stages = det.backbone.stages
        S, T, B = len(stages), ev_seq.shape[0], ev_seq.shape[1]
        dev = ev_seq.device
        self._S, self._T = S, T
        if capture_head is None:
            capture_head = self.dp.world_size == 1          # SyncBatchNorm statistics are all-reduced eagerly
        self._capture_head = capture_head
        self._c_ev = ev_seq.clone()
        self._c_labels = labels.clone()
        self._c_label_tb = [list(x) for x in label_tb]
        self._c_first = is_first.clone()
        padded = det.backbone.in_res_hw
        if padded is not None and tuple(ev_seq.shape[-2:]) == tuple(padded):
            padded = None
        if self._streams is None:
            self._streams = [torch.cuda.Stream(device=dev) for _ in stages]
        streams = self._streams
        pools = [torch.cuda.graph_pool_handle() for _ in stages]

        # static initial LSTM states (zero = fresh sequence); the previous eager steps' states are carried over
        with torch.no_grad():
            _, st = det.forward_backbone(ev_seq[0], None)
        self._state0 = [(torch.zeros_like(h), torch.zeros_like(c)) for h, c in st]
        if self.states is not None:
            for (gh, gc), (h, c) in zip(self._state0, self.states):
                gh.copy_(h)
                gc.copy_(c)
        del st
        # gradient the head sends into each (stage >= 1, t) feature map: one zero-initialised arena, views per cell
        fpn_levels = [k - 1 for k in det.fpn.in_features]                  # stage indices feeding the PAFPN
        sizes = [self._state0[s][0].numel() for s in range(S)]
        self._dfeat_arena = torch.zeros(sum(sizes[s] for s in fpn_levels) * T, dtype=torch.float32, device=dev)
        self._dfeat = [[None] * T for _ in range(S)]
        off = 0
        for s in fpn_levels:
            ref = self._state0[s][0]
            for t in range(T):
                self._dfeat[s][t] = torch.as_strided(self._dfeat_arena, ref.shape, ref.stride(), storage_offset=off)
                off += sizes[s]

        saved = self._snapshot()
        torch.cuda.synchronize()
        # ---- forward cells, in wavefront order ---------------------------------------------------------------
        self._gf = [[None] * T for _ in range(S)]
        self._gb = [[None] * T for _ in range(S)]
        out_h = [[None] * T for _ in range(S)]
        out_c = [[None] * T for _ in range(S)]
        leaves = [[None] * T for _ in range(S)]          # (x_in, h_prev, c_prev) leaf tensors of each cell
        for t in range(T):
            for s, stage in enumerate(stages):
                x_in = self._c_ev[t] if s == 0 else out_h[s - 1][t].detach().requires_grad_()
                if t == 0:
                    h_prev, c_prev = self._state0[s]
                else:
                    h_prev = out_h[s][t - 1].detach().requires_grad_()
                    c_prev = out_c[s][t - 1].detach().requires_grad_()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(streams[s]):
                    g.capture_begin(pool=pools[s])
                    try:
                        h, (h2, c) = stage(x_in, (h_prev, c_prev), None, padded if s == 0 else None)
                    finally:
                        g.capture_end()
                assert h2 is h
                self._gf[s][t], out_h[s][t], out_c[s][t] = g, h, c
                leaves[s][t] = (x_in, h_prev, c_prev)
        torch.cuda.synchronize()
        for t in range(T):                       # run the forward once so the head warm-up below sees real feature maps
            for s in range(S):
                self._gf[s][t].replay()
        torch.cuda.synchronize()
        # ---- head: leaves on the static feature maps of the labelled frames -------------------------------------
        self._out_h, self._out_c = out_h, out_c
        self._head_graph = None
        if capture_head:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                                   # warm-up pass, as torch requires
                self._head_pass(self._c_labels, self._c_label_tb)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self._head_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._head_graph):
                self._head_losses = self._head_pass(self._c_labels, self._c_label_tb)
        # ---- backward cells, in mirrored wavefront order -------------------------------------------------------
        dx_in = [[None] * T for _ in range(S)]
        dh_prev = [[None] * T for _ in range(S)]
        dc_prev = [[None] * T for _ in range(S)]
        for t in reversed(range(T)):
            for s in reversed(range(S)):
                parts = []
                if s + 1 < S:
                    parts.append(dx_in[s + 1][t])
                if t + 1 < T:
                    parts.append(dh_prev[s][t + 1])
                if self._dfeat[s][t] is not None:
                    parts.append(self._dfeat[s][t])
                dc = dc_prev[s][t + 1] if t + 1 < T else None
                x_in, h_prev, c_prev = leaves[s][t]
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(streams[s]):
                    g.capture_begin(pool=pools[s])
                    try:
                        gh = parts[0] if len(parts) == 1 else (parts[0] + parts[1] if len(parts) == 2
                                                               else parts[0] + parts[1] + parts[2])
                        outs, gouts = [out_h[s][t]], [gh]
                        if dc is not None:
                            outs.append(out_c[s][t])
                            gouts.append(dc)
                        # full backward (NOT autograd.grad(inputs=...), which would prune every node that only leads
                        # to parameters and with it the wgrad kernels those nodes launch)
                        torch.autograd.backward(outs, gouts)
                    finally:
                        g.capture_end()
                if x_in.requires_grad:
                    dx_in[s][t] = x_in.grad
                if h_prev.requires_grad:
                    dh_prev[s][t], dc_prev[s][t] = h_prev.grad, c_prev.grad
                self._gb[s][t] = g
                leaves[s][t] = None
        self._keep = (dx_in, dh_prev, dc_prev, pools)                      # static buffers referenced by the graphs
        torch.cuda.synchronize()
        self._ev_f = [[torch.cuda.Event() for _ in range(T)] for _ in range(S)]
        self._ev_b = [[torch.cuda.Event() for _ in range(T)] for _ in range(S)]
        self._restore(saved)
        self._dfeat_arena.zero_()
        self.flat.zero_grad()
        torch.cuda.synchronize()
        self._built = True

    # ------------------------------------------------------------------------------------------------------------
    def _snapshot(self):
        bns = [m.bn for m in self.det.modules() if hasattr(m, 'bn')]
        return (self.flat.data.clone(), self.flat.exp_avg.clone(), self.flat.exp_avg_sq.clone(),
                [(h.clone(), c.clone()) for h, c in self._state0],
                [(bn.running_mean.clone(), bn.running_var.clone(), bn.num_batches_tracked.clone()) for bn in bns])

    def _restore(self, saved):
        self.flat.data.copy_(saved[0]); self.flat.exp_avg.copy_(saved[1]); self.flat.exp_avg_sq.copy_(saved[2])
        for (gh, gc), (h, c) in zip(self._state0, saved[3]):
            gh.copy_(h); gc.copy_(c)
        bns = [m.bn for m in self.det.modules() if hasattr(m, 'bn')]
        for bn, (rm, rv, nb) in zip(bns, saved[4]):
            bn.running_mean.copy_(rm); bn.running_var.copy_(rv); bn.num_batches_tracked.copy_(nb)

    def _head_pass(self, labels, label_tb):
        """PAFPN + head + loss on the labelled frames and its backward into ``d feat``; returns the stacked losses."""
        ops.StatArena.begin_step(labels.device)
        sel, leaf_of = {}, []
        for t, idx in enumerate(label_tb):
            if not len(idx):
                continue
            for k in self.det.fpn.in_features:
                leaf = self._out_h[k - 1][t].detach().requires_grad_()
                leaf_of.append((k - 1, t, leaf))
                v = leaf.permute(0, 2, 3, 1)
                sel.setdefault(k, []).append(v if len(idx) == v.shape[0] else v[self._index(idx, v.device)])
        feats = {k: torch.cat(v, 0).permute(0, 3, 1, 2) for k, v in sel.items()}
        _, losses = self.det.forward_detect(feats, targets=labels)
        losses['loss'].backward()
        for s, t, leaf in leaf_of:
            self._dfeat[s][t].copy_(leaf.grad)
        ops.StatArena.end_step()
        flush_bn_counters(self.det)
        return torch.stack([losses[k].detach() for k in ('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss', 'num_fg')])

    # ------------------------------------------------------------------------------------------------------------
    def step_cells(self, ev_seq=None, labels=None, is_first=None, label_tb=None):
        """One training step through the captured cells.  New inputs of the captured shapes are copied into the static
        buffers; ``label_tb`` may change between steps only when the head is not captured (``capture_head=False``)."""
        assert self._built, 'call build() first'
        S, T = self._S, self._T
        main = torch.cuda.current_stream()
        if ev_seq is not None and ev_seq.data_ptr() != self._c_ev.data_ptr():
            self._c_ev.copy_(ev_seq, non_blocking=True)
        if labels is not None and labels.data_ptr() != self._c_labels.data_ptr():
            self._c_labels.copy_(labels, non_blocking=True)
        if label_tb is not None:
            label_tb = [list(x) for x in label_tb]
            if label_tb != self._c_label_tb:
                if self._head_graph is not None:
                    raise ValueError('the captured head graph is specialised to the label layout given to build(); '
                                     'build(..., capture_head=False) accepts a new layout every step')
                self._c_label_tb = label_tb
                self._dfeat_arena.zero_()
        self.flat.zero_grad()
        self._reset_rows(self._state0, is_first)
        streams = self._streams
        for st in streams:
            st.wait_stream(main)
        # ---- forward wavefront ------------------------------------------------------------------------------------
        for t in range(T):
            for s in range(S):
                st = streams[s]
                if s > 0:
                    st.wait_event(self._ev_f[s - 1][t])
                with torch.cuda.stream(st):
                    self._gf[s][t].replay()
                    if s + 1 < S:
                        self._ev_f[s][t].record(st)
        for st in streams:
            main.wait_stream(st)
        # ---- head -----------------------------------------------------------------------------------------------
        if self._head_graph is not None:
            self._head_graph.replay()
            lv = self._head_losses
        else:
            lv = self._head_pass(self._c_labels if labels is None else labels, self._c_label_tb)
        # ---- mirrored backward wavefront ------------------------------------------------------------------------
        for st in streams:
            st.wait_stream(main)
        for t in reversed(range(T)):
            for s in reversed(range(S)):
                st = streams[s]
                if s + 1 < S:
                    st.wait_event(self._ev_b[s + 1][t])
                with torch.cuda.stream(st):
                    self._gb[s][t].replay()
                    if s > 0:
                        self._ev_b[s][t].record(st)
        for st in streams:
            main.wait_stream(st)
        # ---- all-reduce + optimiser + state hand-over -------------------------------------------------------------
        scale = self.dp.all_reduce_gradients()
        self.flat.adamw_step(self.current_lr(), self.hp['weight_decay'], self.hp['clip_value'], grad_scale=scale)
        with torch.no_grad():
            for s in range(S):
                self._state0[s][0].copy_(self._out_h[s][T - 1])
                self._state0[s][1].copy_(self._out_c[s][T - 1])
        self.states = self._state0
        self.global_step += 1
        names = ('loss', 'iou_loss', 'conf_loss', 'cls_loss', 'l1_loss', 'num_fg')
        self.last_losses = {k: lv[i] for i, k in enumerate(names)}
        return self.last_losses
