"""autograd glue: each Function below is one fused block of the LEOD hot path whose forward AND
backward are sequences of hand-written HIP kernels (leod_amd.ops -> libleod_hip.so).

PyTorch's autograd engine is used purely as the tape between blocks (the graph of a training step is
~350 nodes).  Parameter gradients are NOT returned to autograd: the wgrad kernels accumulate straight
into ``param.grad`` (allocated once, or a view into the flat gradient buffer set up by
``leod_amd.parallel.FlatParams``), so that the 21 timesteps of a sequence add into one buffer without
21 extra elementwise adds per parameter.  The Functions therefore return ``None`` for parameter
inputs; parameters are still passed to ``apply`` so that the outputs are attached to the graph.
"""
import os
from typing import Optional, Tuple

import torch
from torch.autograd import Function

from . import ops
from .comm import NativeComm

_SYNC_BN = {'group': None, 'world_size': 1, 'force': False, 'images': None, 'n_collectives': 0}



class WgradSide:
    """Weight-gradient kernels feed nothing until the optimiser, so the engine lets them run on a side HIP stream
    next to the dgrad chain (the per-kernel work of an RVT stage is too small to fill 256 CUs on its own).
    ``active`` is switched on by ``TrainEngine`` for the duration of its backward pass; it then calls ``join()``
    before the gradient all-reduce.  Tensors read on the side stream are kept alive until the join, so the caching
    allocator cannot hand their memory to a later main-stream kernel while the side stream still reads it."""
    active = False
    hold_main = False       # set around sections that are recorded in many short plan segments (see _conv_bn_bwd): no fork there
    streams = {}            # launch stream (raw handle) -> its side stream
    used = set()
    keep = []

    @classmethod
    def side_of(cls, main):
        key = main.cuda_stream
        st = cls.streams.get(key)
        if st is None:
            # default priority: any other HIP stream priority doubled the step on MI355X / ROCm 7.2 (profiles/r04_a_graph_ab.txt)
            st = cls.streams[key] = torch.cuda.Stream()
        cls.used.add(key)
        return st

    @classmethod
    def join(cls):
        if cls.used:
            cur = torch.cuda.current_stream()
            for key in cls.used:
                cur.wait_stream(cls.streams[key])
            cls.used.clear()
        cls.keep.clear()


class _wgrad_side:
    """``with _wgrad_side(dy, x): ops.linear_wgrad(...)`` -- fork: the side stream waits for everything enqueued so far."""

    def __init__(self, *tensors):
        self.tensors = tensors
        self.ctx = None

    def __enter__(self):
        if not WgradSide.active or WgradSide.hold_main:
            return
        main = torch.cuda.current_stream()
        side = WgradSide.side_of(main)
        side.wait_stream(main)
        WgradSide.keep.extend(t for t in self.tensors if t is not None)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


class BucketBoundaryFn(Function):
    """Identity in the forward pass; its backward runs when the gradients of ALL its inputs are complete, i.e. when every node
    between them and the loss has run -- it then tells the data-parallel engine that gradient bucket ``k`` is final
    (``leod_amd.parallel.GradBuckets.ready``): the all-reduce of that bucket starts under the rest of the backward pass."""

    @staticmethod
    def forward(ctx, k, *xs):
        ctx.k = k
        return tuple(x.view_as(x) for x in xs)

    @staticmethod
    def backward(ctx, *gs):
        from .parallel import GradBuckets
        b = GradBuckets.current
        if b is not None:
            from .modules.step_plan import PlanRecorder
            k = ctx.k
            if not PlanRecorder.split(lambda: GradBuckets.current is not None and GradBuckets.current.ready(k)):
                b.ready(k)
        return (None,) + tuple(gs)


def bucket_boundary(k: int, *xs):
    """``xs`` unchanged unless a bucketed gradient exchange is active for this step (``GradBuckets.current``) and gradients flow."""
    from .parallel import GradBuckets
    if GradBuckets.current is None or not torch.is_grad_enabled() or not any(x.requires_grad for x in xs):
        return xs if len(xs) > 1 else xs[0]
    if k < 0:
        k = GradBuckets.current.head                          # the PAFPN + head bucket
    out = BucketBoundaryFn.apply(k, *xs)
    return out if len(xs) > 1 else out[0]


class ForkSelectFn(Function):
    """A stage's output map x [N,H,W,C] (all T*B frames) has two consumers in the training step: the next stage (all frames) and the
    PAFPN (the labelled frames ``idx`` only, modules/detection.py:209-224).  As two autograd edges the backward pass costs a zero
    fill of an [N,H,W,C] tensor, an index_add into it and a full-size add of the two gradients; as ONE node the labelled frames'
    gradient is added in place into the gradient that arrives from the next stage (a fresh tensor private to this backward pass)."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(idx)
        ctx.shape = tuple(x.shape)
        return x.view_as(x), x.index_select(0, idx)

    @staticmethod
    def backward(ctx, g_pass, g_sel):
        (idx,) = ctx.saved_tensors
        if g_pass is None:
            if g_sel is None:
                return None, None
            dx = torch.zeros(ctx.shape, dtype=g_sel.dtype, device=g_sel.device)
        else:
            dx = _cont(g_pass)
        if g_sel is not None:
            gs = _cont(g_sel)
            if dx.is_cuda and dx.dtype is torch.float32 and gs.dtype is torch.float32 and idx.dtype is torch.int64 and dx[0].numel() % 4 == 0:
                ops.rows_index_add(dx, gs, idx)               # the frame indices of a step are distinct (t * B + b)
            else:
                dx.index_add_(0, idx, gs)
        return dx, None


class ForkInjectFn(Function):
    """ForkSelectFn for a step whose backbone and head are recorded as SEPARATE launch plans (modules/step_plan.py): the forward pass only
    exposes the stage output (``holder.x_out[stage]``, which the head plan gathers the labelled frames from) and hands out a token tensor;
    the backward pass -- started from the tokens -- adds the gradient of the gathered rows, which the head plan left in the static buffer
    ``holder.gsel[stage]``, into the gradient arriving from the next stage through the static row-index buffer ``holder.rows`` (frame
    indices, -1 beyond the step's labelled-frame count: skipped by the kernel).  Nothing here depends on how many frames are labelled."""

    @staticmethod
    def forward(ctx, x, holder, stage):
        ctx.set_materialize_grads(False)
        ctx.holder, ctx.stage, ctx.shape = holder, stage, tuple(x.shape)
        holder.x_out[stage] = x.detach()
        return x.view_as(x), x.new_empty(1)

    @staticmethod
    def backward(ctx, g_pass, g_tok):
        h = ctx.holder
        dx = torch.zeros(ctx.shape, dtype=torch.float32, device=h.rows.device) if g_pass is None else _cont(g_pass)
        ops.rows_index_add(dx, h.gsel[ctx.stage], h.rows)
        return dx, None, None


def fork_select(x_nchw: torch.Tensor, idx: torch.Tensor):
    """x [N,C,H,W] (channels-last memory) -> (x for the next consumer, x[idx]), both logical NCHW views of NHWC memory."""
    xp, xs = ForkSelectFn.apply(to_nhwc(x_nchw), idx)
    return as_nchw(xp), as_nchw(xs)


def set_sync_batchnorm(process_group, world_size: int):
    """Enable SyncBatchNorm semantics (reference: train.py:247 sync_batchnorm=True when >1 GPU): the
    per-channel (sum, sumsq) and backward (sum du, sum du*xhat) vectors are all-reduced over RCCL."""
    _SYNC_BN['group'] = process_group
    _SYNC_BN['world_size'] = int(world_size)


def _sync_bn_on() -> bool:
    return _SYNC_BN['world_size'] > 1 or _SYNC_BN['force']


CAPTURE_COLLECTIVES = False     # see _capture_collectives


def _capture_collectives(group) -> bool:
    """``functions.CAPTURE_COLLECTIVES = True`` (option, RCCL only): SyncBatchNorm exchanges issued while a step is being recorded are left to
    ProcessGroupNCCL's own stream capture and become nodes of the launch plan, instead of closing a plan segment each (a host callback,
    the side lane joined, the neck / head weight gradients held on the launch lane).  One rank, every collective issued
    (profiles/r05_q_captured_collectives.txt): 16.14 ms per step against 17.41 with callbacks and 15.42 without collectives.  NOT the default:
    a one-rank all-reduce puts no kernel into the graph, so the replay of real RCCL kernel nodes by the plan executor has never run."""
    if NativeComm.active and group is None:
        return True                                         # the library's communicator: a recorded exchange is an op of the launch plan (k_plan.hip)
    if not CAPTURE_COLLECTIVES:
        return False
    import torch.distributed as dist
    try:
        return dist.get_backend(group) == 'nccl'
    except Exception:                                      # noqa: BLE001
        return False


def _allreduce_stats(t: torch.Tensor):
    if _sync_bn_on():
        import torch.distributed as dist
        group = _SYNC_BN['group']

        def exchange():
            if NativeComm.usable(t, group):
                NativeComm.all_reduce(t)                        # on the launch stream, like a kernel (comm.py)
            else:
                dist.all_reduce(t, group=group)
            _SYNC_BN['n_collectives'] += 1
        # a step that is being recorded into launch plans takes the exchange as a host callback between two plan segments -- or, as an
        # option, leaves it to ProcessGroupNCCL's own stream capture (_capture_collectives)
        from .modules.step_plan import PlanRecorder
        if _capture_collectives(group) and torch.cuda.is_current_stream_capturing():
            exchange()
            return
        if not PlanRecorder.split(exchange):
            exchange()


def sync_bn_begin(n_images: int, device) -> None:
    """Start of a detection-head pass under SyncBatchNorm: the number of images of THIS rank (labelled frames, data
    dependent) is summed over ranks ONCE; every BatchNorm layer of the pass then derives its global row count on the device as
    rows-per-image x images (exact integers), so the per-layer exchanges carry only the statistics, in place, with no packing."""
    if not _sync_bn_on():
        _SYNC_BN['images'] = None
        return
    images = torch.full((1,), float(n_images), dtype=torch.float64, device=device)
    _allreduce_stats(images)
    _SYNC_BN['images'] = images


def flush_bn_counters(module: torch.nn.Module) -> None:
    """``BatchNorm2d.num_batches_tracked`` of every BaseConv is advanced lazily: the forward pass only counts calls on the
    host, this adds the pending counts to the device buffers in ONE fused launch (39 one-element add kernels per step
    otherwise).  Called by the engines at the end of a step and by ``BaseConv`` before its state dict is read."""
    bufs, incs = [], []
    for m in module.modules():
        n = getattr(m, 'bn_calls_pending', 0)
        if n and hasattr(m, 'bn'):
            bufs.append(m.bn.num_batches_tracked)
            incs.append(n)
            m.bn_calls_pending = 0
    if bufs:
        if len(set(incs)) == 1:
            torch._foreach_add_(bufs, incs[0])
        else:
            for b, n in zip(bufs, incs):
                b += n


def grad_buf(p: torch.nn.Parameter) -> torch.Tensor:
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad


def to_nhwc(t: torch.Tensor) -> torch.Tensor:
    """[B,C,H,W] (any strides) -> contiguous [B,H,W,C]; zero-copy for channels-last memory."""
    v = t.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def as_nchw(t_nhwc: torch.Tensor) -> torch.Tensor:
    return t_nhwc.permute(0, 3, 1, 2)


def _cont(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else (t if t.is_contiguous() else t.contiguous())


def _rows(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """A gradient for the BatchNorm backward kernels: as it is when its rows are evenly spaced (contiguous, or the channel slice that
    ``torch.cat``'s backward hands to each branch of a CSP / PAFPN concat -- read in place with its row stride), else a contiguous copy."""
    return None if t is None else (t if ops.row_stride(t, t.shape[-1]) else t.contiguous())


# ---------------------------------------------------------------------------------------------------
class ConvLNFn(Function):
    """ConvDownsampling_Cf2Cl (maxvit.py:143-182): conv(no bias) -> NHWC -> LayerNorm.
    ``x`` is the raw NCHW event tensor for the stem (uint8 / fp32, optionally unpadded) or an NHWC map."""

    @staticmethod
    def forward(ctx, mod, x, conv_w, ln_w, ln_b, is_stem: bool, stride: int, padded_hw):
        need = any(ctx.needs_input_grad)
        if is_stem:
            z = ops.stem_conv_fwd(x, conv_w, padded_hw, stride, mod.conv.padding[0])
        else:
            z = ops.conv_nhwc_fwd(x, conv_w, None, stride=stride)
        y, stats = ops.layernorm_fwd(z, ln_w, ln_b, want_stats=need)
        ctx.mod, ctx.is_stem, ctx.stride, ctx.padded_hw = mod, is_stem, stride, padded_hw
        ctx.save_for_backward(x, z, stats, conv_w, ln_w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, z, stats, conv_w, ln_w = ctx.saved_tensors
        mod = ctx.mod
        if mod.norm.weight is not None:
            dlw, dlb = grad_buf(mod.norm.weight), grad_buf(mod.norm.bias)
        else:                                                 # norm_affine=False: the kernel's scale / shift gradients go nowhere
            dlw, dlb = torch.zeros_like(ln_w), torch.zeros_like(ln_w)
        dz = ops.layernorm_bwd(_cont(dy), z, stats, ln_w, None, dlw, dlb)
        dx = None
        if ctx.is_stem:
            # the stem is the LAST node of the backward pass: on the launch stream its weight gradient (310 us) runs next to the tail of the
            # side stream's queue instead of behind it (the join before the optimiser waited for both in sequence)
            ops.stem_conv_wgrad(dz, x, grad_buf(mod.conv.weight), ctx.padded_hw, ctx.stride, mod.conv.padding[0])
        else:
            with _wgrad_side(dz, x):
                ops.conv_nhwc_wgrad(dz, x, grad_buf(mod.conv.weight), None, stride=ctx.stride)
            if ctx.needs_input_grad[1]:
                dx = ops.conv_nhwc_dgrad(dz, conv_w, x.shape, stride=ctx.stride)
        return None, dx, None, None, None, None, None, None


# ---------------------------------------------------------------------------------------------------
class AttnBlockFn(Function):
    """PartitionAttentionCl.forward (maxvit.py:185-270): x + ls1(attn(norm1(x))) then + ls2(mlp(norm2(.)))."""

    @staticmethod
    def forward(ctx, mod, x, *params):
        need = any(ctx.needs_input_grad)
        (n1w, n1b, qkv_w, qkv_b, proj_w, proj_b, g1, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b, g2) = params
        heads, part, window = mod.self_attn.num_heads, mod.partition_size, mod.partition_window
        # precision mode bf16: where the LDS attention kernels cover the geometry, qkv (and dqkv in backward) live in HBM as bf16
        q16 = ops.partition_attn_16bit_ok(x.shape[0], x.shape[1], x.shape[2], x.shape[3], heads, part)
        qkv, _, st1 = ops.ln_linear_fwd(x, n1w, n1b, qkv_w, qkv_b, want_stats=need or q16, out_bf16=q16)
        # ... and the attention output O (dO in backward) as bf16 rows where every consumer has the 16-bit path
        o16 = q16 and ops.attn_block_o16_ok(x.shape[0], x.shape[1], x.shape[2], x.shape[3], heads, part)
        o, lse = ops.partition_attn_fwd(qkv, heads, part, window, want_lse=need, out_bf16=o16)
        # the pre-LayerScale outputs are NOT stored: dgamma is recovered from the un-scaled weight gradient in backward
        y, _ = ops.linear_lsres_fwd(o, proj_w, proj_b, g1, x, want_t=False, a_gelu=False)
        # precision mode bf16, stage 1: the whole MLP in one launch, the hidden in registers (csrc/k_mlp.hip); with a backward pass to
        # come it also leaves the fp16 pre-activation and the LayerNorm statistics behind
        fused = ops.mlp_fwd_fused(y, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b, g2, want_saved=need)
        if fused is not None:
            z, u, st2 = fused
            h = None
        else:
            # precision mode bf16, stages 1-2: u comes back as ONE fp16 tensor (h is None) and fc2 applies GELU while loading it
            u, h, st2 = ops.ln_linear_fwd(y, n2w, n2b, fc1_w, fc1_b, want_act=True, want_stats=True)
            z, _ = ops.linear_lsres_fwd(h if h is not None else u, fc2_w, fc2_b, g2, y, want_t=False, a_gelu=h is None)
        if need:
            ctx.mod = mod
            ctx.u16 = h is None
            ctx.save_for_backward(x, qkv, st1, o, lse, y, u, u if h is None else h, st2, *params)
        return z

    @staticmethod
    def backward(ctx, dz):
        (x, qkv, st1, o, lse, y, u, h, st2,
         n1w, n1b, qkv_w, qkv_b, proj_w, proj_b, g1, n2w, n2b, fc1_w, fc1_b, fc2_w, fc2_b, g2) = ctx.saved_tensors
        mod = ctx.mod
        sa, mlp = mod.self_attn, mod.mlp
        heads, part, window = sa.num_heads, mod.partition_size, mod.partition_window
        dz = _cont(dz)
        # ---- dgrad chain (critical path); LayerScale is folded into the dgrad loader (dz * gamma) -------------------------
        fused = None
        if ctx.u16:                                           # stage 1, precision mode bf16: the MLP's dgrad chain in one launch (csrc/k_mlp.hip)
            fused = ops.mlp_bwd_dgrad_fused(dz, y, st2, n2w, n2b, fc1_w, fc1_b, fc2_w, g2, grad_buf(mod.norm2.weight), grad_buf(mod.norm2.bias))
        if fused is not None:
            dy, du = fused
        else:
            du = ops.linear_dgrad(dz, fc2_w, kscale=g2, aux_u=u)
            dy = ops.linear_dgrad_ln_bwd(du, fc1_w, y, st2, n2w, dz, grad_buf(mod.norm2.weight), grad_buf(mod.norm2.bias))
        do = ops.linear_dgrad(dy, proj_w, kscale=g1, out_bf16=o.dtype is not torch.float32)
        dqkv = ops.partition_attn_bwd(qkv, do, lse, heads, part, window)
        # ---- weight gradients: off the critical path (side stream when the engine enables it) ------
        with _wgrad_side(dz, h, du, y, st2, dy, o, dqkv, x, st1):
            # stages 3-4 (short row ranges, LDS-DMA kernel): the four weight gradients as ONE preparation / contraction / reduce launch
            grouped = ops.attn_block_wgrads(dz, h, ctx.u16, du, y, st2, n2w, n2b, dy, o, dqkv, x, st1, n1w, n1b,
                                            (fc2_w, fc2_b, g2, grad_buf(mlp.net[2].weight), grad_buf(mlp.net[2].bias), grad_buf(mod.ls2.gamma)),
                                            (grad_buf(mlp.net[0][0].weight), grad_buf(mlp.net[0][0].bias)),
                                            (proj_w, proj_b, g1, grad_buf(sa.proj.weight), grad_buf(sa.proj.bias), grad_buf(mod.ls1.gamma)),
                                            (grad_buf(sa.qkv.weight), grad_buf(sa.qkv.bias)))
            if not grouped:
                ops.layerscale_linear_wgrad(dz, h, fc2_w, fc2_b, g2, grad_buf(mlp.net[2].weight), grad_buf(mlp.net[2].bias),
                                            grad_buf(mod.ls2.gamma), h_gelu=ctx.u16)
                ops.linear_wgrad(du, y, grad_buf(mlp.net[0][0].weight), grad_buf(mlp.net[0][0].bias), stats=st2, ln_w=n2w, ln_b=n2b)
                ops.layerscale_linear_wgrad(dy, o, proj_w, proj_b, g1, grad_buf(sa.proj.weight), grad_buf(sa.proj.bias),
                                            grad_buf(mod.ls1.gamma), h_gelu=False)
                if n1w is not None:
                    ops.linear_wgrad(dqkv, x, grad_buf(sa.qkv.weight), grad_buf(sa.qkv.bias), stats=st1, ln_w=n1w, ln_b=n1b)
                else:
                    ops.linear_wgrad(dqkv, x, grad_buf(sa.qkv.weight), grad_buf(sa.qkv.bias))
        if n1w is not None:
            dx = ops.linear_dgrad_ln_bwd(dqkv, qkv_w, x, st1, n1w, dy, grad_buf(mod.norm1.weight), grad_buf(mod.norm1.bias))
        else:
            # dy (the residual branch's gradient) is still being read on the weight-gradient side stream: it is the dgrad's second
            # source, the sum goes to a new tensor in the same launch (no separate add kernel)
            dx = ops.linear_dgrad(dqkv, qkv_w, dres=dy)
        return (None, dx) + (None,) * 14


# ---------------------------------------------------------------------------------------------------
class ConvLSTMFn(Function):
    """DWSConvLSTM2d.forward (models/layers/rnn.py:37-70) on channels-last rows."""

    @staticmethod
    def forward(ctx, mod, x, h_prev, c_prev, w, b):
        need = any(ctx.needs_input_grad)
        C = x.shape[-1]
        h, c, gates = ops.convlstm_fwd(x, h_prev, c_prev, w.view(4 * C, 2 * C), b, want_gates=need)
        if need:
            ctx.mod = mod
            ctx.set_materialize_grads(False)
            ctx.save_for_backward(x, h_prev, c_prev, c, gates, w)
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        x, h_prev, c_prev, c, gates, w = ctx.saved_tensors
        mod = ctx.mod
        C = x.shape[-1]
        dgates, dc_prev = ops.convlstm_gates_bwd(_cont(dh), _cont(dc), gates, c_prev, c, want_dc_prev=ctx.needs_input_grad[3])
        with _wgrad_side(dgates, x, h_prev):
            ops.linear_wgrad(dgates, x, grad_buf(mod.conv1x1.weight).view(4 * C, 2 * C), grad_buf(mod.conv1x1.bias), x2=h_prev)
        dx, dh_prev = ops.linear_dgrad(dgates, w.view(4 * C, 2 * C), split=C)
        dx = dx.view(x.shape)
        dh_prev = dh_prev.view(x.shape) if ctx.needs_input_grad[2] else None
        return None, dx, dh_prev, dc_prev, None, None


class DepthwiseConvFn(Function):
    """k x k depthwise convolution with bias on an NHWC map, channels [lo, hi) of the layer's filters: ``conv3x3_dws`` of the ConvLSTM
    (models/layers/rnn.py:26-30,50-55) -- on h alone, or on x and h with the two halves of a [2C,1,k,k] filter bank (a depthwise conv of
    cat(x, h) is the two depthwise convs side by side).  apply(conv module, x, weight, bias, lo, hi) -> y"""

    @staticmethod
    def forward(ctx, conv, x, w, b, lo: int, hi: int):
        wv, bv = w[lo:hi], (b[lo:hi] if b is not None else None)
        y = ops.conv_nhwc_fwd(x, wv, bv, stride=1)
        if any(ctx.needs_input_grad):
            ctx.conv, ctx.lohi = conv, (lo, hi)
            ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        lo, hi = ctx.lohi
        dy = _cont(dy)
        db = grad_buf(ctx.conv.bias)[lo:hi] if ctx.conv.bias is not None else None
        with _wgrad_side(dy, x):
            ops.conv_nhwc_wgrad(dy, x, grad_buf(ctx.conv.weight)[lo:hi], db, stride=1)
        dx = ops.conv_nhwc_dgrad(dy, w[lo:hi], x.shape, stride=1) if ctx.needs_input_grad[1] else None
        return None, dx, None, None, None, None


# ---------------------------------------------------------------------------------------------------
def _lstm_wx(w, W2, C):
    """The x half W[:, :C] of a ConvLSTM 1x1 weight [4C, 2C] as a contiguous matrix, cached in ``ops.PackCache`` until the weights
    change (the x projection of the forward pass and the dx GEMM of the backward pass each made their own copy per step)."""
    buf, valid = ops.PackCache.get(w, ('lstm_wx', C), 4 * C * C)
    wx = buf[:4 * C * C].view(4 * C, C)
    if not valid:
        wx.copy_(W2[:, :C])
    return wx


class ConvLSTMSeqFn(Function):
    """DWSConvLSTM2d (models/layers/rnn.py:37-70) unrolled over a whole sequence: x_seq [T,B,H,W,C] (channels-last rows),
    initial state (h0, c0) or None.  Only this recurrence is sequential in an RVT stage -- everything in front of it
    (downsampling conv, attention blocks) is per-frame and is run time-batched on [T*B] by the caller.

    h and c of all timesteps live in one [T+1, ...] buffer each (slot 0 = initial state), so that ``h_prev`` of all
    timesteps is the contiguous view ``hbuf[:T]``: the weight gradient of the whole sequence is ONE wgrad launch over
    T*B*H*W rows, and BPTT needs two launches per timestep (gate backward with the two h-gradient sources fused, dgrad
    with the [dx | dh_prev] split epilogue)."""

    @staticmethod
    def forward(ctx, mod, x_seq, h0, c0, w, b):
        need = any(ctx.needs_input_grad)
        T, C = x_seq.shape[0], x_seq.shape[-1]
        M = x_seq[0].numel() // C
        hbuf = x_seq.new_empty((T + 1,) + tuple(x_seq.shape[1:]))
        cbuf = x_seq.new_empty((T + 1,) + tuple(x_seq.shape[1:]))
        if h0 is None:
            hbuf[0].zero_()
            cbuf[0].zero_()
        elif h0.dtype is hbuf.dtype and c0.dtype is cbuf.dtype and h0.is_contiguous() and c0.is_contiguous() and \
                h0.shape == hbuf[0].shape and c0.shape == cbuf[0].shape and ops.multi_ok([hbuf[0], cbuf[0], h0, c0]):
            ops.copy_multi([hbuf[0], cbuf[0]], [h0, c0])        # one launch for the pair
        else:
            hbuf[0].copy_(h0)
            cbuf[0].copy_(c0)
        W2 = w.view(4 * C, 2 * C)
        mode = ops.convlstm_seq_mode(C)
        # precision mode bf16 + sequence kernels: gates as fp16 in the kernels' own layout, gate gradients as bf16 rows
        g16 = bool(mode) and ops.convlstm_gates16_ok(C)
        gates = (ops.convlstm_gates16_buffer(T, M, C, x_seq.device) if g16 else x_seq.new_empty((T, M, 4, C))) if need else None
        wpack = None
        if mode:
            # ONE launch for the whole recurrence (csrc/k_lstm.hip); modes 2 / 3: the x projection of all timesteps is one large GEMM;
            # mode 3 (stage 4): the waves stream their weight fragments from a packed bf16 copy (shared with the backward pass)
            xin = x_seq
            if mode >= 2:
                xin, _, _ = ops.ln_linear_fwd(x_seq.view(T * M, C), None, None, _lstm_wx(w, W2, C), b)
            if mode == 3:
                wpack = ops.convlstm_seq_pack(W2, C)
            ops.convlstm_seq_fwd(xin, mode >= 2, hbuf, cbuf, W2, b, gates, zero_state=h0 is None, wpack=wpack)
        else:
            for t in range(T):
                zero_state = h0 is None and t == 0
                ops.convlstm_fwd(x_seq[t], None if zero_state else hbuf[t], None if zero_state else cbuf[t], W2, b,
                                 h_out=hbuf[t + 1], c_out=cbuf[t + 1], gates_out=gates[t] if need else None)
        if need:
            ctx.mod = mod
            ctx.wpack = wpack
            ctx.set_materialize_grads(False)
            ctx.save_for_backward(x_seq, hbuf, cbuf, gates, w)
        return hbuf[1:], cbuf[T]

    @staticmethod
    def backward(ctx, dh_seq, dc_last):
        x_seq, hbuf, cbuf, gates, w = ctx.saved_tensors
        mod = ctx.mod
        T, C = x_seq.shape[0], x_seq.shape[-1]
        M = x_seq[0].numel() // C
        W2 = w.view(4 * C, 2 * C)
        dh_seq = _cont(dh_seq)
        g16 = gates.dtype is torch.float16
        dgates = torch.empty((T, M, 4 * C), dtype=torch.bfloat16 if g16 else torch.float32, device=x_seq.device)
        need_h0, need_c0 = ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        dh0 = x_seq.new_empty(x_seq.shape[1:]) if need_h0 else None
        dc0 = x_seq.new_empty(x_seq.shape[1:]) if need_c0 else None
        mode = ops.convlstm_seq_mode(C)
        wpack = ctx.wpack if mode == 3 and ctx.wpack is not None else (ops.convlstm_seq_pack(W2, C) if mode == 3 else None)
        seq_ok = bool(mode) and ops.convlstm_seq_bwd(dh_seq, _cont(dc_last), gates, cbuf, W2, dgates, dh0, dc0, wpack=wpack)
        assert seq_ok or not g16, 'fp16 gates are only written when the backward sequence kernel exists'
        if seq_ok:
            # backward through time in one launch; dx of all timesteps is ONE GEMM dgates W_x over T*M rows
            dx_seq = ops.linear_dgrad(dgates.view(T * M, 4 * C), _lstm_wx(w, W2, C)).view(x_seq.shape)
            dh_next, dc_next = dh0, dc0
        else:
            dx_seq = torch.empty_like(x_seq)
            dh_next, dc_next = None, _cont(dc_last)
            for t in reversed(range(T)):
                _, dc_next = ops.convlstm_gates_bwd(dh_seq[t] if dh_seq is not None else None, dc_next, gates[t], cbuf[t], cbuf[t + 1],
                                                    want_dc_prev=(t > 0 or need_c0), dh2=dh_next, dgates_out=dgates[t])
                _, dh_next = ops.linear_dgrad(dgates[t], W2, split=C, out=dx_seq[t].view(M, C))
        with _wgrad_side(dgates, x_seq, hbuf):
            ops.linear_wgrad(dgates.view(T * M, 4 * C), x_seq.view(T * M, C), grad_buf(mod.conv1x1.weight).view(4 * C, 2 * C),
                             grad_buf(mod.conv1x1.bias), x2=hbuf[:T].view(T * M, C))
        dh0 = dh_next.view(x_seq.shape[1:]) if need_h0 else None
        dc0 = dc_next if need_c0 else None
        return None, dx_seq, dh0, dc0, None, None


# ---------------------------------------------------------------------------------------------------
class BaseConvFn(Function):
    """BaseConv (network_blocks.py:29-51): conv(no bias) -> BatchNorm2d -> SiLU on NHWC maps."""

    @staticmethod
    def forward(ctx, mod, x, conv_w, bn_w, bn_b, stride: int, training: bool):
        if not training:
            return ops.conv_nhwc_fwd(x, conv_w, None, stride=stride,
                                     bn=(bn_w, bn_b, mod.bn.running_mean, mod.bn.running_var), bn_eps=mod.bn.eps)
        need = any(ctx.needs_input_grad)
        y, saved = _conv_bn_fwd([(mod, x, conv_w, bn_w, bn_b, stride)], need)[0]
        if need:
            ctx.mod, ctx.stride = mod, stride
            ctx.count, ctx.count_dev = saved[-2], saved[-1]
            ctx.save_for_backward(x, *saved[:-2], conv_w, bn_w, bn_b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, z, mean, rstd, conv_w, bn_w, bn_b = ctx.saved_tensors
        dx = _conv_bn_bwd([(ctx.mod, x, z, mean, rstd, conv_w, bn_w, bn_b, ctx.stride, ctx.count, ctx.count_dev, _rows(dy),
                            ctx.needs_input_grad[1])])[0]
        return None, dx, None, None, None, None, None


def _conv_bn_fwd(members, need):
    """Training forward of one or several INDEPENDENT BaseConv layers: all convs (their epilogues accumulate the column
    statistics into one contiguous block), ONE SyncBatchNorm all-reduce for the whole block, then the BN + SiLU kernels.
    members: (mod, x, conv_w, bn_w, bn_b, stride) -> [(y, (z, mean, rstd, count, count_dev))]."""
    dev = members[0][1].device
    sync = _sync_bn_on()
    # the conv epilogues spread their (sum, sumsq) atomics over R copies (R by output rows), bn_silu_fwd folds them
    def rows_of(x, stride):
        # SyncBatchNorm: the block is all-reduced, so its layout must not depend on this rank's (data-dependent) image count
        return (32 if sync else x.shape[0]) * ((x.shape[1] - 1) // stride + 1) * ((x.shape[2] - 1) // stride + 1)
    reps = [ops.stat_replicas(rows_of(m[1], m[5])) for m in members]
    block = ops.StatArena.zeros((sum(2 * R * m[2].shape[0] for m, R in zip(members, reps)),), dev)
    zs, off, stat_slices = None, 0, []
    for (mod, x, conv_w, bn_w, bn_b, stride), R in zip(members, reps):
        N = conv_w.shape[0]
        stat_slices.append(block[off:off + 2 * R * N].view(R, 2, N))
        off += 2 * R * N
    # members of one channel geometry (the tower convs of equal depth over the head levels) as ONE launch
    if all(m[5] == 1 for m in members) and ops.conv3x3_group_ok([m[1] for m in members], [m[2] for m in members]):
        zs = ops.conv3x3_group_fwd([m[1] for m in members], [m[2] for m in members], stat_slices)
    if zs is None:
        zs = [ops.conv_nhwc_fwd(x, conv_w, None, stride=stride, colstats=sl) for (mod, x, conv_w, bn_w, bn_b, stride), sl in zip(members, stat_slices)]
    images = _SYNC_BN['images'] if sync else None
    if sync:
        assert images is not None, 'SyncBatchNorm: functions.sync_bn_begin(n_images) must open the pass (YoloXDetector.forward_detect does)'
        _allreduce_stats(block)
    out = []
    counts, moms = [], []
    for (mod, x, conv_w, bn_w, bn_b, stride), z in zip(members, zs):
        rows = z.numel() // conv_w.shape[0]
        counts.append(rows // z.shape[0] if sync else rows)        # SyncBN: rows per image; the kernels multiply by ``images``
        moms.append(mod.bn.momentum if mod.bn.momentum is not None else 0.1)
        mod.bn_calls_pending = getattr(mod, 'bn_calls_pending', 0) + 1     # flushed into num_batches_tracked lazily (flush_bn_counters)
    same = len(members) > 1 and len(members) <= 8 and len({(m[2].shape[0], m[0].bn.eps) for m in members}) == 1
    if same:                                                        # one launch for the layers of one channel count
        res = ops.bn_silu_fwd_group(zs, stat_slices, [m[3] for m in members], [m[4] for m in members], [m[0].bn.running_mean for m in members],
                                    [m[0].bn.running_var for m in members], counts, members[0][0].bn.eps, moms,
                                    count_devs=[images] * len(members))
    else:
        res = [ops.bn_silu_fwd(z, sl, m[3], m[4], m[0].bn.running_mean, m[0].bn.running_var, c, eps=m[0].bn.eps, momentum=mo, count_dev=images)
               for m, z, sl, c, mo in zip(members, zs, stat_slices, counts, moms)]
    for (y, mean, rstd), z, c in zip(res, zs, counts):
        out.append((y, (z, mean, rstd, c, images) if need else None))
    return out


def _conv_bn_bwd(members):
    """Backward of the same group: the (sum du, sum du*xhat) reductions of all members into one block, ONE all-reduce, then
    per member the BN backward, the weight gradient (side stream) and the input gradient.
    members: (mod, x, z, mean, rstd, conv_w, bn_w, bn_b, stride, count, count_dev, dy, need_dx) -> [dx | None]."""
    live = [m for m in members if m[11] is not None]
    dxs = {}
    if live:
        dev = live[0][1].device
        # the reductions spread their closing atomics over R copies of a member's sums (R by rows), the apply kernels fold them
        sync = _sync_bn_on()

        def rows_of(z):       # SyncBatchNorm: the block is all-reduced, so its layout must not depend on this rank's image count
            return (32 if sync else z.shape[0]) * (z.numel() // (z.shape[0] * z.shape[-1]))
        reps = [ops.bn_bwd_replicas(rows_of(m[2])) for m in live]
        block = ops.StatArena.zeros((sum(2 * R * m[5].shape[0] for m, R in zip(live, reps)),), dev)
        off = 0
        slices = []
        for (mod, x, z, mean, rstd, conv_w, bn_w, bn_b, stride, count, count_dev, dy, need_dx), R in zip(live, reps):
            N = conv_w.shape[0]
            slices.append(block[off:off + 2 * R * N].view(R, 2, N))
            off += 2 * R * N
        same = 1 < len(live) <= 8 and len({m[5].shape[0] for m in live}) == 1       # one launch per kernel kind for layers of one channel count
        if same:
            ops.bn_silu_bwd_reduce_group([m[11] for m in live], [m[2] for m in live], [m[3] for m in live], [m[4] for m in live],
                                         [m[6] for m in live], [m[7] for m in live], slices)
        else:
            for (mod, x, z, mean, rstd, conv_w, bn_w, bn_b, stride, count, count_dev, dy, need_dx), sl in zip(live, slices):
                ops.bn_silu_bwd_reduce(dy, z, mean, rstd, bn_w, bn_b, out=sl)
        _allreduce_stats(block)
        # SyncBatchNorm while the step is being recorded into launch plans: every exchange closes a plan segment, and a segment can only
        # end with the side stream joined -- the small weight gradients of the neck / head then stay on the launch stream instead of
        # forking and joining once per layer (17.97 -> see profiles/r04_*_rccl_force_collectives.txt)
        from .modules.step_plan import PlanRecorder
        hold = sync and PlanRecorder.current is not None and not _capture_collectives(_SYNC_BN['group'])
        prev_hold, WgradSide.hold_main = WgradSide.hold_main, hold or WgradSide.hold_main
        shared_dx = {}
        rounds = ([], [])                                   # dgrad problems: (dz, conv_w, x.shape, stride, out, accumulate) -- first writers, then adders
        if same:
            dzs = ops.bn_silu_bwd_apply_group([m[11] for m in live], [m[2] for m in live], [m[3] for m in live], [m[4] for m in live],
                                              [m[6] for m in live], [m[7] for m in live], slices, [grad_buf(m[0].bn.weight) for m in live],
                                              [grad_buf(m[0].bn.bias) for m in live], [m[9] for m in live], count_devs=[m[10] for m in live])
        else:
            dzs = [ops.bn_silu_bwd_apply(dy, z, mean, rstd, bn_w, bn_b, sl, grad_buf(mod.bn.weight), grad_buf(mod.bn.bias), count, count_dev=count_dev)
                   for (mod, x, z, mean, rstd, conv_w, bn_w, bn_b, stride, count, count_dev, dy, need_dx), sl in zip(live, slices)]
        wg_done = False
        if all(m[8] == 1 for m in live) and ops.conv3x3_group_ok([m[1] for m in live], [m[5] for m in live]):
            with _wgrad_side(*dzs, *[m[1] for m in live]):         # the weight gradients of the group: one launch on the side lane
                wg_done = ops.conv3x3_group_wgrad(dzs, [m[1] for m in live], [grad_buf(m[0].conv.weight) for m in live])
        for (mod, x, z, mean, rstd, conv_w, bn_w, bn_b, stride, count, count_dev, dy, need_dx), dz in zip(live, dzs):
            if not wg_done:
                with _wgrad_side(dz, x):
                    ops.conv_nhwc_wgrad(dz, x, grad_buf(mod.conv.weight), None, stride=stride)
            if not need_dx:
                dxs[id(mod)] = None
                continue
            # members that read the SAME input (conv1 / conv2 of a CSP layer, the first cls / reg tower convs of a head level) add their input
            # gradients in the dgrad epilogue of the later member instead of as two autograd edges summed by an extra add kernel
            key = (x.data_ptr(), tuple(x.shape), stride)
            if key in shared_dx:
                rounds[1].append((dz, conv_w, x.shape, stride, shared_dx[key], True))
                dxs[id(mod)] = None
            else:
                dxs[id(mod)] = shared_dx[key] = ops._empty(tuple(x.shape), dz)
                rounds[0].append((dz, conv_w, x.shape, stride, shared_dx[key], False))
        for probs in rounds:
            # the input gradients of one round write different buffers: one launch where the members share a channel geometry (head towers)
            done = False
            if len(probs) > 1 and all(p[3] == 1 for p in probs) and ops.conv3x3_group_ok([p[0] for p in probs], [p[1].transpose(0, 1) for p in probs]):
                done = ops.conv3x3_group_dgrad([p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs], [p[4] for p in probs],
                                               [p[5] for p in probs])
            if not done:
                for dz, conv_w, xshape, stride, out, acc in probs:
                    ops.conv_nhwc_dgrad(dz, conv_w, xshape, stride=stride, out=out, accumulate=acc)
        WgradSide.hold_main = prev_hold
    return [dxs.get(id(m[0])) for m in members]


class BaseConvGroupFn(Function):
    """Several independent BaseConv layers evaluated as ONE autograd node so that their SyncBatchNorm statistics travel in one
    all-reduce per direction (the three head levels' stems, the cls / reg tower layers of equal depth, conv1 / conv2 of a CSP
    layer).  Only used when SyncBatchNorm is on; values are identical to separate ``BaseConvFn`` calls.
    apply(mods, x_0..x_{n-1}, (conv_w, bn_w, bn_b) x n) -> y_0..y_{n-1}"""

    @staticmethod
    def forward(ctx, mods, *tensors):
        n = len(mods)
        xs, params = tensors[:n], tensors[n:]
        need = any(ctx.needs_input_grad)
        res = _conv_bn_fwd([(mods[i], xs[i], params[3 * i], params[3 * i + 1], params[3 * i + 2], mods[i].stride) for i in range(n)], need)
        if need:
            ctx.mods = mods
            ctx.meta = [(r[1][3], r[1][4]) for r in res]
            flat = []
            for i, r in enumerate(res):
                flat += [xs[i], r[1][0], r[1][1], r[1][2], params[3 * i], params[3 * i + 1], params[3 * i + 2]]
            ctx.save_for_backward(*flat)
        return tuple(r[0] for r in res)

    @staticmethod
    def backward(ctx, *dys):
        sv, mods = ctx.saved_tensors, ctx.mods
        n = len(mods)
        members = []
        for i in range(n):
            x, z, mean, rstd, conv_w, bn_w, bn_b = sv[7 * i:7 * i + 7]
            members.append((mods[i], x, z, mean, rstd, conv_w, bn_w, bn_b, mods[i].stride, ctx.meta[i][0], ctx.meta[i][1],
                            _rows(dys[i]), ctx.needs_input_grad[1 + i]))
        dxs = _conv_bn_bwd(members)
        return (None,) + tuple(dxs) + (None,) * (3 * n)


def base_conv_group(mods, xs):
    """[BaseConv], [NHWC maps] -> [outputs]: independent layers as ONE autograd node in training -- one statistics exchange per direction under
    SyncBatchNorm, members of one channel geometry as one launch per kernel kind (``ops.conv3x3_group_*``), input gradients of members
    that read the same map summed inside the node; plain per-layer calls in evaluation."""
    if len(mods) > 1 and mods[0].training and (_sync_bn_on() or (torch.is_grad_enabled() and xs[0].is_cuda)):
        params = []
        for m in mods:
            params += [m.conv.weight, m.bn.weight, m.bn.bias]
        return list(BaseConvGroupFn.apply(tuple(mods), *xs, *params))
    return [m.forward_nhwc(x) for m, x in zip(mods, xs)]


# ---------------------------------------------------------------------------------------------------
class HeadTailFn(Function):
    """Prediction convs + decode + SimOTA + losses of YOLOXHead (yolo_head.py:216-287,403-597).

    inputs : labels [B,N,7], (cls_feat_k, reg_feat_k) per level (NHWC), then per level
             (cls_w, cls_b, reg_w, reg_b, obj_w, obj_b)
    outputs: losses[6] = (loss, iou_loss, conf_loss, cls_loss, l1_loss, num_fg ratio), decoded predictions.
    Only ``losses[0]`` is differentiable (that is what the reference back-propagates)."""

    @staticmethod
    def forward(ctx, mod, labels, *tensors):
        nl = len(mod.strides)
        feats, params = tensors[:2 * nl], tensors[2 * nl:]
        B = feats[0].shape[0]
        hws = [tuple(feats[2 * k].shape[1:3]) for k in range(nl)]
        A = sum(h * w for h, w in hws)
        nc = mod.num_classes
        dev = feats[0].device
        out_train = torch.empty((B, A, 5 + nc), dtype=torch.float32, device=dev)
        out_infer = torch.empty((B, A, 5 + nc), dtype=torch.float32, device=dev)
        a0 = 0
        offs = []
        for k in range(nl):
            cw, cb, rw, rb, ow, ob = params[6 * k:6 * k + 6]
            ops.head_pred_fwd(feats[2 * k], feats[2 * k + 1], cw.view(nc, -1), cb, rw.view(4, -1), rb, ow.view(1, -1), ob,
                              out_train, out_infer, mod.strides[k], a0)
            offs.append(a0)
            a0 += hws[k][0] * hws[k][1]
        labels = mod._ignore_bbox(labels)
        asg = ops.simota_assign(out_train, labels, hws, mod.strides, ignore_label=float(mod.ignore_label))
        if mod.ignore_bg_k > 0:                                  # top fraction of the background objectness logits leaves the loss (:541-542)
            ops.bg_topk_ignore(out_train, labels, asg, mod.ignore_bg_k, ignore_label=float(mod.ignore_label))
        need = any(ctx.needs_input_grad)
        losses, d_raw = ops.yolox_loss(out_train, labels, asg, hws, mod.strides, want_grad=need,
                                       focal=mod.obj_focal_loss, reg_weight=mod.reg_weight, obj_weight=mod.obj_weight,
                                       cls_weight=mod.cls_weight, label_w=mod._bbox_label_weights(labels))
        mod.last_assignment = asg
        if need:
            ctx.mod, ctx.offs = mod, offs
            ctx.save_for_backward(d_raw, *feats, *params)
        ctx.mark_non_differentiable(out_infer)
        return losses, out_infer

    @staticmethod
    def backward(ctx, dlosses, _dout):
        mod = ctx.mod
        nl = len(mod.strides)
        saved = ctx.saved_tensors
        d_raw, feats, params = saved[0], saved[1:1 + 2 * nl], saved[1 + 2 * nl:]
        nc = mod.num_classes
        gscale = dlosses[0:1].contiguous()           # device scalar: d(loss) seed (1.0 for loss.backward())
        grads = []
        for k in range(nl):
            cw, cb, rw, rb, ow, ob = params[6 * k:6 * k + 6]
            dcf, drf = ops.head_pred_bwd(d_raw, feats[2 * k], feats[2 * k + 1], cw.view(nc, -1), rw.view(4, -1), ow.view(1, -1),
                                         grad_buf(mod.cls_preds[k].weight), grad_buf(mod.cls_preds[k].bias),
                                         grad_buf(mod.reg_preds[k].weight), grad_buf(mod.reg_preds[k].bias),
                                         grad_buf(mod.obj_preds[k].weight), grad_buf(mod.obj_preds[k].bias),
                                         ctx.offs[k], gscale=gscale)
            grads += [dcf, drf]
        return (None, None) + tuple(grads) + (None,) * len(params)


class Cat2Fn(Function):
    """cat([upsample2(a) if up else a, b], channel axis) on NHWC maps as ONE launch each way (``ops.cat2_up_fwd`` / ``_bwd``): the PAFPN
    top-down joins (yolo_pafpn.py:113-123) and the CSPLayer join (network_blocks.py:160-166) were expand + copy + cat forward and two slice
    copies + a reduction backward."""

    @staticmethod
    def forward(ctx, a, b, up):
        ctx.ca, ctx.up = a.shape[-1], bool(up)
        return ops.cat2_up_fwd(a.contiguous(), b.contiguous(), ctx.up)

    @staticmethod
    def backward(ctx, dout):
        da, db = ops.cat2_up_bwd(dout.contiguous(), ctx.ca, ctx.up)
        return da, db, None


def cat2_nhwc(a: torch.Tensor, b: torch.Tensor, up: bool = False) -> torch.Tensor:
    """Channel concat of two NHWC maps, ``a`` nearest-upsampled x2 first if ``up``.  HIP fp32 maps with channel counts % 4 == 0 take the
    fused kernel; anything else the torch ops it replaces."""
    ok = (a.is_cuda and b.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 4 and b.dim() == 4 and
          a.shape[-1] % 4 == 0 and b.shape[-1] % 4 == 0 and (not up or (b.shape[1] % 2 == 0 and b.shape[2] % 2 == 0)))
    if ok:
        return Cat2Fn.apply(a, b, up)
    if up:
        B, H, W, C = a.shape
        a = a[:, :, None, :, None, :].expand(B, H, 2, W, 2, C).reshape(B, 2 * H, 2 * W, C)
    return torch.cat((a, b), dim=-1)


class PickLossFn(Function):
    """losses[0] as its own autograd node.  Plain indexing would put a SelectBackward node in front of ``HeadTailFn``: a zero fill of a
    [6] tensor plus a 4-byte device-to-device memcpy per backward pass -- and a memcpy NODE in a captured step, which a launch plan
    (``ops.LaunchPlan``) cannot read back from the graph.  The backward hands on a real zero-padded vector [g, 0, 0, 0, 0, 0] written by
    one small launch (``leod_onehot_scale``), so a node placed between this one and ``HeadTailFn`` sees the true gradient."""

    @staticmethod
    def forward(ctx, losses):
        ctx.n = losses.numel()
        return losses[0]

    @staticmethod
    def backward(ctx, g):
        if g.is_cuda and g.dtype is torch.float32:
            return ops.onehot_scale(g.contiguous(), ctx.n, 0)
        out = g.new_zeros(ctx.n)
        out[0] = g
        return out


# ---------------------------------------------------------------------------------------------------
# Composed attention block for the OPTIONS no shipped config enables (gated MLP, other activations, torch-MHA parameter layout, no
# LayerScale, no biases: models/layers/maxvit/maxvit.py:56-118,185-270,307-325).  The shipped block is ONE autograd node over fused kernels
# (AttnBlockFn); these variants chain the same C entry points as separate nodes on fp32 rows -- correct in every precision mode, not tuned.
# Parameter gradients are RETURNED to autograd (the torch-MHA layout routes them through an index_select of the in-projection rows).
class LNLinearFn(Function):
    """y = LN(x) W^T + b  (``ln_w`` None: plain rows) on [M, K] fp32 rows."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, W, b):
        y, _, st = ops.ln_linear_fwd(x, ln_w, ln_b, W, b, want_stats=ln_w is not None)
        ctx.save_for_backward(x, st, ln_w, ln_b, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, st, ln_w, ln_b, W = ctx.saved_tensors
        dy = _cont(dy)
        dW, db = torch.zeros_like(W), torch.zeros(W.shape[0], dtype=W.dtype, device=W.device)
        if ln_w is not None:
            ops.linear_wgrad(dy, x, dW, db, stats=st, ln_w=ln_w, ln_b=ln_b)
            dlw, dlb = torch.zeros_like(ln_w), torch.zeros_like(ln_w)
            dx = ops.layernorm_bwd(ops.linear_dgrad(dy, W), x, st, ln_w, None, dlw, dlb)
            return dx, dlw, dlb, dW, db
        ops.linear_wgrad(dy, x, dW, db)
        return ops.linear_dgrad(dy, W), None, None, dW, db


class AttnCoreFn(Function):
    """Partition attention core on fp32 qkv rows [B, H, W, 3C] (head h owns columns [3dh, 3d(h+1)) = q | k | v)."""

    @staticmethod
    def forward(ctx, qkv, heads: int, part, window: bool):
        o, lse = ops.partition_attn_fwd(qkv, heads, part, window, want_lse=True)
        ctx.save_for_backward(qkv, lse)
        ctx.geom = (heads, part, window)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, lse = ctx.saved_tensors
        heads, part, window = ctx.geom
        return ops.partition_attn_bwd(qkv, _cont(do), lse, heads, part, window), None, None, None


class LinearScaleResFn(Function):
    """z = res + gamma * (h W^T + b) on fp32 rows (proj / fc2 + LayerScale + residual; gamma = ones: no LayerScale)."""

    @staticmethod
    def forward(ctx, h, W, b, gamma, res):
        z, _ = ops.linear_lsres_fwd(h, W, b, gamma, res, want_t=False, a_gelu=False)
        ctx.save_for_backward(h, W, b, gamma)
        return z

    @staticmethod
    def backward(ctx, dz):
        h, W, b, gamma = ctx.saved_tensors
        dz = _cont(dz)
        dh = ops.linear_dgrad(dz, W, kscale=gamma)
        dW, db, dg = torch.zeros_like(W), torch.zeros_like(b), torch.zeros_like(gamma)
        ops.layerscale_linear_wgrad(dz, h, W, b, gamma, dW, db, dg, h_gelu=False)
        return dh, dW, db, dg, dz


class ActGluFn(Function):
    """h = a * act(g) with (a | g) = halves of p (GLU.forward, maxvit.py:80-82), or h = act(p)."""

    @staticmethod
    def forward(ctx, p, act: str, gated: bool):
        ctx.save_for_backward(p)
        ctx.opt = (act, gated)
        return ops.act_glu_fwd(p, act, gated)

    @staticmethod
    def backward(ctx, dh):
        (p,) = ctx.saved_tensors
        return ops.act_glu_bwd(p, _cont(dh), *ctx.opt), None, None


class TokenMaskFn(Function):
    """x[token_mask] = mask_token (maxvit_rnn.py:190-192) on a channels-last map; the masked rows' gradient goes to the token."""

    @staticmethod
    def forward(ctx, x, mask, token):
        y = x.clone()
        ops.token_mask_fwd_(y, mask, token.reshape(-1))
        ctx.save_for_backward(mask)
        ctx.tshape = token.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dx = dy.contiguous().clone()
        dt = torch.zeros(dx.shape[-1], dtype=dx.dtype, device=dx.device)
        ops.token_mask_bwd_(dx, mask, dt)
        return dx, None, dt.view(ctx.tshape)
