"""The exchange of the data-parallel step on the library's own RCCL communicator (csrc/k_comm.hip).

``torch.distributed`` stays the rendezvous (``init_process_group``), the parameter broadcast and the barriers; the all-reduces INSIDE a
step -- SyncBatchNorm statistics of the detection head (two per BatchNorm depth), the gradient buckets (reference: Lightning DDP with
``sync_batchnorm``, train.py:131-133,247) -- go through ``leod_comm_allreduce``: enqueued from C on the stream the surrounding kernels run
on, no process-group layer, no hop to a communication stream and back.  Measured with one rank and every collective issued
(profiles/r06_h_native_comm.txt): 17.5 ms per step through ``dist.all_reduce`` between plan segments, 16.7 eager.

``setup`` is collective.  It verifies the new communicator against ``dist.all_reduce`` on a test vector and ALL ranks agree on the result
(a MIN reduction of the verdicts): any failure anywhere leaves every rank on torch.distributed.  ``LEOD_DIST_BACKEND=nccl`` (an explicit
torch backend) skips it."""
import ctypes
import os

import torch

from . import _lib

_DTYPE = {torch.float32: 0, torch.float64: 1, torch.bfloat16: 2}


class NativeComm:
    active = False
    world = 0
    n_calls = 0

    @classmethod
    def setup(cls, group=None) -> bool:
        import torch.distributed as dist
        if cls.active:
            return True
        if not (dist.is_available() and dist.is_initialized()) or group is not None:
            return False                                      # (sub-groups stay on torch.distributed)
        if os.environ.get('LEOD_DIST_BACKEND') or dist.get_backend() != 'nccl' or not torch.cuda.is_available():
            return False
        lib = _lib.lib()
        rank, world = dist.get_rank(), dist.get_world_size()
        ids = [None]
        if rank == 0:
            buf = ctypes.create_string_buffer(128)
            if lib.leod_comm_unique_id(buf) == 0:
                ids = [buf.raw]
        dist.broadcast_object_list(ids, src=0)
        ok = ids[0] is not None and lib.leod_comm_init(ctypes.create_string_buffer(ids[0], 128), rank, world) == 0
        dev = torch.device('cuda', torch.cuda.current_device())
        if ok:
            # the same vectors through both communicators: integers below 2^24 per element, so the sums are exact in any order
            probe = [(torch.arange(4099, device=dev) % 251 + rank).to(dt) for dt in (torch.float32, torch.float64, torch.bfloat16)]
            ref = [p.clone() for p in probe]
            s = torch.cuda.current_stream().cuda_stream
            for p in probe:
                ok = ok and lib.leod_comm_allreduce(ctypes.c_void_p(p.data_ptr()), p.numel(), _DTYPE[p.dtype], ctypes.c_void_p(s)) == 0
            for r in ref:
                dist.all_reduce(r)
            torch.cuda.synchronize()
            ok = ok and all(torch.equal(p, r) for p, r in zip(probe, ref))
        verdict = torch.tensor([1.0 if ok else 0.0], device=dev)
        dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
        cls.active = bool(verdict.item() == 1.0)
        cls.world = world if cls.active else 0
        if not cls.active:
            lib.leod_comm_destroy()
        return cls.active

    @classmethod
    def all_reduce(cls, t: torch.Tensor, stream=None) -> None:
        """t <- sum over ranks, in place, ordered on ``stream`` (default: torch's current stream)."""
        if not t.is_cuda or not t.is_contiguous() or t.dtype not in _DTYPE:
            raise _lib.LeodHipError(f'NativeComm.all_reduce: contiguous float32 / float64 / bfloat16 device tensor expected, got {t.dtype} on {t.device}')
        s = (stream if stream is not None else torch.cuda.current_stream(t.device)).cuda_stream
        rc = _lib.lib().leod_comm_allreduce(ctypes.c_void_p(t.data_ptr()), t.numel(), _DTYPE[t.dtype], ctypes.c_void_p(s))
        if rc != 0:
            raise _lib.LeodHipError(f'leod_comm_allreduce failed: {_lib.lib().leod_comm_last_error().decode()}')
        cls.n_calls += 1

    @classmethod
    def usable(cls, t: torch.Tensor, group=None) -> bool:
        return cls.active and group is None and t.is_cuda and t.dtype in _DTYPE and t.is_contiguous()

    @classmethod
    def shutdown(cls) -> None:
        if cls.active:
            _lib.lib().leod_comm_destroy()
            cls.active, cls.world = False, 0
