"""Host-side resources of a one-process-per-GPU job.

The launch thread of a training / pseudo-labelling process must never be descheduled: it enqueues ~900 kernel launches per RVT-S
step in ~13 ms while the GPU works ~31 ms on them.  PyTorch sizes its intra-op (OpenMP) pool by the VISIBLE cpus (128 threads
on a 256-cpu MI355X host) although containers usually run under a cgroup CPU quota (16 cpus on the measurement hosts).  Every
parallel region then leaves 128 threads spinning, the process burns its quota within a few milliseconds and the kernel
throttles ALL of its threads for the rest of the 100 ms CFS period -- measured on MI355X: 12 of 25 periods throttled, host
enqueue time of single steps jumping from 13 ms to 50-110 ms, 42.7 instead of 31.2 ms per step
(profiles/r02_l_host_thread_throttle.txt).  ``bound_host_threads`` caps the pool by the quota; it is called by ``Module.setup``
and by the engines, i.e. before the first backward pass creates the autograd worker thread (which copies the setting).
"""
import os

import torch

_DONE = False


def usable_cores() -> int:
    """Host cores this process may actually use: min(affinity mask, cgroup v2 / v1 CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


def bound_host_threads(local_world_size: int = None, force: bool = False) -> int:
    """Cap torch's intra-op thread pool at min(current, 4, usable cores / (2 x ranks on this node)).  The hot path has no CPU
    arithmetic (labels are packed by numpy-sized copies), so the cap costs nothing.  An explicit ``OMP_NUM_THREADS`` or
    ``LEOD_HOST_THREADS`` wins.  Idempotent; returns the thread count in force."""
    global _DONE
    if _DONE and not force:
        return torch.get_num_threads()
    _DONE = True
    env = os.environ.get('LEOD_HOST_THREADS')
    if env:
        torch.set_num_threads(max(1, int(env)))
        return torch.get_num_threads()
    if os.environ.get('OMP_NUM_THREADS'):                      # the launcher decided (torch.distributed.run sets 1 for N > 1)
        return torch.get_num_threads()
    if local_world_size is None:
        local_world_size = int(os.environ.get('LOCAL_WORLD_SIZE', '1') or 1)
    cap = max(1, min(4, usable_cores() // (2 * max(1, local_world_size))))
    if torch.get_num_threads() > cap:
        torch.set_num_threads(cap)
    return torch.get_num_threads()
