"""Host -> HBM staging of loader batches one step ahead, on its own HIP stream.

Lightning moves a batch with ``transfer_batch_to_device`` on the launch stream right before ``training_step``: a 245 MB RVT-S Gen1
batch is ~5 ms of PCIe that the GPU would spend idle.  ``DevicePrefetcher`` wraps any loader: batch i + 1 is moved (the module's
own ``transfer_batch_to_device``: ONE copy of the pinned [L,B,20,H,W] buffer, labels stay on the host) on a copy stream while step
i runs; the launch stream only waits for the copy's event.  Use as ``for batch in DevicePrefetcher(loader, module, device)``."""
from typing import Any, Iterable, Iterator

import torch


def _record_stream(obj: Any, stream) -> None:
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream)


class DevicePrefetcher:
    def __init__(self, loader: Iterable, module, device, depth: int = 1):
        self.loader, self.module, self.device, self.depth = loader, module, torch.device(device), max(1, depth)
        self.copy_stream = torch.cuda.Stream(device=self.device)

    def _stage(self, batch):
        with torch.cuda.stream(self.copy_stream):
            moved = self.module.transfer_batch_to_device(batch, self.device, 0)
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        return moved, ready, batch                               # the host batch stays alive until its copy has been consumed

    def __iter__(self) -> Iterator[Any]:
        it = iter(self.loader)
        staged = []
        for batch in it:
            staged.append(self._stage(batch))
            if len(staged) > self.depth:
                yield self._hand_over(staged.pop(0))
        while staged:
            yield self._hand_over(staged.pop(0))

    def _hand_over(self, item):
        moved, ready, _host = item
        main = torch.cuda.current_stream(self.device)
        main.wait_event(ready)
        _record_stream(moved, main)                              # the caching allocator must not recycle the buffers early
        return moved
