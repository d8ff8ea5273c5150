"""Lightning strategy for N > 1 GPUs (reference: ``DDPStrategy`` in train.py:131-133).

The autograd Functions of this package write parameter gradients straight into the flat gradient buffer and return ``None`` to
autograd (21 timesteps accumulate in place, see ``leod_amd.functions``), so a ``DistributedDataParallel`` wrapper would wait for
gradient hooks that never fire.  ``LeodDDPStrategy`` is Lightning's DDP strategy minus the wrapper: it launches / joins the process
group, places the module on its device and leaves the gradient exchange to ``FlatAdamW.step`` (one all-reduce of the flat
buffer over RCCL) and the BatchNorm statistics to ``leod_amd.functions.set_sync_batchnorm`` (switched on by
``Module.configure_optimizers`` when the world size is > 1).  Only importable with pytorch_lightning installed."""
try:  # pragma: no cover - pytorch_lightning is not part of the MI355X image
    from pytorch_lightning.strategies import DDPStrategy
except ImportError:  # pragma: no cover
    DDPStrategy = None


if DDPStrategy is not None:  # pragma: no cover
    class LeodDDPStrategy(DDPStrategy):
        strategy_name = 'leod_ddp'

        def _setup_model(self, model):
            """No DistributedDataParallel wrapper: the LightningModule itself is the model the Trainer calls."""
            return model

        def configure_ddp(self) -> None:
            self.model = self._setup_model(self.model)

        def _register_ddp_hooks(self) -> None:
            return None
else:
    class LeodDDPStrategy:  # pragma: no cover
        def __init__(self, *a, **k):
            raise ImportError('leod_amd.strategy.LeodDDPStrategy needs pytorch_lightning; without it drive the module with '
                              'leod_amd.optim.fit_step under torch.distributed.run (see INTEGRATION.md)')
