import torch


class LightningModule(torch.nn.Module):
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def log_dict(self, *a, **k):
        pass


class LightningDataModule:
    pass


class Trainer:
    pass


class Callback:
    pass


def seed_everything(*a, **k):
    pass
