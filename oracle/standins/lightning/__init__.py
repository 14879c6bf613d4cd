"""Stand-in for lightning 2.6.0 (absent): the minimum the reference's model files touch."""
import torch
from torch import nn


class LightningModule(nn.Module):
    @property
    def device(self) -> torch.device:
        for p in self.parameters():
            return p.device
        return torch.device("cpu")

    def log_dict(self, *args, **kwargs) -> None:  # noqa: ANN002, ANN003
        return None


class Callback:
    pass


class LightningDataModule:
    pass


class Trainer:
    pass
