"""Data transforms of the episode pipeline, with the reference's names, arguments and arithmetic
(``multimodal_rssm/models/transform.py``), working on tensors of any device.

``Compose`` stands in for ``torchvision.transforms.Compose`` (the YAMLs' chain container, default.yaml:176-220;
torchvision is not a dependency here).  ``EpisodeDataModule`` recognises the chains ``[TakeFirstN(n)]`` and
``[TakeFirstN(n), GaussianNoise(std)]`` and runs them fused in ``mtrssm_episode_gather``; any other callable is
applied as is, per episode, to the device-resident tensor.
"""

from __future__ import annotations

from collections.abc import Callable, Sequence

import torch
from torch import Tensor

Transform = Callable[[Tensor], Tensor]


class Compose:
    """``torchvision.transforms.Compose``: apply the transforms in order."""

    def __init__(self, transforms: Sequence[Transform]) -> None:
        self.transforms = list(transforms)

    def __call__(self, data: Tensor) -> Tensor:
        for t in self.transforms:
            data = t(data)
        return data


class RemoveDim:
    """Drop ``indices_to_remove`` along ``axis`` (``transform.py:8-27``)."""

    def __init__(self, axis: int, indices_to_remove: list[int]) -> None:
        self.axis = axis
        self.dropped = frozenset(int(i) for i in indices_to_remove)

    def __call__(self, data: Tensor) -> Tensor:
        kept = torch.tensor([i for i in range(data.size(self.axis)) if i not in self.dropped], device=data.device, dtype=torch.long)
        return data.index_select(self.axis, kept)


class TakeFirstN:
    """First ``n`` timesteps of a time-major tensor (``transform.py:30-52``)."""

    def __init__(self, n: int) -> None:
        self.n = n

    def __call__(self, data: Tensor) -> Tensor:
        return data[: self.n]


class GaussianNoise:
    """``data + randn_like(data) * std`` (``transform.py:55-72``)."""

    def __init__(self, std: float = 0.1) -> None:
        self.std = std

    def __call__(self, data: Tensor) -> Tensor:
        return data + torch.randn_like(data) * self.std


class NormalizeVisionImage:
    """[0, 255] -> [-1, 1]: ``x / 255 * 2 - 1``, never in place (``transform.py:75-98``)."""

    def __call__(self, data: Tensor) -> Tensor:
        return (data.detach() / 255.0) * 2.0 - 1.0


class NormalizeAudioMelSpectrogram:
    """[min, max] -> [-1, 1]: ``(x - min) / (max - min) * 2 - 1``, never in place (``transform.py:101-132``)."""

    def __init__(self, min_value: float = -80.0, max_value: float = 0.1) -> None:
        self.min_value = min_value
        self.max_value = max_value

    def __call__(self, data: Tensor) -> Tensor:
        span = self.max_value - self.min_value
        return ((data.detach() - self.min_value) / span) * 2.0 - 1.0


def fused_chain(transform: Transform | None) -> tuple[int | None, float | None] | None:
    """``(n, std)`` when ``transform`` is ``[TakeFirstN(n)]`` (std None) / ``[TakeFirstN(n), GaussianNoise(std)]`` /
    identity (``(None, None)``) -- the chains ``mtrssm_episode_gather`` implements; ``None`` for anything else."""
    if transform is None or isinstance(transform, torch.nn.Identity):
        return (None, None)
    chain = list(transform.transforms) if hasattr(transform, "transforms") else [transform]
    n: int | None = None
    std: float | None = None
    for i, t in enumerate(chain):
        if type(t) is TakeFirstN and i == 0:  # noqa: E721
            n = int(t.n)
        elif type(t) is GaussianNoise and i == len(chain) - 1:  # noqa: E721
            std = float(t.std)
        elif isinstance(t, torch.nn.Identity):
            continue
        else:
            return None
    return (n, std)
