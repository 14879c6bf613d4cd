"""Multi-one-hot categorical latent: the distribution type the build owns.

The reference takes these from the un-vendored ``distribution_extension`` package
(``MultiOneHotFactory``, ``Distribution``, ``kl_divergence``, ``utils.stack_distribution`` /
``cat_distribution``; call sites ``networks.py:65,83,146,172``, ``state.py:17,134,151``,
``core.py:134,212-216``).  This module provides the same names and behaviour for host-side use
(State construction, callbacks, evaluation).  Inside the rollout the same arithmetic runs in the HIP
scan kernels; these classes only wrap the tensors those kernels produce.

K = ``category_size`` categoricals with C = ``class_size`` classes each; flat size S = K*C; softmax over
the class axis.  Samples are straight-through one-hots drawn by inverse CDF from uniforms
(``torch.rand`` on the tensor's device unless a noise source is installed with ``inject_uniforms``).
"""

from __future__ import annotations

import contextlib
from collections import deque
from collections.abc import Iterable, Iterator

import torch
from torch import Tensor, nn

KL_BALANCE_ALPHA = 0.8

_NOISE: deque[Tensor] | None = None


@contextlib.contextmanager
def inject_uniforms(uniforms: Iterable[Tensor]) -> Iterator[None]:
    """Feed ``MultiOneHot.rsample`` from a FIFO of uniform tensors (tests / reproducible rollouts)."""
    global _NOISE  # noqa: PLW0603
    prev, _NOISE = _NOISE, deque(uniforms)
    try:
        yield
    finally:
        _NOISE = prev


def draw_uniforms(shape: torch.Size | tuple[int, ...], like: Tensor) -> Tensor:
    if _NOISE:
        u = _NOISE.popleft()
        if tuple(u.shape) != tuple(shape):
            msg = f"injected uniforms have shape {tuple(u.shape)}, draw needs {tuple(shape)}"
            raise ValueError(msg)
        return u.to(device=like.device, dtype=like.dtype)
    return torch.rand(shape, device=like.device, dtype=like.dtype)


def onehot_from_uniforms(probs: Tensor, u: Tensor) -> Tensor:
    """Inverse-CDF one-hot: index = #{c <= C-2 : cumsum(probs)[c] <= u}.  probs [*,K,C], u [*,K]."""
    cdf = probs.cumsum(dim=-1)[..., :-1]
    idx = (cdf <= u.unsqueeze(-1)).sum(dim=-1)
    return torch.nn.functional.one_hot(idx, probs.shape[-1]).to(probs.dtype)


class MultiOneHot:
    """K independent categoricals; ``logits`` are normalised log-probs ``[*, K, C]``."""

    __slots__ = ("logits", "probs", "event_dims")

    def __init__(self, logits: Tensor, probs: Tensor | None = None, event_dims: int = 0) -> None:
        self.logits = logits
        self.probs = logits.exp() if probs is None else probs
        self.event_dims = event_dims

    @classmethod
    def from_flat_logits(cls, flat: Tensor, category_size: int, class_size: int) -> "MultiOneHot":
        shaped = flat.reshape(*flat.shape[:-1], category_size, class_size)
        return cls(torch.log_softmax(shaped, dim=-1), torch.softmax(shaped, dim=-1))

    def _apply(self, fn) -> "MultiOneHot":  # noqa: ANN001
        return MultiOneHot(fn(self.logits), fn(self.probs), self.event_dims)

    @property
    def batch_shape(self) -> torch.Size:
        return self.probs.shape[:-2]

    def rsample(self) -> Tensor:
        u = draw_uniforms(self.probs.shape[:-1], self.probs)
        onehot = onehot_from_uniforms(self.probs.detach(), u)
        return (onehot + (self.probs - self.probs.detach())).flatten(start_dim=-2)

    def independent(self, ndims: int) -> "MultiOneHot":
        return MultiOneHot(self.logits, self.probs, ndims)

    def __getitem__(self, loc) -> "MultiOneHot":  # noqa: ANN001
        return self._apply(lambda x: x[loc])

    def to(self, device) -> "MultiOneHot":  # noqa: ANN001
        return self._apply(lambda x: x.to(device))

    def detach(self) -> "MultiOneHot":
        return self._apply(lambda x: x.detach())

    def clone(self) -> "MultiOneHot":
        return self._apply(lambda x: x.clone())

    def _dim(self, dim: int, *, insert: bool) -> int:
        nbatch = self.probs.dim() - 2
        return dim if dim >= 0 else dim + nbatch + (1 if insert else 0)

    def squeeze(self, dim: int) -> "MultiOneHot":
        return self._apply(lambda x: x.squeeze(self._dim(dim, insert=False)))

    def unsqueeze(self, dim: int) -> "MultiOneHot":
        return self._apply(lambda x: x.unsqueeze(self._dim(dim, insert=True)))


Distribution = MultiOneHot


class MultiOneHotFactory(nn.Module):
    """``forward(flat_logits[*, K*C]) -> MultiOneHot`` (YAML: ``l_dist`` / ``h_dist``, mmtrssm yaml 138-147)."""

    def __init__(self, class_size: int, category_size: int) -> None:
        super().__init__()
        self.class_size = int(class_size)
        self.category_size = int(category_size)

    def forward(self, logits: Tensor) -> MultiOneHot:
        return MultiOneHot.from_flat_logits(logits, self.category_size, self.class_size)


def _kl(q: MultiOneHot, p: MultiOneHot) -> Tensor:
    return (q.probs * (q.logits - p.logits)).sum(dim=(-1, -2))


def kl_divergence(q: MultiOneHot, p: MultiOneHot, use_balancing: bool = False) -> Tensor:  # noqa: FBT001, FBT002
    """Mean over batch dims of sum_K KL(q_k || p_k); with balancing 0.8 KL(sg q||p) + 0.2 KL(q||sg p)."""
    if use_balancing:
        return KL_BALANCE_ALPHA * _kl(q.detach(), p).mean() + (1.0 - KL_BALANCE_ALPHA) * _kl(q, p.detach()).mean()
    return _kl(q, p).mean()


def stack_distribution(dists: list[MultiOneHot], dim: int) -> MultiOneHot:
    d = dists[0]._dim(dim, insert=True)
    return MultiOneHot(torch.stack([x.logits for x in dists], d), torch.stack([x.probs for x in dists], d), dists[0].event_dims)


def cat_distribution(dists: list[MultiOneHot], dim: int) -> MultiOneHot:
    d = dists[0]._dim(dim, insert=False)
    return MultiOneHot(torch.cat([x.logits for x in dists], d), torch.cat([x.probs for x in dists], d), dists[0].event_dims)
