"""ORACLE (test infrastructure, CPU only) -- restated third-party arithmetic.

This file is NOT part of the product path.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.

The reference calls three packages that are absent from ``/root/reference`` and from this
image:

* ``distribution_extension`` 1.0.7 @ git e150621 (``uv.lock:744-746``) --
  ``MultiOneHotFactory``, ``Distribution``, ``kl_divergence``,
  ``utils.stack_distribution`` / ``utils.cat_distribution``.
  Call sites: ``networks.py:65,83,146,172``; ``state.py:17,134,151``; ``core.py:134,212-216``;
  ``mmtrssm/mopoe_mmtrssm/core.py:286,312,317,455-456,464,589-600``.
* ``torchrl.modules.MLP`` 0.10.1 (``uv.lock:3347-3348``) -- ``networks.py:57-64,130-145``.

Their arithmetic is *build-defined* here (published algorithm: DreamerV2-style multi-categorical
latent with straight-through one-hot samples and KL balancing).  PARITY UNPINNED at this
boundary: the reference holds no test, fixture or golden vector for it.

Definitions
-----------
``MultiOneHotFactory(class_size=C, category_size=K)`` maps flat logits ``[*, K*C]`` to K independent
categoricals with C classes each: reshape ``[*, K, C]``, softmax over the last axis.

``rsample`` draws a one-hot per categorical by inverse CDF from *injected* uniforms
``u[*, K]`` (index = number of c in [0, C-2] whose inclusive cumulative probability is <= u) and
returns ``onehot + (probs - probs.detach())`` flattened to ``[*, K*C]`` (straight-through).

``kl_divergence(q, p, use_balancing)`` = sum over categoricals of sum_c q (log q - log p), mean
over batch dims; with balancing 0.8 * KL(sg q || p) + 0.2 * KL(q || sg p).
"""

from __future__ import annotations

from collections import deque
from typing import Iterable

import torch
from torch import Tensor, nn

KL_BALANCE_ALPHA = 0.8


class NoiseTape:
    """FIFO of uniform tensors consumed by ``MultiOneHot.rsample`` in call order.

    The reference draws with ``torch.multinomial`` streams inside the absent package; bit parity
    with those streams is a non-goal (SURVEY.md section 7 "RNG contract").  The tape makes every draw an
    explicit input so the reference control flow, the restatement and the HIP kernels can be
    run on identical noise.
    """

    def __init__(self) -> None:
        self._q: deque[Tensor] = deque()
        self.draws = 0

    def push(self, u: Tensor) -> None:
        self._q.append(u)

    def extend(self, us: Iterable[Tensor]) -> None:
        for u in us:
            self.push(u)

    def pop(self, shape: torch.Size, like: Tensor) -> Tensor:
        self.draws += 1
        if not self._q:
            return torch.rand(shape, dtype=like.dtype, device=like.device)
        u = self._q.popleft()
        if tuple(u.shape) != tuple(shape):
            msg = f"noise tape shape {tuple(u.shape)} != draw shape {tuple(shape)} (draw #{self.draws})"
            raise RuntimeError(msg)
        return u.to(like.device, like.dtype)

    def __len__(self) -> int:
        return len(self._q)

    def clear(self) -> None:
        self._q.clear()
        self.draws = 0


TAPE = NoiseTape()


def inverse_cdf_index(probs: Tensor, u: Tensor) -> Tensor:
    """Index of the sampled class.  probs ``[*, K, C]``, u ``[*, K]`` -> int64 ``[*, K]``.

    Sequential fp32 inclusive cumulative sum in class order; the last class absorbs rounding.
    """
    num_classes = probs.shape[-1]
    acc = torch.zeros_like(probs[..., 0])
    idx = torch.zeros(probs.shape[:-1], dtype=torch.int64, device=probs.device)
    for c in range(num_classes - 1):
        acc = acc + probs[..., c]
        idx = idx + (acc <= u).to(torch.int64)
    return idx


def sampling_margin(probs: Tensor, u: Tensor) -> Tensor:
    """min_c |u - cdf_c| per categorical (fixtures are screened on this)."""
    cdf = torch.cumsum(probs, dim=-1)[..., :-1]
    return (cdf - u.unsqueeze(-1)).abs().amin(dim=-1)


class MultiOneHot:
    """K independent C-way categoricals over the trailing ``[K, C]`` axes of ``probs``."""

    def __init__(self, logits: Tensor, *, reinterpreted: int = 0) -> None:
        # logits: [*, K, C] normalised log-probabilities
        self.logits = logits
        self.probs = torch.softmax(logits, dim=-1)
        self.reinterpreted = reinterpreted

    # -- construction helpers ------------------------------------------------------------
    @classmethod
    def _wrap(cls, logits: Tensor, probs: Tensor, reinterpreted: int) -> "MultiOneHot":
        obj = cls.__new__(cls)
        obj.logits = logits
        obj.probs = probs
        obj.reinterpreted = reinterpreted
        return obj

    def _map(self, fn) -> "MultiOneHot":  # noqa: ANN001
        return self._wrap(fn(self.logits), fn(self.probs), self.reinterpreted)

    # -- API used by the reference -------------------------------------------------------
    @property
    def batch_shape(self) -> torch.Size:
        return self.probs.shape[:-2]

    def rsample(self) -> Tensor:
        u = TAPE.pop(self.probs.shape[:-1], self.probs)
        idx = inverse_cdf_index(self.probs.detach(), u)
        onehot = torch.nn.functional.one_hot(idx, self.probs.shape[-1]).to(self.probs.dtype)
        sample = onehot + (self.probs - self.probs.detach())
        return sample.flatten(start_dim=-2)

    def independent(self, ndims: int) -> "MultiOneHot":
        return self._wrap(self.logits, self.probs, ndims)

    def __getitem__(self, loc) -> "MultiOneHot":  # noqa: ANN001
        # loc indexes batch dims only; the trailing [K, C] axes are kept whole
        return self._map(lambda x: x[loc])

    def to(self, device) -> "MultiOneHot":  # noqa: ANN001
        return self._map(lambda x: x.to(device))

    def detach(self) -> "MultiOneHot":
        return self._map(lambda x: x.detach())

    def clone(self) -> "MultiOneHot":
        return self._map(lambda x: x.clone())

    def _batch_dim(self, dim: int) -> int:
        nb = self.probs.dim() - 2
        return dim if dim >= 0 else dim + nb + 1

    def squeeze(self, dim: int) -> "MultiOneHot":
        d = dim if dim >= 0 else dim + self.probs.dim() - 2
        return self._map(lambda x: x.squeeze(d))

    def unsqueeze(self, dim: int) -> "MultiOneHot":
        d = self._batch_dim(dim)
        return self._map(lambda x: x.unsqueeze(d))


Distribution = MultiOneHot


class MultiOneHotFactory(nn.Module):
    """``[*, K*C]`` logits -> ``MultiOneHot`` (K = category_size categoricals, C = class_size classes)."""

    def __init__(self, class_size: int, category_size: int) -> None:
        super().__init__()
        self.class_size = class_size
        self.category_size = category_size

    def forward(self, logits: Tensor) -> MultiOneHot:
        shaped = logits.reshape(*logits.shape[:-1], self.category_size, self.class_size)
        log_probs = torch.log_softmax(shaped, dim=-1)
        return MultiOneHot._wrap(log_probs, torch.softmax(shaped, dim=-1), 0)


def _categorical_kl(q: MultiOneHot, p: MultiOneHot) -> Tensor:
    per_cat = (q.probs * (q.logits - p.logits)).sum(dim=-1)  # [*, K]
    return per_cat.sum(dim=-1)  # independent(1): sum over the K categoricals


def kl_divergence(q: MultiOneHot, p: MultiOneHot, use_balancing: bool = False) -> Tensor:  # noqa: FBT001, FBT002
    if use_balancing:
        lhs = _categorical_kl(q.detach(), p).mean()
        rhs = _categorical_kl(q, p.detach()).mean()
        return KL_BALANCE_ALPHA * lhs + (1.0 - KL_BALANCE_ALPHA) * rhs
    return _categorical_kl(q, p).mean()


def stack_distribution(dists: list[MultiOneHot], dim: int) -> MultiOneHot:
    d = dists[0]._batch_dim(dim)
    return MultiOneHot._wrap(
        torch.stack([x.logits for x in dists], dim=d),
        torch.stack([x.probs for x in dists], dim=d),
        dists[0].reinterpreted,
    )


def cat_distribution(dists: list[MultiOneHot], dim: int) -> MultiOneHot:
    d = dim if dim >= 0 else dim + dists[0].probs.dim() - 2
    return MultiOneHot._wrap(
        torch.cat([x.logits for x in dists], dim=d),
        torch.cat([x.probs for x in dists], dim=d),
        dists[0].reinterpreted,
    )


class MLP(nn.Sequential):
    """depth-1 MLP = Linear -> act -> Linear (state-dict keys ``0.*``, ``2.*``); default act Tanh.

    Restates ``torchrl.modules.MLP(in_features, out_features, num_cells, depth,
    activation_class, activate_last_layer=False)`` for the only shape the reference uses
    (``depth=1``; ``networks.py:57-64,130-145``; yaml ``init_proj`` / ``l_prior`` ...).
    """

    def __init__(  # noqa: PLR0913
        self,
        in_features: int,
        out_features: int,
        num_cells: int,
        depth: int = 1,
        activation_class: type[nn.Module] | str = nn.Tanh,
        activate_last_layer: bool = False,  # noqa: FBT001, FBT002
    ) -> None:
        if isinstance(activation_class, str):
            activation_class = getattr(nn, activation_class.rsplit(".", 1)[-1])
        layers: list[nn.Module] = []
        width = in_features
        for _ in range(depth):
            layers += [nn.Linear(width, num_cells), activation_class()]
            width = num_cells
        layers.append(nn.Linear(width, out_features))
        if activate_last_layer:
            layers.append(activation_class())
        super().__init__(*layers)
        self.in_features = in_features
        self.out_features = out_features
