"""``State`` / ``MTState`` value objects with the reference's surface.

Mirrors ``models/state.py:11-152`` and ``models/mmtrssm/state.py:11-248`` (fields, sampling on
construction, ``feature`` layout, indexing / to / detach / clone / squeeze / unsqueeze, and the
stack / cat helpers used by the callbacks, ``mrssm/callback.py:156-189``).  The rollout kernels
write ``[B, T, .]`` tensors directly, so the per-step ``stack_states`` of the reference never runs
on the hot path; these helpers exist for API users.

One deliberate difference: ``MTState.clone`` clones ``distribution_h`` from ``distribution_h``
(the reference clones ``distribution_l`` into it, ``mmtrssm/state.py:133`` -- a bug).
"""

from __future__ import annotations

from collections.abc import Iterator

import torch
from torch import Tensor

from multimodal_mtrssm_amd.distributions import Distribution, cat_distribution, stack_distribution


class State:
    """Latent state: ``deter [*, D]``, ``distribution`` over ``[*, K, C]``, ``stoch [*, K*C]``."""

    def __init__(self, deter: Tensor, distribution: Distribution, stoch: Tensor | None = None) -> None:
        self.deter = deter
        self.distribution = distribution
        self.stoch = distribution.rsample() if stoch is None else stoch
        self.feature = torch.cat([self.deter, self.stoch], dim=-1)

    def _map(self, fn, dist_fn) -> "State":  # noqa: ANN001
        return type(self)(deter=fn(self.deter), stoch=fn(self.stoch), distribution=dist_fn(self.distribution))

    def __iter__(self) -> Iterator["State"]:
        return (self[i] for i in range(self.deter.shape[0]))

    def __getitem__(self, loc) -> "State":  # noqa: ANN001
        return self._map(lambda x: x[loc], lambda d: d[loc])

    def to(self, device) -> "State":  # noqa: ANN001
        return self._map(lambda x: x.to(device), lambda d: d.to(device))

    def detach(self) -> "State":
        return self._map(lambda x: x.detach(), lambda d: d.detach())

    def clone(self) -> "State":
        return self._map(lambda x: x.clone(), lambda d: d.clone())

    def squeeze(self, dim: int) -> "State":
        return self._map(lambda x: x.squeeze(dim), lambda d: d.squeeze(dim))

    def unsqueeze(self, dim: int) -> "State":
        return self._map(lambda x: x.unsqueeze(dim), lambda d: d.unsqueeze(dim))


def stack_states(states: list[State], dim: int) -> State:
    return State(
        deter=torch.stack([s.deter for s in states], dim=dim),
        stoch=torch.stack([s.stoch for s in states], dim=dim),
        distribution=stack_distribution([s.distribution for s in states], dim),
    )


def cat_states(states: list[State], dim: int) -> State:
    return State(
        deter=torch.cat([s.deter for s in states], dim=dim),
        stoch=torch.cat([s.stoch for s in states], dim=dim),
        distribution=cat_distribution([s.distribution for s in states], dim),
    )


class MTState:
    """Two-level state; ``feature = cat(deter_h, stoch_h, deter_l, stoch_l)`` (``mmtrssm/state.py:51``)."""

    _TENSORS = ("deter_h", "deter_l", "stoch_h", "stoch_l")

    def __init__(  # noqa: PLR0913
        self,
        deter_h: Tensor,
        deter_l: Tensor,
        distribution_h: Distribution,
        distribution_l: Distribution,
        hidden_h: Tensor,
        hidden_l: Tensor,
        stoch_h: Tensor | None = None,
        stoch_l: Tensor | None = None,
    ) -> None:
        self.deter_h = deter_h
        self.deter_l = deter_l
        self.distribution_h = distribution_h
        self.distribution_l = distribution_l
        self.hidden_h = hidden_h
        self.hidden_l = hidden_l
        # draw order h then l, as the reference (mmtrssm/state.py:48-49)
        self.stoch_h = distribution_h.rsample() if stoch_h is None else stoch_h
        self.stoch_l = distribution_l.rsample() if stoch_l is None else stoch_l
        self.feature = torch.cat([self.deter_h, self.stoch_h, self.deter_l, self.stoch_l], dim=-1)

    def _map(self, fn, dist_fn, hidden_fn=None) -> "MTState":  # noqa: ANN001
        hidden_fn = hidden_fn or fn
        return type(self)(
            deter_h=fn(self.deter_h), deter_l=fn(self.deter_l),
            distribution_h=dist_fn(self.distribution_h), distribution_l=dist_fn(self.distribution_l),
            hidden_h=hidden_fn(self.hidden_h), hidden_l=hidden_fn(self.hidden_l),
            stoch_h=fn(self.stoch_h), stoch_l=fn(self.stoch_l),
        )

    def __iter__(self) -> Iterator["MTState"]:
        return (self[i] for i in range(self.deter_h.shape[0]))

    @staticmethod
    def _if_batched(fn):  # noqa: ANN001, ANN205
        # hidden_* is only indexed / reshaped when it carries batch dims (mmtrssm/state.py:78-79)
        return lambda h: fn(h) if h.dim() > 1 else h

    def __getitem__(self, loc) -> "MTState":  # noqa: ANN001
        return self._map(lambda x: x[loc], lambda d: d[loc], self._if_batched(lambda x: x[loc]))

    def to(self, device) -> "MTState":  # noqa: ANN001
        return self._map(lambda x: x.to(device), lambda d: d.to(device))

    def detach(self) -> "MTState":
        return self._map(lambda x: x.detach(), lambda d: d.detach())

    def clone(self) -> "MTState":
        return self._map(lambda x: x.clone(), lambda d: d.clone())

    def squeeze(self, dim: int) -> "MTState":
        return self._map(lambda x: x.squeeze(dim), lambda d: d.squeeze(dim), self._if_batched(lambda x: x.squeeze(dim)))

    def unsqueeze(self, dim: int) -> "MTState":
        return self._map(lambda x: x.unsqueeze(dim), lambda d: d.unsqueeze(dim), self._if_batched(lambda x: x.unsqueeze(dim)))


def stack_mtstates(states: list[MTState], dim: int) -> MTState:
    def stk(name: str) -> Tensor:
        return torch.stack([getattr(s, name) for s in states], dim=dim)

    first = states[0]
    return MTState(
        deter_h=stk("deter_h"), deter_l=stk("deter_l"),
        distribution_h=stack_distribution([s.distribution_h for s in states], dim),
        distribution_l=stack_distribution([s.distribution_l for s in states], dim),
        hidden_h=stk("hidden_h") if first.hidden_h.dim() > 1 else first.hidden_h,
        hidden_l=stk("hidden_l") if first.hidden_l.dim() > 1 else first.hidden_l,
        stoch_h=stk("stoch_h"), stoch_l=stk("stoch_l"),
    )


def cat_mtstates(states: list[MTState], dim: int) -> MTState:
    def cat(name: str) -> Tensor:
        return torch.cat([getattr(s, name) for s in states], dim=dim)

    return MTState(
        deter_h=cat("deter_h"), deter_l=cat("deter_l"),
        distribution_h=cat_distribution([s.distribution_h for s in states], dim),
        distribution_l=cat_distribution([s.distribution_l for s in states], dim),
        hidden_h=states[-1].hidden_h, hidden_l=states[-1].hidden_l,  # keeps the last hidden (mmtrssm/state.py:237-238)
        stoch_h=cat("stoch_h"), stoch_l=cat("stoch_l"),
    )
