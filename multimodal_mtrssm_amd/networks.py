"""Parameter containers with the reference's constructors and state-dict names.

``Representation`` / ``Transition`` mirror ``models/networks.py:18-173`` (kwargs, ``ValueError`` on a
malformed ``distribution_config``, sub-module names ``rnn_to_post_projector`` / ``rnn_cell`` /
``action_state_projector`` / ``rnn_to_prior_projector`` / ``distribution_factory``); ``MLP`` restates the
``torchrl.modules.MLP`` the YAMLs instantiate (``depth=1``: keys ``0.*``, ``2.*``); ``MTRNN`` mirrors
``mmtrssm/mopoe_mmtrssm/core.py:12-74`` (keys ``_d2h.*``, ``_input2h.*``).

During a rollout none of these modules' ``forward`` runs: the model classes hand their weights to the
HIP scan kernels.  The single-step ``forward`` methods are kept for API users and run the same
kernels with T = 1.
"""

from __future__ import annotations

import importlib

import torch
from torch import Tensor, nn

from multimodal_mtrssm_amd.distributions import MultiOneHotFactory
from multimodal_mtrssm_amd.state import State


def resolve_activation(spec) -> type[nn.Module]:  # noqa: ANN001
    """'ELU' | 'torch.nn.ELU' | nn.ELU -> class."""
    if isinstance(spec, str):
        if "." in spec:
            mod, _, name = spec.rpartition(".")
            return getattr(importlib.import_module(mod), name)
        return getattr(nn, spec)
    return spec


def activation_name(module_or_cls) -> str:  # noqa: ANN001
    cls = module_or_cls if isinstance(module_or_cls, type) else type(module_or_cls)
    return cls.__name__


class MLP(nn.Sequential):
    """``Linear -> act -> ... -> Linear`` with torchrl's constructor names (default activation Tanh)."""

    def __init__(  # noqa: PLR0913
        self,
        in_features: int,
        out_features: int,
        num_cells: int,
        depth: int = 1,
        activation_class: type[nn.Module] | str = nn.Tanh,
        activate_last_layer: bool = False,  # noqa: FBT001, FBT002
    ) -> None:
        act = resolve_activation(activation_class)
        layers: list[nn.Module] = []
        width = in_features
        for _ in range(depth):
            layers += [nn.Linear(width, num_cells), act()]
            width = num_cells
        layers.append(nn.Linear(width, out_features))
        if activate_last_layer:
            layers.append(act())
        super().__init__(*layers)
        self.in_features, self.out_features, self.num_cells, self.depth = in_features, out_features, num_cells, depth
        self.activation_name = activation_name(act)
        self.activate_last_layer = bool(activate_last_layer)

    def forward(self, x: Tensor) -> Tensor:  # noqa: A003
        """On the GPU: the fp32-MFMA GEMMs of ``linear.py`` with each activation fused into the NEXT layer's operand staging
        (and its derivative into the data gradient's epilogue).  CPU tensors (host-side API use) take nn.Sequential's path."""
        from multimodal_mtrssm_amd import _lib  # noqa: PLC0415
        from multimodal_mtrssm_amd.linear import linear  # noqa: PLC0415

        act = _lib.ACT_IDS.get(self.activation_name)
        if not x.is_cuda or act is None or x.dtype != torch.float32:
            return super().forward(x)
        layers = [m for m in self if isinstance(m, nn.Linear)]
        h = linear(x, layers[0].weight, layers[0].bias)
        for lin in layers[1:]:
            h = linear(h, lin.weight, lin.bias, pre_act=act)
        return self[-1](h) if self.activate_last_layer else h

    def two_layer(self) -> tuple[nn.Linear, nn.Linear]:
        """The (first, last) Linear of a depth-1 MLP -- the only shape the scan kernels implement."""
        if self.depth != 1 or len(self) != 3:  # noqa: PLR2004
            msg = "the HIP rollout implements depth=1 MLP heads (Linear-act-Linear), as every reference YAML uses"
            raise NotImplementedError(msg)
        return self[0], self[2]


def _distribution_config(cfg: tuple[int, int] | list[int]) -> tuple[int, int]:
    if isinstance(cfg, list):
        if len(cfg) != 2:  # noqa: PLR2004
            msg = f"distribution_config must have 2 elements, got {len(cfg)}"
            raise ValueError(msg)
        return int(cfg[0]), int(cfg[1])
    class_size, category_size = cfg
    return int(class_size), int(category_size)


class Representation(nn.Module):
    """Posterior head: MLP on ``cat(deter, obs_embed)`` -> K x C logits (``networks.py:18-84``)."""

    def __init__(
        self,
        *,
        deterministic_size: int,
        hidden_size: int,
        obs_embed_size: int,
        distribution_config: tuple[int, int] | list[int],
        activation_name: str = "ReLU",
    ) -> None:
        super().__init__()
        class_size, category_size = _distribution_config(distribution_config)
        self.deterministic_size, self.hidden_size, self.obs_embed_size = deterministic_size, hidden_size, obs_embed_size
        self.rnn_to_post_projector = MLP(
            in_features=obs_embed_size + deterministic_size,
            out_features=class_size * category_size,
            num_cells=hidden_size,
            depth=1,
            activation_class=getattr(torch.nn, activation_name),
        )
        self.distribution_factory = MultiOneHotFactory(class_size=class_size, category_size=category_size)

    def forward(self, obs_embed: Tensor, prior_state: State) -> State:
        """Two plain library GEMMs (``F.linear``) + factory; off the rollout path."""
        logits = self.rnn_to_post_projector(torch.cat([prior_state.deter, obs_embed], -1))
        return State(deter=prior_state.deter, distribution=self.distribution_factory(logits))


class Transition(nn.Module):
    """Prior: MLP -> GRUCell -> MLP (``networks.py:87-173``)."""

    def __init__(
        self,
        *,
        deterministic_size: int,
        hidden_size: int,
        action_size: int,
        distribution_config: tuple[int, int] | list[int],
        activation_name: str,
    ) -> None:
        super().__init__()
        class_size, category_size = _distribution_config(distribution_config)
        self.deterministic_size, self.hidden_size, self.action_size = deterministic_size, hidden_size, action_size
        act = getattr(torch.nn, activation_name)
        self.rnn_cell = nn.GRUCell(input_size=hidden_size, hidden_size=deterministic_size)
        self.action_state_projector = MLP(action_size + class_size * category_size, hidden_size, hidden_size, 1, act)
        self.rnn_to_prior_projector = MLP(deterministic_size, class_size * category_size, hidden_size, 1, act)
        self.distribution_factory = MultiOneHotFactory(class_size=class_size, category_size=category_size)

    def forward(self, action: Tensor, prev_state: State) -> State:
        """One prior step = the prior-only HIP scan with T = 1 (no eager re-implementation)."""
        from multimodal_mtrssm_amd.scan import mrssm_prior_rollout  # noqa: PLC0415

        out = mrssm_prior_rollout(self, action.unsqueeze(1), prev_state.deter, prev_state.stoch, u_prior=None)
        dist = self.distribution_factory(out["prior_logits"][:, 0])
        return State(deter=out["deter"][:, 0], distribution=dist, stoch=out["prior_stoch"][:, 0])


class MTRNN(nn.Module):
    """Leaky-integrator cell ``hidden = (1-1/tau) hidden + (W_d d + W_x x)/tau ; d = tanh(hidden)``.

    Parameter container (``mmtrssm/mopoe_mmtrssm/core.py:12-74``).  The reference keeps ``hidden`` as
    mutable module state; here it lives in ``MTState.hidden_*`` and inside the scan kernel only.
    """

    def __init__(self, input_dim: int, hidden_dim: int, bias: bool = True, tau: float = 2.0) -> None:  # noqa: FBT001, FBT002
        super().__init__()
        assert tau > 1.0, "tau must be greater than 1.0"
        self.hidden_dim, self.input_dim, self.tau = hidden_dim, input_dim, tau
        self._d2h = nn.Linear(hidden_dim, hidden_dim, bias=bias)
        self._input2h = nn.Linear(input_dim, hidden_dim, bias=bias)
        self.hidden: Tensor | None = None  # kept for attribute compatibility; never read on the HIP path
