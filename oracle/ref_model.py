"""ORACLE (test infrastructure, CPU only) -- op-for-op restatement of the reference hot path.

NOT part of the product path: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module (see ``oracle/README.md``).

Restates, in eager PyTorch on plain tensors (no State objects inside the loop), the per-timestep
arithmetic of

* ``models/core.py:121-135`` (``initial_state``), ``:170-185`` (``rollout_transition``),
  ``:187-221`` (``shared_step``);
* ``models/networks.py:70-84`` (posterior head), ``:151-173`` (Transition: MLP -> GRUCell -> MLP);
* ``models/mrssm/mopoe_mrssm/core.py:112-163`` (MoE fusion), ``:184-260`` (the T loop),
  ``:262-308`` (decode + Gaussian NLL);
* ``models/mmtrssm/mopoe_mmtrssm/core.py:12-74`` (MTRNN), ``:263-319`` (lower/higher priors and
  posteriors), ``:321-362`` (initial MTState), ``:364-494`` (the T loop), ``:496-544``
  (prior-only loop), ``:563-606`` (``shared_step`` with ``kl`` + ``kl_h``);
* ``models/objective.py:7-23`` (``likelihood``).

It is pinned against the reference's *own* control flow by ``oracle/gen_golden.py`` (run in the
build container only), which executes the reference files from ``/root/reference/src`` on the same
weights and injected noise and asserts equality; the results are frozen in ``tests/golden/*.npz``.
The third-party arithmetic underneath (``oracle/ref_dists.py``, ``oracle/ref_cnn.py``) is
build-defined: parity unpinned there.

State-dict names equal the reference's so weights move by name.

Noise contract (all uniforms in [0,1), one per categorical):
  MRSSM : ``u_init[B,K]``, ``u_prior[B,T,K]``, ``u_post[B,T,K]``
  MMTRSSM: ``u_init_h[B,Kh]``, ``u_init_l[B,Kl]``, ``u_post_l[B,T,Kl]``, ``u_post_h[B,T,Kh]``,
           ``u_prior_h[B,T,Kh]``, ``u_prior_l[B,T,Kl]``
The reference additionally draws two samples per step that it throws away
(``mrssm/mopoe_mrssm/core.py:83`` via ``:226-238``); ``wasteful=True`` reproduces those draws (cost
model for the CPU baseline), results do not depend on them.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Any

import torch
import torch.nn.functional as F  # noqa: N812
from torch import Tensor, nn

from oracle.ref_cnn import Decoder, Encoder
from oracle.ref_dists import KL_BALANCE_ALPHA, MLP, inverse_cdf_index

LOG_THIRD = math.log(1.0 / 3.0)


# ----------------------------------------------------------------------------------------
# categorical helpers on plain tensors
# ----------------------------------------------------------------------------------------
def cat_probs(logits: Tensor, cats: int, classes: int) -> tuple[Tensor, Tensor]:
    """flat ``[*, K*C]`` -> (log-probs, probs) each ``[*, K, C]`` (softmax per categorical)."""
    shaped = logits.reshape(*logits.shape[:-1], cats, classes)
    return torch.log_softmax(shaped, dim=-1), torch.softmax(shaped, dim=-1)


def st_sample(probs: Tensor, u: Tensor) -> Tensor:
    """Straight-through one-hot sample, flat ``[*, K*C]``."""
    idx = inverse_cdf_index(probs.detach(), u)
    onehot = F.one_hot(idx, probs.shape[-1]).to(probs.dtype)
    return (onehot + (probs - probs.detach())).flatten(start_dim=-2)


def mopoe_mix(audio_logits: Tensor, vision_logits: Tensor) -> Tensor:
    """``mrssm/mopoe_mrssm/core.py:241-243`` + ``:112-163``: flat log-softmax per expert, PoE =
    sum of log-probs (left un-normalised), MoE = logsumexp over {A, V, A+V} with weight 1/3."""
    ls_a = torch.log_softmax(audio_logits, dim=-1)
    ls_v = torch.log_softmax(vision_logits, dim=-1)
    fused = ls_a + ls_v
    stacked = torch.stack([LOG_THIRD + ls_a, LOG_THIRD + ls_v, LOG_THIRD + fused], dim=-2)
    return torch.logsumexp(stacked, dim=-2)


def kl_cat(q_logp: Tensor, q_p: Tensor, p_logp: Tensor) -> Tensor:
    """sum_K sum_C q (log q - log p) -> ``[*]``."""
    return (q_p * (q_logp - p_logp)).sum(dim=(-1, -2))


def kl_loss(q_logits: Tensor, p_logits: Tensor, cats: int, classes: int, balancing: bool) -> Tensor:  # noqa: FBT001
    ql, qp = cat_probs(q_logits, cats, classes)
    pl, _ = cat_probs(p_logits, cats, classes)
    if balancing:
        lhs = kl_cat(ql.detach(), qp.detach(), pl).mean()
        rhs = kl_cat(ql, qp, pl.detach()).mean()
        return KL_BALANCE_ALPHA * lhs + (1.0 - KL_BALANCE_ALPHA) * rhs
    return kl_cat(ql, qp, pl).mean()


def gaussian_nll(prediction: Tensor, target: Tensor, event_ndims: int = 3) -> Tensor:
    """``objective.py:7-23``: -mean over batch dims of sum over event dims of log N(target; pred, 1)."""
    dims = tuple(range(-event_ndims, 0))
    logp = -0.5 * (target - prediction) ** 2 - 0.5 * math.log(2.0 * math.pi)
    return -logp.sum(dim=dims).mean()


# ----------------------------------------------------------------------------------------
# shared sub-networks (names follow the reference state-dict)
# ----------------------------------------------------------------------------------------
class _Transition(nn.Module):
    def __init__(self, deter: int, hidden: int, action: int, stoch: int, act: str) -> None:
        super().__init__()
        self.rnn_cell = nn.GRUCell(input_size=hidden, hidden_size=deter)
        self.action_state_projector = MLP(action + stoch, hidden, hidden, 1, getattr(nn, act))
        self.rnn_to_prior_projector = MLP(deter, stoch, hidden, 1, getattr(nn, act))


class _Representation(nn.Module):
    def __init__(self, deter: int, hidden: int, embed: int, stoch: int, act: str) -> None:
        super().__init__()
        self.rnn_to_post_projector = MLP(embed + deter, stoch, hidden, 1, getattr(nn, act))


class _MTRNN(nn.Module):
    """``mmtrssm/mopoe_mmtrssm/core.py:12-74`` with the hidden state made an explicit argument."""

    def __init__(self, input_dim: int, hidden_dim: int, tau: float) -> None:
        super().__init__()
        assert tau > 1.0, "tau must be greater than 1.0"
        self.tau = tau
        self._d2h = nn.Linear(hidden_dim, hidden_dim)
        self._input2h = nn.Linear(input_dim, hidden_dim)

    def forward(self, x: Tensor, prev_d: Tensor, hidden: Tensor) -> tuple[Tensor, Tensor]:
        hidden = (1 - 1 / self.tau) * hidden + (self._d2h(prev_d) + self._input2h(x)) / self.tau
        return torch.tanh(hidden), hidden


@dataclass
class MRSSMDims:
    deter: int = 32
    hidden: int = 32
    classes: int = 4  # class_size  (softmax axis)
    cats: int = 4  # category_size (number of categoricals)
    action: int = 6
    embed: int = 64
    activation: str = "ELU"
    init_cells: int = 200
    kl_coeff: float = 1.0
    use_kl_balancing: bool = True
    enc_audio: dict[str, Any] = field(default_factory=dict)
    enc_vision: dict[str, Any] = field(default_factory=dict)
    dec_audio: dict[str, Any] = field(default_factory=dict)
    dec_vision: dict[str, Any] = field(default_factory=dict)

    @property
    def stoch(self) -> int:
        return self.classes * self.cats


class OracleMRSSM(nn.Module):
    """Eager CPU restatement of ``MoPoE_MRSSM`` (``mrssm/mopoe_mrssm/core.py``)."""

    def __init__(self, dims: MRSSMDims) -> None:
        super().__init__()
        d = dims
        self.dims = d
        # registration order follows the reference (core.py:27-29 then mrssm core.py:55-60); the audio head is
        # registered twice upstream (``representation`` and ``audio_representation`` are one module)
        self.representation = _Representation(d.deter, d.hidden, d.embed, d.stoch, d.activation)
        self.transition = _Transition(d.deter, d.hidden, d.action, d.stoch, d.activation)
        self.init_proj = MLP(d.embed, d.deter, d.init_cells, 1, nn.Tanh)
        self.audio_representation = self.representation
        self.vision_representation = _Representation(d.deter, d.hidden, d.embed, d.stoch, d.activation)
        self.audio_encoder = Encoder(d.enc_audio)
        self.vision_encoder = Encoder(d.enc_vision)
        self.audio_decoder = Decoder(d.dec_audio)
        self.vision_decoder = Decoder(d.dec_vision)

    # -- pieces ---------------------------------------------------------------------------
    def initial_state(self, audio0: Tensor, vision0: Tensor, u_init: Tensor) -> dict[str, Tensor]:
        d = self.dims
        embed = (self.audio_encoder(audio0) + self.vision_encoder(vision0)) / 2.0
        deter = self.init_proj(embed)
        logits = self.transition.rnn_to_prior_projector(deter)
        _, probs = cat_probs(logits, d.cats, d.classes)
        return {"deter": deter, "logits": logits, "stoch": st_sample(probs, u_init)}

    def _prior_step(self, action: Tensor, deter: Tensor, stoch: Tensor, u: Tensor) -> tuple[Tensor, Tensor, Tensor]:
        d = self.dims
        tr = self.transition
        x = tr.action_state_projector(torch.cat([action, stoch], dim=-1))
        deter = tr.rnn_cell(x, deter)
        logits = tr.rnn_to_prior_projector(deter)
        _, probs = cat_probs(logits, d.cats, d.classes)
        return deter, logits, st_sample(probs, u)

    def rollout_representation(  # noqa: PLR0913
        self,
        actions: Tensor,
        audio_embed: Tensor,
        vision_embed: Tensor,
        state0: dict[str, Tensor],
        u_prior: Tensor,
        u_post: Tensor,
        *,
        wasteful: bool = False,
    ) -> dict[str, Tensor]:
        d = self.dims
        deter, stoch = state0["deter"], state0["stoch"]
        keep: dict[str, list[Tensor]] = {
            k: [] for k in ("deter", "prior_logits", "prior_stoch", "audio_logits", "vision_logits", "post_logits", "post_stoch")
        }
        for t in range(actions.shape[1]):
            deter, prior_logits, prior_stoch = self._prior_step(actions[:, t], deter, stoch, u_prior[:, t])
            a_logits = self.audio_representation.rnn_to_post_projector(torch.cat([deter, audio_embed[:, t]], -1))
            v_logits = self.vision_representation.rnn_to_post_projector(torch.cat([deter, vision_embed[:, t]], -1))
            if wasteful:  # the two State() constructions the reference discards (core.py:83)
                for lg in (a_logits, v_logits):
                    _, pr = cat_probs(lg, d.cats, d.classes)
                    st_sample(pr, torch.rand(pr.shape[:-1]))
                    torch.cat([deter, lg], dim=-1)
            mixed = mopoe_mix(a_logits, v_logits)
            _, post_probs = cat_probs(mixed, d.cats, d.classes)
            stoch = st_sample(post_probs, u_post[:, t])
            for k, v in (
                ("deter", deter), ("prior_logits", prior_logits), ("prior_stoch", prior_stoch),
                ("audio_logits", a_logits), ("vision_logits", v_logits), ("post_logits", mixed), ("post_stoch", stoch),
            ):
                keep[k].append(v)
        return {k: torch.stack(v, dim=1) for k, v in keep.items()}

    def rollout_transition(self, actions: Tensor, state0: dict[str, Tensor], u_prior: Tensor) -> dict[str, Tensor]:
        deter, stoch = state0["deter"], state0["stoch"]
        keep: dict[str, list[Tensor]] = {"deter": [], "prior_logits": [], "prior_stoch": []}
        for t in range(actions.shape[1]):
            deter, logits, stoch = self._prior_step(actions[:, t], deter, stoch, u_prior[:, t])
            keep["deter"].append(deter)
            keep["prior_logits"].append(logits)
            keep["prior_stoch"].append(stoch)
        return {k: torch.stack(v, dim=1) for k, v in keep.items()}

    def shared_step(self, batch: tuple[Tensor, ...], noise: dict[str, Tensor], *, wasteful: bool = False) -> dict[str, Tensor]:
        d = self.dims
        act_in, audio_in, vision_in, _, audio_tgt, vision_tgt = batch
        state0 = self.initial_state(audio_in[:, 0], vision_in[:, 0], noise["u_init"])
        audio_embed = self.audio_encoder(audio_in)
        vision_embed = self.vision_encoder(vision_in)
        roll = self.rollout_representation(
            act_in, audio_embed, vision_embed, state0, noise["u_prior"], noise["u_post"], wasteful=wasteful
        )
        feature = torch.cat([roll["deter"], roll["post_stoch"]], dim=-1)
        recon_a = self.audio_decoder(feature)
        recon_v = self.vision_decoder(feature)
        nll_a = gaussian_nll(recon_a, audio_tgt)
        nll_v = gaussian_nll(recon_v, vision_tgt)
        kl = kl_loss(roll["post_logits"], roll["prior_logits"], d.cats, d.classes, d.use_kl_balancing) * d.kl_coeff
        out = {
            "loss": nll_a + nll_v + kl,
            "recon": nll_a + nll_v,
            "recon/audio": nll_a,
            "recon/vision": nll_v,
            "kl": kl,
        }
        out.update({f"_{k}": v for k, v in roll.items()})
        out["_deter0"], out["_stoch0"], out["_logits0"] = state0["deter"], state0["stoch"], state0["logits"]
        out["_audio_embed"], out["_vision_embed"] = audio_embed, vision_embed
        return out


# ----------------------------------------------------------------------------------------
# two-timescale variant
# ----------------------------------------------------------------------------------------
@dataclass
class MMTRSSMDims:
    hd: int = 32
    hs_classes: int = 2
    hs_cats: int = 8
    ld: int = 32
    ls_classes: int = 4
    ls_cats: int = 4
    hidden: int = 32  # num_cells of l_prior / h_prior / h_posterior and of the two posterior heads
    action: int = 6
    embed: int = 64
    l_tau: float = 2.0
    h_tau: float = 4.0
    activation: str = "ELU"
    init_cells: int = 200
    kl_coeff: float = 1.0
    w_kl_h: float = 1.0
    use_kl_balancing: bool = True
    enc_audio: dict[str, Any] = field(default_factory=dict)
    enc_vision: dict[str, Any] = field(default_factory=dict)
    dec_audio: dict[str, Any] = field(default_factory=dict)
    dec_vision: dict[str, Any] = field(default_factory=dict)

    @property
    def hs(self) -> int:
        return self.hs_classes * self.hs_cats

    @property
    def ls(self) -> int:
        return self.ls_classes * self.ls_cats

    @property
    def feature(self) -> int:
        return self.hd + self.hs + self.ld + self.ls


class OracleMMTRSSM(nn.Module):
    """Eager CPU restatement of ``MoPoE_MMTRSSM`` (``mmtrssm/mopoe_mmtrssm/core.py``)."""

    def __init__(self, dims: MMTRSSMDims) -> None:
        super().__init__()
        d = dims
        self.dims = d
        act = getattr(nn, d.activation)
        # parameters the reference registers but never trains on this path (SURVEY.md section 2 "Hazard")
        self.representation = _Representation(d.ld, d.hidden, d.embed, d.ls, d.activation)
        self.transition = _Transition(d.ld, d.ld, 1, 1, "ELU")
        self.init_proj = MLP(d.embed, d.hd + d.ld, d.init_cells, 1, nn.Tanh)
        self.audio_representation = self.representation
        self.vision_representation = _Representation(d.ld, d.hidden, d.embed, d.ls, d.activation)
        self.audio_encoder = Encoder(d.enc_audio)
        self.vision_encoder = Encoder(d.enc_vision)
        self.audio_decoder = Decoder(d.dec_audio)
        self.vision_decoder = Decoder(d.dec_vision)
        self.l_rnn = _MTRNN(d.action + d.ls + d.hs, d.ld, d.l_tau)
        self.h_rnn = _MTRNN(d.hs, d.hd, d.h_tau)
        self.l_prior = MLP(d.ld, d.ls, d.hidden, 1, act)
        self.l_posterior = MLP(d.ld + d.embed, d.ls, d.hidden, 1, act)  # registered, never called upstream (core.py:188)
        self.h_prior = MLP(d.hd, d.hs, d.hidden, 1, act)
        self.h_posterior = MLP(d.ld + d.hd, d.hs, d.hidden, 1, act)

    def initial_state(self, audio0: Tensor, vision0: Tensor, u_init_h: Tensor, u_init_l: Tensor) -> dict[str, Tensor]:
        d = self.dims
        embed = (self.audio_encoder(audio0) + self.vision_encoder(vision0)) / 2.0
        h = self.init_proj(embed)
        higher, lower = h[..., : d.hd], h[..., d.hd :]
        h_logits = self.h_prior(higher)
        l_logits = self.l_prior(lower)
        _, hp = cat_probs(h_logits, d.hs_cats, d.hs_classes)
        _, lp = cat_probs(l_logits, d.ls_cats, d.ls_classes)
        return {
            "deter_h": higher, "deter_l": lower, "hidden_h": higher, "hidden_l": lower,
            "logits_h": h_logits, "logits_l": l_logits,
            "stoch_h": st_sample(hp, u_init_h), "stoch_l": st_sample(lp, u_init_l),
        }

    def rollout_representation(  # noqa: PLR0913, PLR0914
        self,
        actions: Tensor,
        audio_embed: Tensor,
        vision_embed: Tensor,
        state0: dict[str, Tensor],
        noise: dict[str, Tensor],
    ) -> dict[str, Tensor]:
        d = self.dims
        deter_l, deter_h = state0["deter_l"], state0["deter_h"]
        hidden_l, hidden_h = state0["hidden_l"], state0["hidden_h"]
        stoch_l, stoch_h = state0["stoch_l"], state0["stoch_h"]
        names = (
            "deter_l", "deter_h", "hidden_l", "hidden_h", "prior_logits_l", "prior_logits_h", "prior_stoch_l",
            "prior_stoch_h", "audio_logits", "vision_logits", "post_logits_l", "post_logits_h", "post_stoch_l", "post_stoch_h",
        )
        keep: dict[str, list[Tensor]] = {k: [] for k in names}
        for t in range(actions.shape[1]):
            l_in = torch.cat([actions[:, t], stoch_l, stoch_h], dim=-1)
            deter_l, hidden_l = self.l_rnn(l_in, deter_l, hidden_l)
            prior_logits_l = self.l_prior(deter_l)
            a_logits = self.audio_representation.rnn_to_post_projector(torch.cat([deter_l, audio_embed[:, t]], -1))
            v_logits = self.vision_representation.rnn_to_post_projector(torch.cat([deter_l, vision_embed[:, t]], -1))
            post_logits_l = mopoe_mix(a_logits, v_logits)
            _, ql = cat_probs(post_logits_l, d.ls_cats, d.ls_classes)
            new_stoch_l = st_sample(ql, noise["u_post_l"][:, t])

            deter_h, hidden_h = self.h_rnn(stoch_h, deter_h, hidden_h)
            prior_logits_h = self.h_prior(deter_h)
            post_logits_h = self.h_posterior(torch.cat([deter_l, deter_h], dim=-1))
            _, qh = cat_probs(post_logits_h, d.hs_cats, d.hs_classes)
            new_stoch_h = st_sample(qh, noise["u_post_h"][:, t])

            _, ph = cat_probs(prior_logits_h, d.hs_cats, d.hs_classes)
            _, pl = cat_probs(prior_logits_l, d.ls_cats, d.ls_classes)
            prior_stoch_h = st_sample(ph, noise["u_prior_h"][:, t])
            prior_stoch_l = st_sample(pl, noise["u_prior_l"][:, t])
            stoch_l, stoch_h = new_stoch_l, new_stoch_h
            vals = (
                deter_l, deter_h, hidden_l, hidden_h, prior_logits_l, prior_logits_h, prior_stoch_l, prior_stoch_h,
                a_logits, v_logits, post_logits_l, post_logits_h, stoch_l, stoch_h,
            )
            for k, v in zip(names, vals, strict=True):
                keep[k].append(v)
        return {k: torch.stack(v, dim=1) for k, v in keep.items()}

    def rollout_transition(self, actions: Tensor, state0: dict[str, Tensor], noise: dict[str, Tensor]) -> dict[str, Tensor]:
        d = self.dims
        deter_l, deter_h = state0["deter_l"], state0["deter_h"]
        hidden_l, hidden_h = state0["hidden_l"], state0["hidden_h"]
        stoch_l, stoch_h = state0["stoch_l"], state0["stoch_h"]
        names = ("deter_l", "deter_h", "hidden_l", "hidden_h", "prior_logits_l", "prior_logits_h", "prior_stoch_l", "prior_stoch_h")
        keep: dict[str, list[Tensor]] = {k: [] for k in names}
        for t in range(actions.shape[1]):
            l_in = torch.cat([actions[:, t], stoch_l, stoch_h], dim=-1)
            deter_l, hidden_l = self.l_rnn(l_in, deter_l, hidden_l)
            logits_l = self.l_prior(deter_l)
            deter_h, hidden_h = self.h_rnn(stoch_h, deter_h, hidden_h)
            logits_h = self.h_prior(deter_h)
            _, ph = cat_probs(logits_h, d.hs_cats, d.hs_classes)
            _, pl = cat_probs(logits_l, d.ls_cats, d.ls_classes)
            stoch_h = st_sample(ph, noise["u_prior_h"][:, t])
            stoch_l = st_sample(pl, noise["u_prior_l"][:, t])
            for k, v in zip(names, (deter_l, deter_h, hidden_l, hidden_h, logits_l, logits_h, stoch_l, stoch_h), strict=True):
                keep[k].append(v)
        return {k: torch.stack(v, dim=1) for k, v in keep.items()}

    def shared_step(self, batch: tuple[Tensor, ...], noise: dict[str, Tensor]) -> dict[str, Tensor]:
        d = self.dims
        act_in, audio_in, vision_in, _, audio_tgt, vision_tgt = batch
        state0 = self.initial_state(audio_in[:, 0], vision_in[:, 0], noise["u_init_h"], noise["u_init_l"])
        audio_embed = self.audio_encoder(audio_in)
        vision_embed = self.vision_encoder(vision_in)
        roll = self.rollout_representation(act_in, audio_embed, vision_embed, state0, noise)
        feature = torch.cat([roll["deter_h"], roll["post_stoch_h"], roll["deter_l"], roll["post_stoch_l"]], dim=-1)
        nll_a = gaussian_nll(self.audio_decoder(feature), audio_tgt)
        nll_v = gaussian_nll(self.vision_decoder(feature), vision_tgt)
        kl_l = kl_loss(roll["post_logits_l"], roll["prior_logits_l"], d.ls_cats, d.ls_classes, d.use_kl_balancing) * d.kl_coeff
        kl_h = kl_loss(roll["post_logits_h"], roll["prior_logits_h"], d.hs_cats, d.hs_classes, d.use_kl_balancing) * (
            d.kl_coeff * d.w_kl_h
        )
        out = {
            "loss": nll_a + nll_v + kl_l + kl_h,
            "recon": nll_a + nll_v,
            "recon/audio": nll_a,
            "recon/vision": nll_v,
            "kl": kl_l,
            "kl_h": kl_h,
        }
        out.update({f"_{k}": v for k, v in roll.items()})
        out.update({f"_init_{k}": v for k, v in state0.items()})
        out["_audio_embed"], out["_vision_embed"] = audio_embed, vision_embed
        return out
