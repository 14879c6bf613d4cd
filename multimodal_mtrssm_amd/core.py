"""``MoPoE_MRSSM`` / ``MoPoE_MMTRSSM`` with the reference's Python surface, on HIP kernels.

Drop-in for ``multimodal_rssm.models.mrssm.mopoe_mrssm.MoPoE_MRSSM``
(``mrssm/mopoe_mrssm/core.py:12-355`` over ``models/core.py:13-266``) and
``multimodal_rssm.models.mmtrssm.mopoe_mmtrssm.MoPoE_MMTRSSM`` (``mmtrssm/mopoe_mmtrssm/core.py:77-610``):
same kwargs-only constructors, method names, loss-dict keys, state-dict names, ``TypeError`` when
``observations`` is not a tuple.  A YAML ``class_path`` can name these classes directly
(INTEGRATION.md).

What differs underneath: the T loop is one persistent HIP launch (``scan.py``); encoders and decoders
see all B*T frames as one batch; the KL term comes out of the scan kernel; the encoder runs once per
``shared_step`` (the reference encodes frame 0 twice, ``core.py:132`` and ``mrssm core.py:215-216`` -- same
values, and the gradient of both uses is accumulated).  Sampling noise is explicit: every rollout
takes an optional ``noise`` dict of uniforms and draws ``torch.rand`` on the device otherwise.
"""

from __future__ import annotations

import torch
from torch import Tensor, nn

from multimodal_mtrssm_amd import cnn, conv, scan
from multimodal_mtrssm_amd.distributions import MultiOneHot, MultiOneHotFactory, draw_uniforms, kl_divergence, onehot_from_uniforms
from multimodal_mtrssm_amd.networks import MTRNN, Representation, Transition
from multimodal_mtrssm_amd.objective import likelihood
from multimodal_mtrssm_amd.state import MTState, State

try:  # the real trainer, when it is installed next to the reference
    from lightning import LightningModule as _Base
except ImportError:  # this image: a LightningModule-shaped nn.Module

    class _Base(nn.Module):  # type: ignore[no-redef]
        @property
        def device(self) -> torch.device:
            for p in self.parameters():
                return p.device
            return torch.device("cpu")

        def log_dict(self, *args, **kwargs) -> None:  # noqa: ANN002, ANN003
            return None


Noise = dict[str, Tensor | None]

def _pairable(a: nn.Module, b: nn.Module, kind: type) -> bool:
    """Paired launches (``conv.paired``): the audio and the vision stack's equal layers share one grid (this package's own
    stacks only).  Everything runs on ONE stream: a two-stream variant of the step was removed in round 2 (it bought nothing
    once the layers were paired and stalled the GPU in about one run in fifteen, profiles/round1_notes.md)."""
    return conv.PAIR_LAUNCH and isinstance(a, kind) and isinstance(b, kind)


def _st_onehot(dist: MultiOneHot, u: Tensor | None) -> Tensor:
    """Straight-through one-hot sample of ``dist`` from uniforms ``u`` (drawn on device when None)."""
    if u is None:
        return dist.rsample()
    onehot = onehot_from_uniforms(dist.probs.detach(), u.to(dist.probs))
    return (onehot + (dist.probs - dist.probs.detach())).flatten(start_dim=-2)


class _CategoricalHead(torch.autograd.Function):
    """Initial-state categorical head in ONE launch (``mtrssm_categorical_sample_fwd``): flat logits + uniforms ->
    (straight-through sample, log-probabilities, probabilities).  Same values as ``factory(logits)`` + ``_st_onehot`` -- a dozen
    eager launches per level --, same gradients: the sample's gradient flows into the probabilities (``onehot + p - p.detach()``)."""

    @staticmethod
    def forward(ctx, logits: Tensor, u: Tensor, cats: int, classes: int):  # noqa: ANN001, ANN205
        from multimodal_mtrssm_amd import _lib  # noqa: PLC0415

        ctx.set_materialize_grads(False)
        logits, u = logits.contiguous().float(), u.contiguous().float()
        rows = logits.numel() // (cats * classes)
        logp = torch.empty(*logits.shape[:-1], cats, classes, device=logits.device, dtype=torch.float32)
        probs, onehot = torch.empty_like(logp), torch.empty_like(logits)
        _lib.check(_lib.load().mtrssm_categorical_sample_fwd(_lib.ptr(logits), _lib.ptr(u), rows, cats, classes, _lib.ptr(logp), _lib.ptr(probs),
                                                             _lib.ptr(onehot), _lib.stream_ptr(logits.device)), "mtrssm_categorical_sample_fwd")
        ctx.save_for_backward(probs)
        ctx.dims = (rows, cats, classes)
        return onehot, logp, probs

    @staticmethod
    def backward(ctx, g_stoch, g_logp, g_probs):  # noqa: ANN001, ANN205
        from multimodal_mtrssm_amd import _lib  # noqa: PLC0415

        (probs,) = ctx.saved_tensors
        rows, cats, classes = ctx.dims
        gp = None
        if g_stoch is not None:
            gp = g_stoch.reshape(probs.shape)
        if g_probs is not None:
            gp = g_probs if gp is None else gp + g_probs
        if gp is None and g_logp is None:
            return None, None, None, None
        gp = None if gp is None else gp.contiguous().float()
        gl = None if g_logp is None else g_logp.contiguous().float()
        d = torch.empty(*probs.shape[:-2], cats * classes, device=probs.device, dtype=torch.float32)
        _lib.check(_lib.load().mtrssm_categorical_sample_bwd(_lib.ptr(probs), _lib.ptr(gp), _lib.ptr(gl), rows, cats, classes, _lib.ptr(d),
                                                             _lib.stream_ptr(probs.device)), "mtrssm_categorical_sample_bwd")
        return d, None, None, None


class _ElboCombine(torch.autograd.Function):
    """``recon = nll_a + nll_v; kl_j = coeff_j mean(kl_bt_j); loss = recon + sum kl_j`` in one launch each way
    (``mtrssm_elbo_combine_fwd / _bwd``) instead of the eager add / mean / mul / add chain and its autograd nodes."""

    @staticmethod
    def forward(ctx, nll_a: Tensor, nll_v: Tensor, kl0: Tensor, kl1: Tensor | None, c0: float, c1: float):  # noqa: ANN001, ANN205, PLR0913
        from multimodal_mtrssm_amd import _lib  # noqa: PLC0415

        ctx.set_materialize_grads(False)
        nll_a, nll_v, kl0 = nll_a.contiguous().float(), nll_v.contiguous().float(), kl0.contiguous().float()
        kl1 = None if kl1 is None else kl1.contiguous().float()
        outs = [torch.empty((), device=kl0.device, dtype=torch.float32) for _ in range(4)]
        _lib.check(_lib.load().mtrssm_elbo_combine_fwd(_lib.ptr(nll_a), _lib.ptr(nll_v), _lib.ptr(kl0), _lib.ptr(kl1), kl0.numel(), float(c0), float(c1),
                                                       _lib.ptr(outs[0]), _lib.ptr(outs[1]), _lib.ptr(outs[2]) if kl1 is not None else None,
                                                       _lib.ptr(outs[3]), _lib.stream_ptr(kl0.device)), "mtrssm_elbo_combine_fwd")
        ctx.meta = (kl0.shape, None if kl1 is None else kl1.shape, float(c0), float(c1))
        ctx.dev = kl0.device
        return outs[0], outs[1], outs[2], outs[3]

    @staticmethod
    def backward(ctx, g_recon, g_k0, g_k1, g_loss):  # noqa: ANN001, ANN205
        from multimodal_mtrssm_amd import _lib  # noqa: PLC0415

        shape0, shape1, c0, c1 = ctx.meta
        n = 1
        for d in shape0:
            n *= d
        g_a, g_v = (torch.empty((), device=ctx.dev, dtype=torch.float32) for _ in range(2))
        g_kl0 = torch.empty(shape0, device=ctx.dev, dtype=torch.float32)
        g_kl1 = None if shape1 is None else torch.empty(shape1, device=ctx.dev, dtype=torch.float32)
        opt = lambda t: None if t is None else _lib.ptr(t.contiguous().float())  # noqa: E731
        _lib.check(_lib.load().mtrssm_elbo_combine_bwd(opt(g_recon), opt(g_k0), opt(g_k1) if shape1 is not None else None, opt(g_loss), n, c0, c1,
                                                       _lib.ptr(g_a), _lib.ptr(g_v), _lib.ptr(g_kl0), _lib.ptr(g_kl1), _lib.stream_ptr(ctx.dev)),
                   "mtrssm_elbo_combine_bwd")
        return g_a, g_v, g_kl0, g_kl1, None, None


def _elbo(nll_a: Tensor, nll_v: Tensor, kl0: Tensor, c0: float, kl1: Tensor | None = None, c1: float = 0.0) -> tuple[Tensor, Tensor, Tensor, Tensor]:
    """``(recon, kl_0, kl_1, loss)``; on the GPU one fused launch, elsewhere the eager arithmetic of the reference."""
    if kl0.is_cuda and nll_a.dim() == 0 and nll_v.dim() == 0:
        return _ElboCombine.apply(nll_a, nll_v, kl0, kl1, c0, c1)
    recon = nll_a + nll_v
    k0 = kl0.mean().mul(c0)
    k1 = kl1.mean().mul(c1) if kl1 is not None else torch.zeros((), device=kl0.device)
    return recon, k0, k1, recon + k0 + (k1 if kl1 is not None else 0.0)


def _sampled_head(factory, logits: Tensor, u: Tensor | None) -> tuple[MultiOneHot, Tensor]:  # noqa: ANN001
    """``(factory(logits), straight-through sample)``: one fused launch on the GPU when the uniforms are given."""
    if not logits.is_cuda:
        dist = factory(logits)
        return dist, _st_onehot(dist, u)
    if u is None:  # (the draw MultiOneHot.rsample would make: injected noise tape or the device generator)
        u = draw_uniforms((*logits.shape[:-1], factory.category_size), logits)
    stoch, logp, probs = _CategoricalHead.apply(logits, u.to(logits), factory.category_size, factory.class_size)
    return MultiOneHot(logp, probs), stoch


def _rand(like: Tensor, *shape: int) -> Tensor:
    return torch.rand(shape, device=like.device, dtype=torch.float32)


class MoPoE_MRSSM(_Base):  # noqa: N801
    """Multimodal RSSM with MoPoE posteriors: PoE of {audio, vision}, then MoE over {A, V, A+V}."""

    def __init__(  # noqa: PLR0913
        self,
        *,
        audio_representation: Representation,
        vision_representation: Representation,
        transition: Transition,
        audio_encoder: nn.Module,
        vision_encoder: nn.Module,
        audio_decoder: nn.Module,
        vision_decoder: nn.Module,
        init_proj: nn.Module,
        kl_coeff: float,
        use_kl_balancing: bool,
    ) -> None:
        super().__init__()
        # registration order and the double registration of the audio head follow the reference
        # (core.py:27-29, mrssm core.py:55-60): checkpoints carry both `representation.*` and `audio_representation.*`
        self.representation = audio_representation
        self.transition = transition
        self.init_proj = init_proj
        self.kl_coeff = kl_coeff
        self.use_kl_balancing = use_kl_balancing
        self.audio_representation = audio_representation
        self.vision_representation = vision_representation
        self.audio_encoder = audio_encoder
        self.vision_encoder = vision_encoder
        self.audio_decoder = audio_decoder
        self.vision_decoder = vision_decoder
        self.scan_rows_per_block = 0  # 0 = library default (tuning knobs)
        self.scan_threads = 0

    # -- batch accessors (mrssm core.py:310-355) --------------------------------------------
    @staticmethod
    def get_observations_from_batch(batch: tuple[Tensor, ...]) -> tuple[Tensor, Tensor]:
        return batch[1], batch[2]

    @staticmethod
    def get_initial_observation(observations: tuple[Tensor, Tensor]) -> tuple[Tensor, Tensor]:
        audio_obs, vision_obs = observations
        return audio_obs[:, 0], vision_obs[:, 0]

    @staticmethod
    def get_targets_from_batch(batch: tuple[Tensor, ...]) -> dict[str, Tensor]:
        return {"recon/audio": batch[4], "recon/vision": batch[5]}

    # -- encoders / decoders / losses ---------------------------------------------------------
    def encode_observation(self, observation: tuple[Tensor, Tensor] | Tensor) -> Tensor:
        if isinstance(observation, tuple):
            audio_obs, vision_obs = observation
            return (self.audio_encoder(audio_obs) + self.vision_encoder(vision_obs)) / 2.0
        return observation

    def decode_state(self, state: State | MTState) -> dict[str, Tensor]:
        return {"recon/audio": self.audio_decoder(state.feature), "recon/vision": self.vision_decoder(state.feature)}

    @staticmethod
    def compute_reconstruction_loss(reconstructions: dict[str, Tensor], targets: dict[str, Tensor]) -> dict[str, Tensor]:
        audio = likelihood(prediction=reconstructions["recon/audio"], target=targets["recon/audio"], event_ndims=3)
        vision = likelihood(prediction=reconstructions["recon/vision"], target=targets["recon/vision"], event_ndims=3)
        if not sum_recon:  # (shared_step adds them in its fused scalar epilogue)
            return {"recon/audio": audio, "recon/vision": vision}
        return {"recon": audio + vision, "recon/audio": audio, "recon/vision": vision}

    def _encode_both(self, audio_obs: Tensor, vision_obs: Tensor) -> tuple[Tensor, Tensor]:
        """Both encoders over all B*T frames; equal-shaped residual blocks of the two stacks share launches."""
        if _pairable(self.audio_encoder, self.vision_encoder, cnn.Encoder):
            return cnn.encode_pair(self.audio_encoder, self.vision_encoder, audio_obs, vision_obs)
        return self.audio_encoder(audio_obs), self.vision_encoder(vision_obs)

    def _reconstruction_losses(self, feature: Tensor, targets: dict[str, Tensor], *, sum_recon: bool = True) -> dict[str, Tensor]:
        """``decode_state`` + ``compute_reconstruction_loss`` (``mrssm core.py:262-308``).  With this package's decoders the
        out_activation (Tanh) is applied inside the NLL kernels: the activated reconstructions are never written in training."""
        da, dv = self.audio_decoder, self.vision_decoder
        fused = isinstance(da, cnn.Decoder) and isinstance(dv, cnn.Decoder) and da.out_act_id is not None and dv.out_act_id is not None
        if fused and _pairable(da, dv, cnn.Decoder):
            pa, pv = cnn.decode_pair(da, dv, feature, feature, raw=True)
        elif fused:
            pa, pv = da(feature, raw=True), dv(feature, raw=True)
        else:
            pa, pv = da(feature), dv(feature)
        audio = likelihood(prediction=pa, target=targets["recon/audio"], event_ndims=3, out_act=da.out_act_id if fused else 0)
        vision = likelihood(prediction=pv, target=targets["recon/vision"], event_ndims=3, out_act=dv.out_act_id if fused else 0)
        if not sum_recon:  # (shared_step adds them in its fused scalar epilogue)
            return {"recon/audio": audio, "recon/vision": vision}
        return {"recon": audio + vision, "recon/audio": audio, "recon/vision": vision}

    # -- states ---------------------------------------------------------------------------------
    def _initial_from_embed(self, obs_embed: Tensor, u_init: Tensor | None) -> State:
        deter = self.init_proj(obs_embed)
        logits = self.transition.rnn_to_prior_projector(deter)
        dist, stoch = _sampled_head(self.representation.distribution_factory, logits, u_init)
        return State(deter=deter, distribution=dist, stoch=stoch)

    def initial_state(self, observation: tuple[Tensor, Tensor] | Tensor, noise: Noise | None = None) -> State:
        """``core.py:121-135``: fused embedding -> ``init_proj`` -> prior head -> sampled State."""
        u = None if noise is None else noise.get("u_init")
        return self._initial_from_embed(self.encode_observation(observation), u).to(self.device)

    def noise_shapes(self, batch: int, steps: int) -> dict[str, tuple[int, ...]]:
        """Uniforms one ``shared_step`` consumes (one per categorical and draw; row = batch row).  Data-parallel runs
        draw them for the GLOBAL batch and slice rows (``parallel.GlobalRowNoise``) so results do not depend on the rank count."""
        k = self.transition.distribution_factory.category_size
        return {"u_init": (batch, k), "u_post": (batch, steps, k)}

    def _rollout_embedded(self, actions: Tensor, audio_embed: Tensor, vision_embed: Tensor, prev_state: State,
                          noise: Noise | None, *, sample_prior: bool) -> dict[str, Tensor]:
        noise = noise or {}
        B, T = actions.shape[:2]
        K = self.transition.distribution_factory.category_size
        u_post = noise.get("u_post")
        u_prior = noise.get("u_prior")
        if u_post is None:
            u_post = _rand(actions, B, T, K)
        if u_prior is None and sample_prior:
            u_prior = _rand(actions, B, T, K)
        return scan.mrssm_posterior_rollout(
            self.transition, self.audio_representation, self.vision_representation, actions, audio_embed, vision_embed,
            prev_state.deter, prev_state.stoch, u_post, u_prior, balancing=bool(self.use_kl_balancing),
            rows_per_block=self.scan_rows_per_block, threads=self.scan_threads,
        )

    def _states_from_rollout(self, out: dict[str, Tensor]) -> tuple[State, State]:
        factory = self.audio_representation.distribution_factory
        post = State(deter=out["deter"], distribution=factory(out["post_logits"]), stoch=out["post_stoch"])
        prior = State(deter=out["deter"], distribution=factory(out["prior_logits"]), stoch=out["prior_stoch"])
        post.kl_per_step = out["kl"]  # sum_K KL(q||p) per (b, t), straight from the scan kernel
        return post, prior

    def rollout_representation(
        self,
        *,
        actions: Tensor,
        observations: Tensor | tuple[Tensor, ...],
        prev_state: State,
        noise: Noise | None = None,
    ) -> tuple[State, State]:
        """``mrssm core.py:184-260``: returns (mixed posterior, prior), each ``[B, T, .]``."""
        if not isinstance(observations, tuple):
            msg = "MoPoE-MRSSM requires tuple of (audio_obs, vision_obs)"
            raise TypeError(msg)
        audio_obs, vision_obs = observations
        out = self._rollout_embedded(actions, self.audio_encoder(audio_obs), self.vision_encoder(vision_obs), prev_state,
                                     noise, sample_prior=True)
        return self._states_from_rollout(out)

    def rollout_transition(self, *, actions: Tensor, prev_state: State, noise: Noise | None = None) -> State:
        """``core.py:170-185``: prior-only rollout (callbacks / evaluation)."""
        u = None if noise is None else noise.get("u_prior")
        out = scan.mrssm_prior_rollout(self.transition, actions, prev_state.deter, prev_state.stoch, u,
                                       rows_per_block=self.scan_rows_per_block, threads=self.scan_threads)
        dist = self.transition.distribution_factory(out["prior_logits"])
        return State(deter=out["deter"], distribution=dist, stoch=out["prior_stoch"])

    # -- train / val ----------------------------------------------------------------------------
    def shared_step(self, batch: tuple[Tensor, ...], noise: Noise | None = None) -> dict[str, Tensor]:
        """``core.py:187-221``: ``loss = recon + kl_coeff * KL(post || prior)``."""
        action_input = batch[0]
        audio_obs, vision_obs = self.get_observations_from_batch(batch)
        conv.begin_step(audio_obs.device)
        audio_embed, vision_embed = self._encode_both(audio_obs, vision_obs)
        u_init = None if noise is None else noise.get("u_init")
        state0 = self._initial_from_embed((audio_embed[:, 0] + vision_embed[:, 0]) / 2.0, u_init)
        out = self._rollout_embedded(action_input, audio_embed, vision_embed, state0, noise, sample_prior=False)
        feature = torch.cat([out["deter"], out["post_stoch"]], dim=-1)
        parts = self._reconstruction_losses(feature, self.get_targets_from_batch(batch), sum_recon=False)
        recon, kl_div, _, loss = _elbo(parts["recon/audio"], parts["recon/vision"], out["kl"], float(self.kl_coeff))
        return {"recon": recon, **parts, "kl": kl_div, "loss": loss}

    def _step(self, batch: tuple[Tensor, ...], prefix: str, *, with_loss_key: bool) -> dict[str, Tensor]:
        loss_dict = self.shared_step(batch)
        renamed = {"loss": loss_dict["loss"]} if with_loss_key else {}
        renamed[f"{prefix}/loss"] = loss_dict["loss"]
        for key, value in loss_dict.items():
            if key != "loss":
                renamed[f"{prefix}/{key}"] = value
        self.log_dict(renamed, prog_bar=True, sync_dist=True, on_step=False, on_epoch=True)
        return renamed

    def training_step(self, batch: tuple[Tensor, ...], _: int = 0) -> dict[str, Tensor]:
        return self._step(batch, "train", with_loss_key=True)

    def validation_step(self, batch: tuple[Tensor, ...], _batch_index: int = 0) -> dict[str, Tensor]:
        return self._step(batch, "val", with_loss_key=False)


class MoPoE_MMTRSSM(MoPoE_MRSSM):  # noqa: N801
    """Two-timescale variant: MTRNN lower (tau_l) / higher (tau_h) levels, MoPoE on the lower level."""

    def __init__(  # noqa: PLR0913
        self,
        *,
        audio_representation: Representation,
        vision_representation: Representation,
        audio_encoder: nn.Module,
        vision_encoder: nn.Module,
        audio_decoder: nn.Module,
        vision_decoder: nn.Module,
        init_proj: nn.Module,
        kl_coeff: float,
        use_kl_balancing: bool,
        action_size: int,
        hd_dim: int,
        hs_dim: int,
        ld_dim: int,
        ls_dim: int,
        l_tau: float,
        h_tau: float,
        l_prior: nn.Module,
        l_posterior: nn.Module,
        h_prior: nn.Module,
        h_posterior: nn.Module,
        l_dist: MultiOneHotFactory,
        h_dist: MultiOneHotFactory,
        w_kl_h: float = 1.0,
    ) -> None:
        # the reference registers a never-trained Transition (A=1, S=1) for its base class (core.py:143-151);
        # kept so that state-dicts interchange
        dummy_transition = Transition(deterministic_size=ld_dim, hidden_size=ld_dim, action_size=1,
                                      distribution_config=[1, 1], activation_name="ELU")
        super().__init__(
            audio_representation=audio_representation, vision_representation=vision_representation,
            transition=dummy_transition, audio_encoder=audio_encoder, vision_encoder=vision_encoder,
            audio_decoder=audio_decoder, vision_decoder=vision_decoder, init_proj=init_proj, kl_coeff=kl_coeff,
            use_kl_balancing=use_kl_balancing,
        )
        self.action_dim = action_size
        self.hd_dim, self.hs_dim, self.ld_dim, self.ls_dim = hd_dim, hs_dim, ld_dim, ls_dim
        self.w_kl_h = w_kl_h
        self.l_rnn = MTRNN(input_dim=action_size + ls_dim + hs_dim, hidden_dim=ld_dim, tau=l_tau)
        self.h_rnn = MTRNN(input_dim=hs_dim, hidden_dim=hd_dim, tau=h_tau)
        self.l_prior = l_prior
        self.l_posterior = l_posterior  # registered, never used on the path upstream either (core.py:188)
        self.h_prior = h_prior
        self.h_posterior = h_posterior
        self.l_dist = l_dist
        self.h_dist = h_dist

    @property
    def feature_dim(self) -> int:
        return self.hd_dim + self.hs_dim + self.ld_dim + self.ls_dim

    def _initial_from_embed(self, obs_embed: Tensor, noise: Noise | None) -> MTState:  # type: ignore[override]
        h = self.init_proj(obs_embed)
        higher, lower = h[..., : self.hd_dim], h[..., self.hd_dim :]
        noise = noise or {}
        h_dist, stoch_h = _sampled_head(self.h_dist, self.h_prior(higher), noise.get("u_init_h"))
        l_dist, stoch_l = _sampled_head(self.l_dist, self.l_prior(lower), noise.get("u_init_l"))
        return MTState(
            deter_h=higher, deter_l=lower, distribution_h=h_dist, distribution_l=l_dist, hidden_h=higher, hidden_l=lower,
            stoch_h=stoch_h, stoch_l=stoch_l,
        )

    def initial_state(self, observation: tuple[Tensor, Tensor] | Tensor, noise: Noise | None = None) -> MTState:  # type: ignore[override]
        """``mmtrssm core.py:321-362``: ``init_proj`` output split into raw hiddens = deters (no tanh at t=0)."""
        obs_embed = self.encode_observation(observation) if isinstance(observation, tuple) else observation
        return self._initial_from_embed(obs_embed, noise).to(obs_embed.device)

    def noise_shapes(self, batch: int, steps: int) -> dict[str, tuple[int, ...]]:  # type: ignore[override]
        kl, kh = self.l_dist.category_size, self.h_dist.category_size
        return {"u_init_h": (batch, kh), "u_init_l": (batch, kl), "u_post_l": (batch, steps, kl), "u_post_h": (batch, steps, kh)}

    @staticmethod
    def _state_dict_of(state: MTState) -> dict[str, Tensor]:
        return {"deter_l": state.deter_l, "deter_h": state.deter_h, "hidden_l": state.hidden_l, "hidden_h": state.hidden_h,
                "stoch_l": state.stoch_l, "stoch_h": state.stoch_h}

    def _rollout_embedded(self, actions: Tensor, audio_embed: Tensor, vision_embed: Tensor, prev_state: MTState,  # type: ignore[override]
                          noise: Noise | None, *, sample_prior: bool) -> dict[str, Tensor]:
        noise = dict(noise or {})
        B, T = actions.shape[:2]
        KL, KH = self.l_dist.category_size, self.h_dist.category_size
        for key, k in (("u_post_l", KL), ("u_post_h", KH)):
            if noise.get(key) is None:
                noise[key] = _rand(actions, B, T, k)
        if sample_prior:
            for key, k in (("u_prior_l", KL), ("u_prior_h", KH)):
                if noise.get(key) is None:
                    noise[key] = _rand(actions, B, T, k)
        return scan.mmtrssm_posterior_rollout(self, actions, audio_embed, vision_embed, self._state_dict_of(prev_state), noise,
                                              rows_per_block=self.scan_rows_per_block, threads=self.scan_threads)

    def _states_from_rollout(self, out: dict[str, Tensor]) -> tuple[MTState, MTState]:  # type: ignore[override]
        common = dict(deter_h=out["deter_h"], deter_l=out["deter_l"], hidden_h=out["hidden_h"], hidden_l=out["hidden_l"])
        post = MTState(distribution_h=self.h_dist(out["post_logits_h"]), distribution_l=self.l_dist(out["post_logits_l"]),
                       stoch_h=out["post_stoch_h"], stoch_l=out["post_stoch_l"], **common)
        prior = MTState(distribution_h=self.h_dist(out["prior_logits_h"]), distribution_l=self.l_dist(out["prior_logits_l"]),
                        stoch_h=out["prior_stoch_h"], stoch_l=out["prior_stoch_l"], **common)
        post.kl_per_step, post.kl_h_per_step = out["kl_l"], out["kl_h"]
        return post, prior

    def rollout_representation(  # type: ignore[override]
        self,
        *,
        actions: Tensor,
        observations: Tensor | tuple[Tensor, ...],
        prev_state: MTState,
        noise: Noise | None = None,
    ) -> tuple[MTState, MTState]:
        """``mmtrssm core.py:364-494``."""
        if not isinstance(observations, tuple):
            msg = "MoPoE-MMTRSSM requires tuple of (audio_obs, vision_obs)"
            raise TypeError(msg)
        audio_obs, vision_obs = observations
        out = self._rollout_embedded(actions, self.audio_encoder(audio_obs), self.vision_encoder(vision_obs), prev_state,
                                     noise, sample_prior=True)
        return self._states_from_rollout(out)

    def rollout_transition(self, *, actions: Tensor, prev_state: MTState, noise: Noise | None = None) -> MTState:  # type: ignore[override]
        """``mmtrssm core.py:496-544``."""
        out = scan.mmtrssm_prior_rollout(self, actions, self._state_dict_of(prev_state), noise or {},
                                         rows_per_block=self.scan_rows_per_block, threads=self.scan_threads)
        return MTState(
            deter_h=out["deter_h"], deter_l=out["deter_l"], distribution_h=self.h_dist(out["prior_logits_h"]),
            distribution_l=self.l_dist(out["prior_logits_l"]), hidden_h=out["hidden_h"], hidden_l=out["hidden_l"],
            stoch_h=out["prior_stoch_h"], stoch_l=out["prior_stoch_l"],
        )

    def shared_step(self, batch: tuple[Tensor, ...], noise: Noise | None = None) -> dict[str, Tensor]:
        """``mmtrssm core.py:563-606``: ``loss = recon + kl_coeff KL_l + kl_coeff w_kl_h KL_h``."""
        action_input = batch[0]
        audio_obs, vision_obs = self.get_observations_from_batch(batch)
        conv.begin_step(audio_obs.device)
        audio_embed, vision_embed = self._encode_both(audio_obs, vision_obs)
        state0 = self._initial_from_embed((audio_embed[:, 0] + vision_embed[:, 0]) / 2.0, noise)
        out = self._rollout_embedded(action_input, audio_embed, vision_embed, state0, noise, sample_prior=False)
        feature = torch.cat([out["deter_h"], out["post_stoch_h"], out["deter_l"], out["post_stoch_l"]], dim=-1)
        parts = self._reconstruction_losses(feature, self.get_targets_from_batch(batch), sum_recon=False)
        recon, kl_div_l, kl_div_h, loss = _elbo(parts["recon/audio"], parts["recon/vision"], out["kl_l"], float(self.kl_coeff), out["kl_h"],
                                                float(self.kl_coeff * self.w_kl_h))
        return {"recon": recon, **parts, "kl": kl_div_l, "kl_h": kl_div_h, "loss": loss}


__all__ = ["MoPoE_MMTRSSM", "MoPoE_MRSSM", "kl_divergence"]
