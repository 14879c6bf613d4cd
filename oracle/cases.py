"""ORACLE (test infrastructure) -- the parity cases: dims, synthetic inputs, seeded weights, noise.

Shared by ``oracle/gen_golden.py`` (build container: writes ``tests/golden/*.npz``), by ``tests/`` and
by ``bench.py``'s ``cpu_baseline`` leg.  Everything is a pure function of the case's seeds so the
(large) weights never need to be stored: fixtures keep per-parameter checksums instead and the
tests re-create the weights with ``torch.manual_seed`` (same image -> same bits).

Synthetic inputs follow SURVEY.md section 8d: targets ~ U(-1, 1), inputs = targets + N(0, 0.1^2)
(mirrors ``transform.py:55-72``), actions ~ N(0, 1), uniforms u ~ U(0, 1) per categorical.
"""

from __future__ import annotations

from dataclasses import dataclass, replace
from typing import Any

import torch
from torch import Tensor

from oracle.ref_dists import sampling_margin
from oracle.ref_model import MMTRSSMDims, MRSSMDims, OracleMMTRSSM, OracleMRSSM, cat_probs

MARGIN = 1e-3  # min |u - cdf| a fixture draw must keep ("Hard parts")


def encoder_config(  # noqa: PLR0913
    input_shape: tuple[int, int, int],
    embed: int,
    channels: tuple[int, ...] = (8, 16, 32),
    res_blocks: int = 3,
    res_inter: int = 64,
    res_out: int = 64,
    activation: str = "ELU",
) -> dict[str, Any]:
    """Field names follow ``mrssm/mopoe_mrssm/configs/default.yaml:31-60`` (+ ``input_shape``)."""
    n = len(channels)
    return {
        "linear_sizes": [embed],
        "activation_name": activation,
        "out_activation_name": "Identity",
        "channels": list(channels),
        "kernel_sizes": [3] * n,
        "strides": [2] * n,
        "paddings": [1] * n,
        "num_residual_blocks": res_blocks,
        "residual_intermediate_size": res_inter,
        "residual_output_size": res_out,
        "coord_conv": True,
        "input_shape": list(input_shape),
    }


def decoder_config(  # noqa: PLR0913
    in_features: int,
    out_shape: tuple[int, int, int],
    channels: tuple[int, ...] = (32, 16),
    res_blocks: int = 3,
    res_inter: int = 128,
    res_in: int = 64,
    hidden: int = 64,
    activation: str = "ELU",
) -> dict[str, Any]:
    """Field names follow ``default.yaml:61-92`` (+ ``in_features``); ``channels`` excludes the output channel."""
    c, h, w = out_shape
    n = len(channels) + 1
    h0, w0 = h >> n, w >> n
    return {
        "linear_sizes": [hidden, res_in * h0 * w0],
        "conv_in_shape": [res_in, h0, w0],
        "activation_name": activation,
        "out_activation_name": "Tanh",
        "channels": [*channels, c],
        "kernel_sizes": [4] * n,
        "strides": [2] * n,
        "paddings": [1] * n,
        "output_paddings": [0] * n,
        "num_residual_blocks": res_blocks,
        "residual_intermediate_size": res_inter,
        "residual_input_size": res_in,
        "in_features": in_features,
    }


@dataclass
class Case:
    name: str
    kind: str  # "mrssm" | "mmtrssm"
    dims: Any
    batch: int
    steps: int
    audio_shape: tuple[int, int, int]
    vision_shape: tuple[int, int, int]
    weight_seed: int = 42
    data_seed: int = 1234
    noise_seed: int = 7
    query: int = 5  # rollout_transition starts from posterior[:, query-1]
    head_gain: float = 4.0  # last-layer gain of every prior/posterior head (default init gives KL ~ 0)


def _mrssm_dims(deter, hidden, classes, cats, action, embed, a_shape, v_shape, **enc_dec) -> MRSSMDims:  # noqa: ANN001, ANN003, PLR0913
    feat = deter + classes * cats
    enc_kw = {k[4:]: v for k, v in enc_dec.items() if k.startswith("enc_")}
    dec_kw = {k[4:]: v for k, v in enc_dec.items() if k.startswith("dec_")}
    return MRSSMDims(
        deter=deter, hidden=hidden, classes=classes, cats=cats, action=action, embed=embed,
        enc_audio=encoder_config(a_shape, embed, **enc_kw), enc_vision=encoder_config(v_shape, embed, **enc_kw),
        dec_audio=decoder_config(feat, a_shape, **dec_kw), dec_vision=decoder_config(feat, v_shape, **dec_kw),
    )


def _mmtrssm_dims(hd, hs, ld, ls, hidden, action, embed, a_shape, v_shape, **enc_dec) -> MMTRSSMDims:  # noqa: ANN001, ANN003, PLR0913
    feat = hd + hs[0] * hs[1] + ld + ls[0] * ls[1]
    enc_kw = {k[4:]: v for k, v in enc_dec.items() if k.startswith("enc_")}
    dec_kw = {k[4:]: v for k, v in enc_dec.items() if k.startswith("dec_")}
    return MMTRSSMDims(
        hd=hd, hs_classes=hs[0], hs_cats=hs[1], ld=ld, ls_classes=ls[0], ls_cats=ls[1], hidden=hidden,
        action=action, embed=embed,
        enc_audio=encoder_config(a_shape, embed, **enc_kw), enc_vision=encoder_config(v_shape, embed, **enc_kw),
        dec_audio=decoder_config(feat, a_shape, **dec_kw), dec_vision=decoder_config(feat, v_shape, **dec_kw),
    )


_SMALL = {"enc_channels": (4, 8), "enc_res_blocks": 1, "enc_res_inter": 16, "enc_res_out": 16,
          "dec_channels": (8,), "dec_res_blocks": 1, "dec_res_inter": 16, "dec_res_in": 16, "dec_hidden": 32}

CASES: dict[str, Case] = {
    # (i) default.yaml dims (mrssm yaml 10-13, 28-29, 31-99), BASELINE config 1: B=2 T=16
    "mrssm_default": Case(
        "mrssm_default", "mrssm",
        _mrssm_dims(32, 32, 4, 4, 6, 64, (1, 32, 32), (1, 32, 32)), 2, 16, (1, 32, 32), (1, 32, 32), query=10,
    ),
    # (ii) non-square categorical (2 classes x 8 categoricals), ragged sizes, different audio/vision shapes
    "mrssm_nonsquare": Case(
        "mrssm_nonsquare", "mrssm",
        _mrssm_dims(24, 40, 2, 8, 3, 20, (1, 16, 8), (1, 8, 8), **_SMALL), 3, 7, (1, 16, 8), (1, 8, 8), query=3,
    ),
    # (iii) BASELINE config-2 core dims (deter=200 stoch=30=5x6 action=4; build-chosen hidden=200, embed=256), small frames
    "mrssm_cfg2dims": Case(
        "mrssm_cfg2dims", "mrssm",
        _mrssm_dims(200, 200, 5, 6, 4, 256, (1, 16, 8), (1, 8, 8), **_SMALL), 4, 8, (1, 16, 8), (1, 8, 8), query=4,
    ),
    # MTState variant at its default.yaml dims (mmtrssm yaml 95-148): l_dist 4x4, h_dist 2 classes x 8 categoricals
    "mmtrssm_default": Case(
        "mmtrssm_default", "mmtrssm",
        _mmtrssm_dims(32, (2, 8), 32, (4, 4), 32, 6, 64, (1, 32, 32), (1, 32, 32)), 2, 16, (1, 32, 32), (1, 32, 32), query=10,
    ),
    # BASELINE config-3 core dims: ld=hd=200, ls=hs=30
    "mmtrssm_cfg3dims": Case(
        "mmtrssm_cfg3dims", "mmtrssm",
        _mmtrssm_dims(200, (5, 6), 200, (5, 6), 200, 4, 256, (1, 16, 8), (1, 8, 8), **_SMALL), 4, 8, (1, 16, 8), (1, 8, 8), query=4,
    ),
    # BASELINE "Large" core dims (deter=1024 stoch=128=8x16, hidden=1024, embed=1024) at a tiny B, T: exercises the
    # > 64 KiB dynamic-LDS path of the scan kernels.  Not a golden fixture (weights too large): GPU-vs-oracle only.
    "mrssm_large": Case(
        "mrssm_large", "mrssm",
        _mrssm_dims(1024, 1024, 8, 16, 4, 1024, (1, 16, 8), (1, 8, 8), **_SMALL), 3, 4, (1, 16, 8), (1, 8, 8), query=2,
    ),
    "mmtrssm_large": Case(
        "mmtrssm_large", "mmtrssm",
        _mmtrssm_dims(1024, (8, 16), 1024, (8, 16), 1024, 4, 1024, (1, 16, 8), (1, 8, 8), **_SMALL), 3, 4, (1, 16, 8), (1, 8, 8), query=2,
    ),
    # BASELINE configs[1] / configs[2] EXACTLY as bench.py builds them (bench.WORKLOAD): 1x128x32 audio + 1x64x64 vision
    # frames, conv channels [8,16,32] / [32,16,1], 3 residual blocks (64 / 128 intermediate channels), deter = hidden = 200,
    # stoch 6 x 5, action 4, embed 256, T = 50.  Two sequences: the oracle finishes fwd + bwd in seconds.  No fixture
    # (4 M weights): GPU-vs-oracle only; the tests screen the noise seed for a sampling margin as gen_golden.py does.
    "mrssm_bench": Case(
        "mrssm_bench", "mrssm",
        _mrssm_dims(200, 200, 5, 6, 4, 256, (1, 128, 32), (1, 64, 64)), 2, 50, (1, 128, 32), (1, 64, 64), query=25,
    ),
    "mmtrssm_bench": Case(
        "mmtrssm_bench", "mmtrssm",
        _mmtrssm_dims(200, (5, 6), 200, (5, 6), 200, 4, 256, (1, 128, 32), (1, 64, 64)), 2, 50, (1, 128, 32), (1, 64, 64), query=25,
    ),
    # BASELINE configs[4] "Large" with the bench's frames (bench.py --model large): two sequences, T = 100
    "mrssm_large_bench": Case(
        "mrssm_large_bench", "mrssm",
        _mrssm_dims(1024, 1024, 8, 16, 4, 1024, (1, 128, 32), (1, 64, 64)), 2, 100, (1, 128, 32), (1, 64, 64), query=50,
    ),
}

GOLDEN_CASES = ("mrssm_default", "mrssm_nonsquare", "mrssm_cfg2dims", "mmtrssm_default", "mmtrssm_cfg3dims")


def build_model(case: Case) -> torch.nn.Module:
    torch.manual_seed(case.weight_seed)
    model = OracleMRSSM(case.dims) if case.kind == "mrssm" else OracleMMTRSSM(case.dims)
    heads = ("rnn_to_prior_projector.2.", "rnn_to_post_projector.2.", "l_prior.2.", "h_prior.2.", "h_posterior.2.")
    with torch.no_grad():
        for name, p in model.named_parameters():
            if any(h in name for h in heads):
                p.mul_(case.head_gain)
    return model.float()


def build_batch(case: Case, *, batch: int | None = None, steps: int | None = None) -> tuple[Tensor, ...]:
    """6-tuple (action_in, audio_in, vision_in, action_tgt, audio_tgt, vision_tgt), ``mrssm/dataset.py:168-175``."""
    b = batch or case.batch
    t = steps or case.steps
    g = torch.Generator().manual_seed(case.data_seed)
    act_t = torch.randn(b, t, case.dims.action, generator=g)
    aud_t = torch.rand(b, t, *case.audio_shape, generator=g) * 2 - 1
    vis_t = torch.rand(b, t, *case.vision_shape, generator=g) * 2 - 1
    act_i = act_t + 0.1 * torch.randn(act_t.shape, generator=g)
    aud_i = aud_t + 0.1 * torch.randn(aud_t.shape, generator=g)
    vis_i = vis_t + 0.1 * torch.randn(vis_t.shape, generator=g)
    return act_i, aud_i, vis_i, act_t, aud_t, vis_t


def noise_shapes(case: Case, batch: int, steps: int) -> dict[str, tuple[int, ...]]:
    d = case.dims
    if case.kind == "mrssm":
        return {"u_init": (batch, d.cats), "u_prior": (batch, steps, d.cats), "u_post": (batch, steps, d.cats),
                "u_trans": (batch, steps, d.cats)}
    return {
        "u_init_h": (batch, d.hs_cats), "u_init_l": (batch, d.ls_cats),
        "u_post_l": (batch, steps, d.ls_cats), "u_post_h": (batch, steps, d.hs_cats),
        "u_prior_h": (batch, steps, d.hs_cats), "u_prior_l": (batch, steps, d.ls_cats),
        "u_trans_h": (batch, steps, d.hs_cats), "u_trans_l": (batch, steps, d.ls_cats),
    }


def build_noise(case: Case, seed: int | None = None, *, batch: int | None = None, steps: int | None = None) -> dict[str, Tensor]:
    g = torch.Generator().manual_seed(case.noise_seed if seed is None else seed)
    shapes = noise_shapes(case, batch or case.batch, steps or case.steps)
    return {k: torch.rand(s, generator=g) for k, s in shapes.items()}


def min_margin(case: Case, out: dict[str, Tensor], noise: dict[str, Tensor]) -> float:
    """Smallest |u - cdf| over every draw that shapes the trajectory or an output of ``shared_step``."""
    d = case.dims
    pairs: list[tuple[Tensor, Tensor, int, int]]
    if case.kind == "mrssm":
        pairs = [
            (out["_logits0"], noise["u_init"], d.cats, d.classes),
            (out["_prior_logits"], noise["u_prior"], d.cats, d.classes),
            (out["_post_logits"], noise["u_post"], d.cats, d.classes),
        ]
    else:
        pairs = [
            (out["_init_logits_h"], noise["u_init_h"], d.hs_cats, d.hs_classes),
            (out["_init_logits_l"], noise["u_init_l"], d.ls_cats, d.ls_classes),
            (out["_prior_logits_l"], noise["u_prior_l"], d.ls_cats, d.ls_classes),
            (out["_prior_logits_h"], noise["u_prior_h"], d.hs_cats, d.hs_classes),
            (out["_post_logits_l"], noise["u_post_l"], d.ls_cats, d.ls_classes),
            (out["_post_logits_h"], noise["u_post_h"], d.hs_cats, d.hs_classes),
        ]
    worst = 1.0
    for logits, u, cats, classes in pairs:
        _, probs = cat_probs(logits.detach(), cats, classes)
        worst = min(worst, float(sampling_margin(probs, u).min()))
    return worst


def screened_noise(case: Case, model: torch.nn.Module, batch: tuple[Tensor, ...], *, margin: float = 1e-4, first_seed: int = 100,
                   tries: int = 40) -> tuple[dict[str, Tensor], float, int]:
    """The first noise seed whose every draw keeps ``margin`` from the CDF edges under ``model`` (oracle forward passes
    only).  Discrete samples fork the trajectory, so a comparison against a second implementation is meaningful only when
    no draw sits within that implementation's rounding distance (~1e-6) of an edge.  Returns (noise, margin, seed)."""
    best: tuple[dict[str, Tensor], float, int] | None = None
    b, t = batch[0].shape[:2]
    for seed in range(first_seed, first_seed + tries):
        noise = build_noise(case, seed, batch=b, steps=t)
        with torch.no_grad():
            out = model.shared_step(batch, noise)
        m = min_margin(case, out, noise)
        if best is None or m > best[1]:
            best = (noise, m, seed)
        if m >= margin:
            break
    assert best is not None
    return best


def with_sizes(case: Case, batch: int, steps: int) -> Case:
    return replace(case, batch=batch, steps=steps)
