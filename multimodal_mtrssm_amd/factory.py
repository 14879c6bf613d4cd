"""Builders that assemble the models exactly as the reference YAMLs do (``configs/default.yaml``).

``make_mrssm`` / ``make_mmtrssm`` take the dimensions BASELINE.json leaves open (hidden, embed,
categoricals x classes, conv channels) as explicit arguments, so every benchmark line can print them.
"""

from __future__ import annotations

from typing import Any

from torch import nn

from multimodal_mtrssm_amd.cnn import Decoder, Encoder
from multimodal_mtrssm_amd.core import MoPoE_MMTRSSM, MoPoE_MRSSM
from multimodal_mtrssm_amd.distributions import MultiOneHotFactory
from multimodal_mtrssm_amd.networks import MLP, Representation, Transition


def encoder_config(input_shape: tuple[int, int, int], embed: int, channels: tuple[int, ...] = (8, 16, 32),  # noqa: PLR0913
                   res_blocks: int = 3, res_inter: int = 64, res_out: int = 64, activation: str = "ELU") -> dict[str, Any]:
    """``default.yaml:31-60`` field names (+ ``input_shape`` so the Linear can be built eagerly)."""
    n = len(channels)
    return {"linear_sizes": [embed], "activation_name": activation, "out_activation_name": "Identity",
            "channels": list(channels), "kernel_sizes": [3] * n, "strides": [2] * n, "paddings": [1] * n,
            "num_residual_blocks": res_blocks, "residual_intermediate_size": res_inter, "residual_output_size": res_out,
            "coord_conv": True, "input_shape": list(input_shape)}


def decoder_config(in_features: int, out_shape: tuple[int, int, int], channels: tuple[int, ...] = (32, 16),  # noqa: PLR0913
                   res_blocks: int = 3, res_inter: int = 128, res_in: int = 64, hidden: int = 64,
                   activation: str = "ELU") -> dict[str, Any]:
    """``default.yaml:61-92`` field names (+ ``in_features``); ``channels`` excludes the output channel."""
    c, h, w = out_shape
    n = len(channels) + 1
    h0, w0 = h >> n, w >> n
    return {"linear_sizes": [hidden, res_in * h0 * w0], "conv_in_shape": [res_in, h0, w0], "activation_name": activation,
            "out_activation_name": "Tanh", "channels": [*channels, c], "kernel_sizes": [4] * n, "strides": [2] * n,
            "paddings": [1] * n, "output_paddings": [0] * n, "num_residual_blocks": res_blocks,
            "residual_intermediate_size": res_inter, "residual_input_size": res_in, "in_features": in_features}


def make_mrssm(*, deter: int, hidden: int, classes: int, cats: int, action: int, embed: int,  # noqa: PLR0913
               enc_audio: dict[str, Any], enc_vision: dict[str, Any], dec_audio: dict[str, Any], dec_vision: dict[str, Any],
               activation: str = "ELU", init_cells: int = 200, kl_coeff: float = 1.0,
               use_kl_balancing: bool = True) -> MoPoE_MRSSM:
    rep = {"deterministic_size": deter, "hidden_size": hidden, "obs_embed_size": embed,
           "distribution_config": [classes, cats], "activation_name": activation}
    return MoPoE_MRSSM(
        audio_representation=Representation(**rep), vision_representation=Representation(**rep),
        transition=Transition(deterministic_size=deter, hidden_size=hidden, action_size=action,
                              distribution_config=[classes, cats], activation_name=activation),
        audio_encoder=Encoder(enc_audio), vision_encoder=Encoder(enc_vision),
        audio_decoder=Decoder(dec_audio), vision_decoder=Decoder(dec_vision),
        init_proj=MLP(in_features=embed, out_features=deter, num_cells=init_cells, depth=1),
        kl_coeff=kl_coeff, use_kl_balancing=use_kl_balancing,
    )


def make_mmtrssm(*, hd: int, hs: tuple[int, int], ld: int, ls: tuple[int, int], hidden: int, action: int, embed: int,  # noqa: PLR0913
                 enc_audio: dict[str, Any], enc_vision: dict[str, Any], dec_audio: dict[str, Any],
                 dec_vision: dict[str, Any], l_tau: float = 2.0, h_tau: float = 4.0, activation: str = "ELU",
                 init_cells: int = 200, kl_coeff: float = 1.0, w_kl_h: float = 1.0,
                 use_kl_balancing: bool = True) -> MoPoE_MMTRSSM:
    """``hs`` / ``ls`` = (class_size, category_size) of ``h_dist`` / ``l_dist`` (mmtrssm yaml 138-147)."""
    hs_dim, ls_dim = hs[0] * hs[1], ls[0] * ls[1]
    rep = {"deterministic_size": ld, "hidden_size": hidden, "obs_embed_size": embed,
           "distribution_config": [ls[0], ls[1]], "activation_name": activation}
    act = getattr(nn, activation)
    return MoPoE_MMTRSSM(
        audio_representation=Representation(**rep), vision_representation=Representation(**rep),
        audio_encoder=Encoder(enc_audio), vision_encoder=Encoder(enc_vision),
        audio_decoder=Decoder(dec_audio), vision_decoder=Decoder(dec_vision),
        init_proj=MLP(in_features=embed, out_features=hd + ld, num_cells=init_cells, depth=1),
        kl_coeff=kl_coeff, use_kl_balancing=use_kl_balancing,
        action_size=action, hd_dim=hd, hs_dim=hs_dim, ld_dim=ld, ls_dim=ls_dim, l_tau=l_tau, h_tau=h_tau,
        l_prior=MLP(ld, ls_dim, hidden, 1, act), l_posterior=MLP(ld + embed, ls_dim, hidden, 1, act),
        h_prior=MLP(hd, hs_dim, hidden, 1, act), h_posterior=MLP(ld + hd, hs_dim, hidden, 1, act),
        l_dist=MultiOneHotFactory(class_size=ls[0], category_size=ls[1]),
        h_dist=MultiOneHotFactory(class_size=hs[0], category_size=hs[1]),
        w_kl_h=w_kl_h,
    )
