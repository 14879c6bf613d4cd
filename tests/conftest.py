"""Shared test plumbing.  ``-m "not gpu"`` runs here on CPU; ``-m gpu`` runs on an MI355X box."""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config: pytest.Config) -> None:
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name: str) -> dict[str, np.ndarray]:
    with np.load(GOLDEN / f"{name}.npz", allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_batch(fx: dict[str, np.ndarray], device: str = "cpu") -> tuple[torch.Tensor, ...]:
    names = ("action_in", "audio_in", "vision_in", "action_tgt", "audio_tgt", "vision_tgt")
    return tuple(torch.from_numpy(fx[f"batch/{n}"]).to(device) for n in names)


def golden_noise(fx: dict[str, np.ndarray], device: str = "cpu") -> dict[str, torch.Tensor]:
    return {k[len("noise/"):]: torch.from_numpy(v).to(device) for k, v in fx.items() if k.startswith("noise/")}


def check_weight_sums(model: torch.nn.Module, fx: dict[str, np.ndarray]) -> None:
    """The seeded weights re-created here must be the ones the fixture was generated with."""
    for k, p in model.state_dict().items():
        want = fx[f"wsum/{k}"]
        got = np.asarray([p.double().sum().item(), p.double().abs().sum().item()])
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12, err_msg=f"weights differ from fixture: {k}")


def product_from_case(case, oracle_model: torch.nn.Module, device: str):  # noqa: ANN001, ANN201
    """The product model for a parity case, with the oracle's weights copied BY STATE-DICT NAME."""
    import multimodal_mtrssm_amd as mt

    d = case.dims
    if case.kind == "mrssm":
        model = mt.make_mrssm(
            deter=d.deter, hidden=d.hidden, classes=d.classes, cats=d.cats, action=d.action, embed=d.embed,
            enc_audio=d.enc_audio, enc_vision=d.enc_vision, dec_audio=d.dec_audio, dec_vision=d.dec_vision,
            activation=d.activation, init_cells=d.init_cells, kl_coeff=d.kl_coeff, use_kl_balancing=d.use_kl_balancing)
    else:
        model = mt.make_mmtrssm(
            hd=d.hd, hs=(d.hs_classes, d.hs_cats), ld=d.ld, ls=(d.ls_classes, d.ls_cats), hidden=d.hidden, action=d.action,
            embed=d.embed, enc_audio=d.enc_audio, enc_vision=d.enc_vision, dec_audio=d.dec_audio, dec_vision=d.dec_vision,
            l_tau=d.l_tau, h_tau=d.h_tau, activation=d.activation, init_cells=d.init_cells, kl_coeff=d.kl_coeff,
            w_kl_h=d.w_kl_h, use_kl_balancing=d.use_kl_balancing)
    result = model.load_state_dict(oracle_model.state_dict(), strict=True)
    assert not result.missing_keys and not result.unexpected_keys
    return model.to(device)
