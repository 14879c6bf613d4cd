"""CPU: the oracle restatement (oracle/ref_model.py) against the committed golden vectors.

The fixtures were produced in the build container by ``oracle/gen_golden.py`` from the reference's OWN
control flow (``/root/reference/src/.../core.py`` etc. over build-defined stand-ins for the absent
third-party packages) -- this test pins the restatement to them wherever it runs.
Tolerances: losses 2e-6 rel; deter / logits / probs 1e-6; one-hot indices exact; gradients 1e-4 rel.
"""

from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle.cases import CASES, GOLDEN_CASES, build_model, min_margin
from oracle.ref_model import cat_probs
from tests.conftest import check_weight_sums, golden_batch, golden_noise, load_golden


def _index(stoch: torch.Tensor, cats: int, classes: int) -> np.ndarray:
    return stoch.detach().reshape(*stoch.shape[:-1], cats, classes).argmax(-1).numpy().astype(np.int8)


@pytest.mark.parametrize("name", list(GOLDEN_CASES))
def test_restatement_matches_golden(name: str) -> None:
    torch.set_num_threads(1)
    case = CASES[name]
    fx = load_golden(name)
    model = build_model(case)
    check_weight_sums(model, fx)
    batch, noise = golden_batch(fx), golden_noise(fx)
    out = model.shared_step(batch, noise)
    out["loss"].backward()
    assert min_margin(case, out, noise) >= 1e-3  # every draw is far from a CDF edge: samples are stable
    for k in [k for k in fx if k.startswith("loss/")]:
        np.testing.assert_allclose(float(out[k[5:]]), float(fx[k]), rtol=2e-6, err_msg=k)
    d = case.dims
    if case.kind == "mrssm":
        for k in ("deter", "prior_logits", "audio_logits", "vision_logits", "post_logits"):
            np.testing.assert_allclose(out[f"_{k}"].detach().numpy(), fx[f"out/{k}"], rtol=1e-6, atol=1e-6, err_msg=k)
        _, q = cat_probs(out["_post_logits"].detach(), d.cats, d.classes)
        np.testing.assert_allclose(q.numpy(), fx["out/post_probs"], atol=1e-6)
        assert (_index(out["_post_stoch"], d.cats, d.classes) == fx["out/post_index"]).all()
        assert (_index(out["_prior_stoch"], d.cats, d.classes) == fx["out/prior_index"]).all()
        assert (_index(out["_stoch0"], d.cats, d.classes) == fx["out/stoch0_index"]).all()
        q0 = case.query
        with torch.no_grad():
            tr = model.rollout_transition(
                batch[0][:, q0:], {"deter": out["_deter"][:, q0 - 1], "stoch": out["_post_stoch"][:, q0 - 1]}, noise["u_trans"])
        np.testing.assert_allclose(tr["deter"].numpy(), fx["trans/deter"], rtol=1e-6, atol=1e-6)
        assert (_index(tr["prior_stoch"], d.cats, d.classes) == fx["trans/index"]).all()
    else:
        for k in ("deter_l", "deter_h", "hidden_l", "hidden_h", "prior_logits_l", "prior_logits_h", "audio_logits",
                  "vision_logits", "post_logits_l", "post_logits_h"):
            np.testing.assert_allclose(out[f"_{k}"].detach().numpy(), fx[f"out/{k}"], rtol=1e-6, atol=1e-6, err_msg=k)
        assert (_index(out["_post_stoch_l"], d.ls_cats, d.ls_classes) == fx["out/post_index_l"]).all()
        assert (_index(out["_post_stoch_h"], d.hs_cats, d.hs_classes) == fx["out/post_index_h"]).all()
        assert (_index(out["_prior_stoch_l"], d.ls_cats, d.ls_classes) == fx["out/prior_index_l"]).all()
        assert (_index(out["_prior_stoch_h"], d.hs_cats, d.hs_classes) == fx["out/prior_index_h"]).all()
        # the parameters the reference never trains on this path (SURVEY.md section 2 "Hazard")
        dead = {k for k, p in model.named_parameters() if p.grad is None}
        assert dead == set(fx["meta/no_grad_params"].tolist())
    grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    for k in [k for k in fx if k.startswith("gradnorm/")]:
        g = grads[k[9:]].double()
        np.testing.assert_allclose(g.norm().item(), fx[k][0], rtol=1e-4, atol=1e-9, err_msg=k)
    for k in [k for k in fx if k.startswith("grad/")]:
        g = grads[k[5:]].flatten()
        stride = max(1, g.numel() // 4096)
        scale = float(np.abs(fx[k]).max()) + 1e-12
        np.testing.assert_allclose(g[::stride].numpy(), fx[k], rtol=1e-4, atol=1e-5 * scale, err_msg=k)
