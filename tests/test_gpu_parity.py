"""GPU (MI355X): the HIP path, called through the C-ABI, against the oracle and the golden vectors.

Tolerances (fp32 path; north_star: losses 1e-4 rel, probabilities 1e-5):
  loss / recon / kl           2e-5 relative vs golden (reference-generated)
  deter, logits, probs        1e-5 absolute
  one-hot samples             exact (fixture draws keep a 1e-3 margin from every CDF edge)
  gradients                   2e-4 relative to the tensor's largest entry
The conv encoders / decoders are build-defined (``cnn`` is absent upstream): their parity is
HIP-build vs ``oracle/ref_cnn.py``, "parity unpinned" upstream.
"""

from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle.cases import CASES, GOLDEN_CASES, build_batch, build_model, build_noise, screened_noise, with_sizes
from oracle.ref_model import cat_probs
from tests.conftest import check_weight_sums, golden_batch, golden_noise, load_golden, product_from_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _np(t: torch.Tensor) -> np.ndarray:
    return t.detach().float().cpu().numpy()


def _index(stoch: torch.Tensor, cats: int, classes: int) -> np.ndarray:
    return _np(stoch).reshape(*stoch.shape[:-1], cats, classes).argmax(-1).astype(np.int8)


def _assert_onehot(stoch: torch.Tensor, cats: int, classes: int) -> None:
    s = _np(stoch).reshape(*stoch.shape[:-1], cats, classes)
    assert ((s == 0) | (s == 1)).all()
    assert (s.sum(-1) == 1).all()


def _to(noise: dict[str, torch.Tensor], device: str) -> dict[str, torch.Tensor]:
    return {k: v.to(device) for k, v in noise.items()}


@pytest.fixture(scope="module")
def lib_loaded() -> None:
    import multimodal_mtrssm_amd as mt

    assert torch.cuda.is_available(), "these tests need the MI355X"
    assert mt._lib.load().mtrssm_version() == 100  # noqa: SLF001


# ---------------------------------------------------------------------------------------------
# shared_step: losses + gradients
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", list(GOLDEN_CASES))
def test_shared_step_matches_golden_and_oracle(name: str, lib_loaded: None) -> None:
    case = CASES[name]
    fx = load_golden(name)
    oracle = build_model(case)
    check_weight_sums(oracle, fx)
    batch, noise = golden_batch(fx), golden_noise(fx)
    ref = oracle.shared_step(batch, noise)
    ref["loss"].backward()
    model = product_from_case(case, oracle, DEV)
    out = model.shared_step(tuple(b.to(DEV) for b in batch), _to(noise, DEV))
    out["loss"].backward()
    torch.cuda.synchronize()
    keys = [k[5:] for k in fx if k.startswith("loss/")]
    assert set(out) == set(keys)
    for k in keys:
        np.testing.assert_allclose(float(out[k]), float(fx[f"loss/{k}"]), rtol=2e-5, err_msg=f"{k} vs golden")
        np.testing.assert_allclose(float(out[k]), float(ref[k]), rtol=2e-5, err_msg=f"{k} vs oracle")
    ref_grads = {k: p.grad for k, p in oracle.named_parameters() if p.grad is not None}
    got = dict(model.named_parameters())
    for k, g in ref_grads.items():
        assert got[k].grad is not None, k
        scale = float(g.abs().max()) + 1e-12
        np.testing.assert_allclose(_np(got[k].grad), g.numpy(), rtol=2e-4, atol=2e-4 * scale, err_msg=f"grad {k}")
    # golden gradient samples (reference-generated)
    for k in [k for k in fx if k.startswith("grad/")]:
        g = got[k[5:]].grad.flatten()
        stride = max(1, g.numel() // 4096)
        scale = float(np.abs(fx[k]).max()) + 1e-12
        np.testing.assert_allclose(_np(g[::stride]), fx[k], rtol=2e-4, atol=2e-4 * scale, err_msg=k)
    if case.kind == "mmtrssm":  # parameters the reference never trains stay untouched (zero or no gradient)
        for k in fx["meta/no_grad_params"].tolist():
            assert got[k].grad is None or float(got[k].grad.abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------
# rollout_representation / rollout_transition / State API
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", [n for n in GOLDEN_CASES if CASES[n].kind == "mrssm"])
def test_mrssm_rollout_matches_golden(name: str, lib_loaded: None) -> None:
    import multimodal_mtrssm_amd as mt

    case = CASES[name]
    d = case.dims
    fx = load_golden(name)
    model = product_from_case(case, build_model(case), DEV)
    batch = tuple(b.to(DEV) for b in golden_batch(fx))
    noise = _to(golden_noise(fx), DEV)
    with torch.no_grad():
        state0 = model.initial_state((batch[1][:, 0], batch[2][:, 0]), noise)
        post, prior = model.rollout_representation(actions=batch[0], observations=(batch[1], batch[2]), prev_state=state0,
                                                   noise=noise)
        np.testing.assert_allclose(_np(state0.deter), fx["out/deter0"], atol=1e-5)
        assert (_index(state0.stoch, d.cats, d.classes) == fx["out/stoch0_index"]).all()
        np.testing.assert_allclose(_np(post.deter), fx["out/deter"], atol=1e-5)
        np.testing.assert_allclose(_np(post.distribution.probs), fx["out/post_probs"], atol=1e-5)
        np.testing.assert_allclose(_np(prior.distribution.probs), fx["out/prior_probs"], atol=1e-5)
        assert (_index(post.stoch, d.cats, d.classes) == fx["out/post_index"]).all()
        assert (_index(prior.stoch, d.cats, d.classes) == fx["out/prior_index"]).all()
        _assert_onehot(post.stoch, d.cats, d.classes)
        assert post.feature.shape == (case.batch, case.steps, d.deter + d.stoch)
        # callbacks' use of the API (mrssm/callback.py:156-189)
        q = case.query
        start = post[:, q - 1]
        trans = model.rollout_transition(actions=batch[0][:, q:], prev_state=start, noise={"u_prior": noise["u_trans"][:, : case.steps - q]})
        np.testing.assert_allclose(_np(trans.deter), fx["trans/deter"], atol=1e-5)
        np.testing.assert_allclose(_np(trans.distribution.probs), fx["trans/prior_probs"], atol=1e-5)
        assert (_index(trans.stoch, d.cats, d.classes) == fx["trans/index"]).all()
        joined = mt.cat_states([post[:, :q], trans], dim=1)
        assert joined.deter.shape == post.deter.shape
        recon = model.decode_state(joined)
        assert recon["recon/vision"].shape == batch[5].shape
        # KL from the distribution objects == KL from the kernel
        kl_obj = mt.kl_divergence(post.distribution.independent(1), prior.distribution.independent(1), use_balancing=True)
        np.testing.assert_allclose(float(kl_obj), float(post.kl_per_step.mean()), rtol=1e-5)
        # single prior step through Transition.forward (networks.py:151-173) == first step of the prior-only scan
        step = model.transition.forward(batch[0][:, q], start)
        np.testing.assert_allclose(_np(step.deter), fx["trans/deter"][:, 0], atol=1e-5)
        np.testing.assert_allclose(_np(step.distribution.probs), fx["trans/prior_probs"][:, 0], atol=1e-5)
    with pytest.raises(TypeError):
        model.rollout_representation(actions=batch[0], observations=batch[1], prev_state=state0)


@pytest.mark.parametrize("name", [n for n in GOLDEN_CASES if CASES[n].kind == "mmtrssm"])
def test_mmtrssm_rollout_matches_golden(name: str, lib_loaded: None) -> None:
    import multimodal_mtrssm_amd as mt

    case = CASES[name]
    d = case.dims
    fx = load_golden(name)
    model = product_from_case(case, build_model(case), DEV)
    batch = tuple(b.to(DEV) for b in golden_batch(fx))
    noise = _to(golden_noise(fx), DEV)
    with torch.no_grad():
        state0 = model.initial_state((batch[1][:, 0], batch[2][:, 0]), noise)
        np.testing.assert_allclose(_np(state0.deter_h), fx["out/init_deter_h"], atol=1e-5)
        np.testing.assert_allclose(_np(state0.deter_l), fx["out/init_deter_l"], atol=1e-5)
        assert (_index(state0.stoch_h, d.hs_cats, d.hs_classes) == fx["out/init_index_h"]).all()
        assert (_index(state0.stoch_l, d.ls_cats, d.ls_classes) == fx["out/init_index_l"]).all()
        post, prior = model.rollout_representation(actions=batch[0], observations=(batch[1], batch[2]), prev_state=state0,
                                                   noise=noise)
        for k in ("deter_l", "deter_h", "hidden_l", "hidden_h"):
            np.testing.assert_allclose(_np(getattr(post, k)), fx[f"out/{k}"], atol=1e-5, err_msg=k)
        np.testing.assert_allclose(_np(post.distribution_l.probs), fx["out/post_probs_l"], atol=1e-5)
        np.testing.assert_allclose(_np(post.distribution_h.probs), fx["out/post_probs_h"], atol=1e-5)
        np.testing.assert_allclose(_np(prior.distribution_l.probs), fx["out/prior_probs_l"], atol=1e-5)
        np.testing.assert_allclose(_np(prior.distribution_h.probs), fx["out/prior_probs_h"], atol=1e-5)
        assert (_index(post.stoch_l, d.ls_cats, d.ls_classes) == fx["out/post_index_l"]).all()
        assert (_index(post.stoch_h, d.hs_cats, d.hs_classes) == fx["out/post_index_h"]).all()
        assert (_index(prior.stoch_l, d.ls_cats, d.ls_classes) == fx["out/prior_index_l"]).all()
        assert (_index(prior.stoch_h, d.hs_cats, d.hs_classes) == fx["out/prior_index_h"]).all()
        assert post.feature.shape == (case.batch, case.steps, d.feature)
        q = case.query
        n = case.steps - q
        trans = model.rollout_transition(actions=batch[0][:, q:], prev_state=post[:, q - 1],
                                         noise={"u_prior_h": noise["u_trans_h"][:, :n], "u_prior_l": noise["u_trans_l"][:, :n]})
        np.testing.assert_allclose(_np(trans.deter_l), fx["trans/deter_l"], atol=1e-5)
        np.testing.assert_allclose(_np(trans.deter_h), fx["trans/deter_h"], atol=1e-5)
        assert (_index(trans.stoch_l, d.ls_cats, d.ls_classes) == fx["trans/index_l"]).all()
        assert (_index(trans.stoch_h, d.hs_cats, d.hs_classes) == fx["trans/index_h"]).all()
        joined = mt.cat_mtstates([post[:, :q], trans], dim=1)
        assert joined.feature.shape == post.feature.shape


@pytest.mark.parametrize("name", list(GOLDEN_CASES))
def test_prior_rollout_is_differentiable_and_matches_the_oracle(name: str, lib_loaded: None) -> None:
    """``rollout_transition`` with autograd on (reference ``core.py:170-185`` / mmtrssm ``core.py:496-544`` are plain differentiable
    loops): values equal the fused inference kernel's, and the gradients of a fixed random functional of every output -- with
    respect to the parameters, the actions and the start state -- equal the oracle's autograd on the host."""
    from multimodal_mtrssm_amd import scan

    case = CASES[name]
    d = case.dims
    fx = load_golden(name)
    oracle = build_model(case)
    model = product_from_case(case, oracle, DEV)
    batch, noise = golden_batch(fx), golden_noise(fx)
    q = case.query
    n = case.steps - q
    gen = torch.Generator().manual_seed(5)
    actions = batch[0][:, q:].clone()
    mr = case.kind == "mrssm"
    with torch.no_grad():
        if mr:
            s0 = oracle.initial_state(batch[1][:, 0], batch[2][:, 0], noise["u_init"])
            state0 = {"deter": s0["deter"], "stoch": s0["stoch"]}
            u = {"u_prior": noise["u_trans"][:, :n]}
        else:
            s0 = oracle.initial_state(batch[1][:, 0], batch[2][:, 0], noise["u_init_h"], noise["u_init_l"])
            state0 = {k: s0[k] for k in ("deter_l", "deter_h", "hidden_l", "hidden_h", "stoch_l", "stoch_h")}
            u = {"u_prior_h": noise["u_trans_h"][:, :n], "u_prior_l": noise["u_trans_l"][:, :n]}

    def run(on_gpu: bool):  # noqa: ANN202
        dev = DEV if on_gpu else "cpu"
        a = actions.detach().clone().to(dev).requires_grad_(True)
        st = {k: v.detach().clone().to(dev).requires_grad_(not k.startswith("stoch")) for k, v in state0.items()}
        un = {k: v.to(dev) for k, v in u.items()}
        if on_gpu:
            out = (scan.mrssm_prior_rollout(model.transition, a, st["deter"], st["stoch"], un["u_prior"]) if mr
                   else scan.mmtrssm_prior_rollout(model, a, st, un))
            with torch.no_grad():  # the fused inference kernel on the same inputs
                fused = (scan.mrssm_prior_rollout(model.transition, a.detach(), st["deter"].detach(), st["stoch"], un["u_prior"]) if mr
                         else scan.mmtrssm_prior_rollout(model, a.detach(), {k: v.detach() for k, v in st.items()}, un))
            for k, v in fused.items():
                np.testing.assert_allclose(_np(out[k]), _np(v), atol=1e-5, err_msg=f"composed vs fused {k}")
        else:
            out = oracle.rollout_transition(a, st, un["u_prior"]) if mr else oracle.rollout_transition(a, st, un)
        gen.manual_seed(5)
        loss = sum((v * torch.randn(v.shape, generator=gen).to(dev)).sum() for _, v in sorted(out.items()))
        for p_ in (model.parameters() if on_gpu else oracle.parameters()):
            p_.grad = None
        loss.backward()
        return out, a.grad, {k: v.grad for k, v in st.items() if v.requires_grad}

    ref_out, ref_ga, ref_gs = run(False)
    out, ga, gs = run(True)
    torch.cuda.synchronize()
    for k in ref_out:
        np.testing.assert_allclose(_np(out[k]), ref_out[k].detach().numpy(), atol=1e-5, err_msg=k)
    scale = float(ref_ga.abs().max()) + 1e-12
    np.testing.assert_allclose(_np(ga), ref_ga.numpy(), rtol=2e-4, atol=2e-4 * scale, err_msg="d actions")
    for k, g in ref_gs.items():
        scale = float(g.abs().max()) + 1e-12
        np.testing.assert_allclose(_np(gs[k]), g.numpy(), rtol=2e-4, atol=2e-4 * scale, err_msg=f"d {k}")
    got = dict(model.named_parameters())
    seen = 0
    for k, p_ in oracle.named_parameters():
        if p_.grad is None:
            continue
        assert got[k].grad is not None, k
        scale = float(p_.grad.abs().max()) + 1e-12
        np.testing.assert_allclose(_np(got[k].grad), p_.grad.numpy(), rtol=2e-4, atol=2e-4 * scale, err_msg=f"grad {k}")
        seen += 1
    assert seen >= 6  # the prior path's Linear layers (and nothing of the posterior / encoders)


# ---------------------------------------------------------------------------------------------
# BASELINE "Large" core dims: the > 64 KiB dynamic-LDS path of the scan kernels (GPU vs oracle, no fixture)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["mrssm_large", "mmtrssm_large"])
def test_large_dims_match_oracle(name: str, lib_loaded: None) -> None:
    torch.set_num_threads(8)
    case = CASES[name]
    oracle = build_model(case)
    batch, noise = build_batch(case), build_noise(case)
    ref = oracle.shared_step(batch, noise)
    ref["loss"].backward()
    model = product_from_case(case, oracle, DEV)
    out = model.shared_step(tuple(b.to(DEV) for b in batch), _to(noise, DEV))
    out["loss"].backward()
    for k in out:
        np.testing.assert_allclose(float(out[k]), float(ref[k]), rtol=1e-4, err_msg=k)
    names = ("transition.rnn_cell.weight_hh", "representation.rnn_to_post_projector.0.weight") if case.kind == "mrssm" else (
        "l_rnn._d2h.weight", "h_posterior.0.weight")
    got = dict(model.named_parameters())
    want = dict(oracle.named_parameters())
    for k in names:
        g = want[k].grad
        np.testing.assert_allclose(_np(got[k].grad), g.numpy(), rtol=1e-3, atol=1e-3 * float(g.abs().max()), err_msg=k)


# ---------------------------------------------------------------------------------------------
# cluster scan (one row on four CUs, weights resident) against the single-CU scan
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize(("name", "batch", "steps"), [("mrssm_default", 5, 9), ("mrssm_cfg2dims", 70, 6), ("mrssm_bench", 3, 50)])
def test_cluster_scan_matches_single_cu_scan(name: str, batch: int, steps: int, lib_loaded: None) -> None:
    """csrc/mrssm_cluster.hip (default for D = H in {32, 64, 128, 200}) vs csrc/mrssm_scan.hip on the same inputs: same samples,
    deter / logits to 2e-6 (the fused W_ih W2 product rounds differently), same losses to 1e-6, gradients to 1e-5 of the
    tensor's max; more rows than clusters (70 > 64: a cluster walks two rows) and ragged cluster groups (5, 3 rows)."""
    from multimodal_mtrssm_amd import scan

    case = with_sizes(CASES[name], batch, steps)
    oracle = build_model(case)
    batch_t = tuple(b.to(DEV) for b in build_batch(case))
    noise = _to(build_noise(case), DEV)
    runs = {}
    for cluster in (False, True):
        scan.CLUSTER_SCAN = cluster
        try:
            model = product_from_case(case, oracle, DEV)
            with torch.no_grad():
                state0 = model.initial_state((batch_t[1][:, 0], batch_t[2][:, 0]), noise)
                post, prior = model.rollout_representation(actions=batch_t[0], observations=(batch_t[1], batch_t[2]), prev_state=state0,
                                                           noise=noise)
            out = model.shared_step(batch_t, noise)
            out["loss"].backward()
            scan.check_cluster_status()
            runs[cluster] = (post, prior, {k: float(v) for k, v in out.items()},
                             {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
        finally:
            scan.CLUSTER_SCAN = True
    (p0, q0, l0, g0), (p1, q1, l1, g1) = runs[False], runs[True]
    assert torch.equal(p0.stoch, p1.stoch) and torch.equal(q0.stoch, q1.stoch)
    np.testing.assert_allclose(_np(p1.deter), _np(p0.deter), atol=2e-6)
    np.testing.assert_allclose(_np(p1.distribution.probs), _np(p0.distribution.probs), atol=2e-6)
    np.testing.assert_allclose(_np(q1.distribution.probs), _np(q0.distribution.probs), atol=2e-6)
    np.testing.assert_allclose(_np(p1.kl_per_step), _np(p0.kl_per_step), rtol=1e-4, atol=1e-6)
    for k in l0:
        np.testing.assert_allclose(l1[k], l0[k], rtol=2e-6, err_msg=k)
    for k, g in g0.items():
        np.testing.assert_allclose(_np(g1[k]), _np(g), rtol=1e-4, atol=1e-5 * (float(g.abs().max()) + 1e-9), err_msg=k)


# ---------------------------------------------------------------------------------------------
# wide scan (all CUs on one tile of 32 batch rows, MFMA products, grid barriers) against the single-CU scan
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize(("pieces", "batch", "steps"), [(3, 3, 4), (3, 37, 3), (2, 5, 6)])
def test_wide_scan_matches_single_cu_scan(pieces: int, batch: int, steps: int, lib_loaded: None) -> None:
    """csrc/mrssm_wide.hip (default for D or H >= 256: BASELINE configs[4] dims, D = H = 1024, S = 16 x 8) vs csrc/mrssm_scan.hip on
    the same inputs, forward and BPTT: same samples, deter / probabilities to 5e-6 (three bf16 pieces per operand; the fused
    W_ih W2 product rounds differently), losses to 5e-6, gradients to 2e-5 of the tensor's max (two pieces: 16 significant bits
    per operand, tolerances x 20).  37 rows = two row tiles, the second one ragged; 3 and 5 rows = a partly filled MFMA tile."""
    from multimodal_mtrssm_amd import scan

    case = with_sizes(CASES["mrssm_large"], batch, steps)
    oracle = build_model(case)
    batch_t = tuple(b.to(DEV) for b in build_batch(case))
    noise = _to(build_noise(case), DEV)
    runs = {}
    saved = (scan.WIDE_SCAN, scan.WIDE_PIECES)
    for wide in (False, True):
        scan.WIDE_SCAN, scan.WIDE_PIECES = wide, pieces
        try:
            model = product_from_case(case, oracle, DEV)
            with torch.no_grad():
                state0 = model.initial_state((batch_t[1][:, 0], batch_t[2][:, 0]), noise)
                post, prior = model.rollout_representation(actions=batch_t[0], observations=(batch_t[1], batch_t[2]), prev_state=state0,
                                                           noise=noise)
            out = model.shared_step(batch_t, noise)
            out["loss"].backward()
            scan.check_cluster_status()
            runs[wide] = (post, prior, {k: float(v) for k, v in out.items()},
                          {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
        finally:
            scan.WIDE_SCAN, scan.WIDE_PIECES = saved
    (p0, q0, l0, g0), (p1, q1, l1, g1) = runs[False], runs[True]
    loose = 1.0 if pieces == 3 else 20.0
    assert torch.equal(p0.stoch, p1.stoch) and torch.equal(q0.stoch, q1.stoch)
    np.testing.assert_allclose(_np(p1.deter), _np(p0.deter), atol=5e-6 * loose)
    np.testing.assert_allclose(_np(p1.distribution.probs), _np(p0.distribution.probs), atol=5e-6 * loose)
    np.testing.assert_allclose(_np(q1.distribution.probs), _np(q0.distribution.probs), atol=5e-6 * loose)
    np.testing.assert_allclose(_np(p1.kl_per_step), _np(p0.kl_per_step), rtol=1e-4 * loose, atol=2e-6 * loose)
    for k in l0:
        np.testing.assert_allclose(l1[k], l0[k], rtol=5e-6 * loose, err_msg=k)
    for k, g in g0.items():
        np.testing.assert_allclose(_np(g1[k]), _np(g), rtol=1e-3, atol=2e-5 * loose * (float(g.abs().max()) + 1e-9), err_msg=k)


@pytest.mark.parametrize(("pieces", "batch", "steps"), [(3, 5, 6), (3, 70, 3), (2, 9, 4)])
def test_wide_mmtrssm_scan_matches_single_cu_scan(pieces: int, batch: int, steps: int, lib_loaded: None) -> None:
    """csrc/mmtrssm_wide.hip (default for ld or hd >= 128: BASELINE configs[2] dims, ld = hd = H = 200, 6 x 5 categoricals on both
    levels) vs csrc/mmtrssm_scan.hip on the same inputs, forward and BPTT: same samples on both levels, deter / hidden /
    probabilities to 5e-6, losses to 5e-6, gradients to 2e-5 of the tensor's max (two pieces: x 20).  70 rows = two passes of 64
    rows, the second one ragged; 200 is not a multiple of 16: the last column tile of every product is half empty."""
    from multimodal_mtrssm_amd import scan

    case = with_sizes(CASES["mmtrssm_cfg3dims"], batch, steps)
    oracle = build_model(case)
    batch_t = tuple(b.to(DEV) for b in build_batch(case))
    noise = _to(build_noise(case), DEV)
    runs = {}
    saved = (scan.WIDE_SCAN, scan.WIDE_PIECES)
    for wide in (False, True):
        scan.WIDE_SCAN, scan.WIDE_PIECES = wide, pieces
        try:
            model = product_from_case(case, oracle, DEV)
            with torch.no_grad():
                state0 = model.initial_state((batch_t[1][:, 0], batch_t[2][:, 0]), noise)
                post, prior = model.rollout_representation(actions=batch_t[0], observations=(batch_t[1], batch_t[2]), prev_state=state0,
                                                           noise=noise)
            out = model.shared_step(batch_t, noise)
            out["loss"].backward()
            scan.check_cluster_status()
            runs[wide] = (post, prior, {k: float(v) for k, v in out.items()},
                          {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
        finally:
            scan.WIDE_SCAN, scan.WIDE_PIECES = saved
    (p0, q0, l0, g0), (p1, q1, l1, g1) = runs[False], runs[True]
    loose = 1.0 if pieces == 3 else 20.0
    for k in ("stoch_l", "stoch_h"):
        assert torch.equal(getattr(p0, k), getattr(p1, k)) and torch.equal(getattr(q0, k), getattr(q1, k)), k
    for k in ("deter_l", "deter_h", "hidden_l", "hidden_h"):
        np.testing.assert_allclose(_np(getattr(p1, k)), _np(getattr(p0, k)), atol=5e-6 * loose, err_msg=k)
    for a, b in ((p1.distribution_l, p0.distribution_l), (p1.distribution_h, p0.distribution_h), (q1.distribution_l, q0.distribution_l),
                 (q1.distribution_h, q0.distribution_h)):
        np.testing.assert_allclose(_np(a.probs), _np(b.probs), atol=5e-6 * loose)
    np.testing.assert_allclose(_np(p1.kl_per_step), _np(p0.kl_per_step), rtol=1e-4 * loose, atol=2e-6 * loose)
    np.testing.assert_allclose(_np(p1.kl_h_per_step), _np(p0.kl_h_per_step), rtol=1e-4 * loose, atol=2e-6 * loose)
    for k in l0:
        np.testing.assert_allclose(l1[k], l0[k], rtol=5e-6 * loose, err_msg=k)
    assert set(g0) == set(g1)
    for k, g in g0.items():
        np.testing.assert_allclose(_np(g1[k]), _np(g), rtol=1e-3, atol=2e-5 * loose * (float(g.abs().max()) + 1e-9), err_msg=k)


# ---------------------------------------------------------------------------------------------
# row-tile variants and ragged batches: every tiling gives the same answer
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["mrssm_nonsquare", "mmtrssm_default"])
@pytest.mark.parametrize(("rows", "threads"), [(2, 256), (4, 256), (1, 512), (2, 128)])
def test_row_tiles_agree(name: str, rows: int, threads: int, lib_loaded: None) -> None:
    case = with_sizes(CASES[name], 5, 6)  # 5 rows: ragged for 2- and 4-row tiles
    oracle = build_model(case)
    batch = tuple(b.to(DEV) for b in build_batch(case))
    noise = _to(build_noise(case), DEV)
    model = product_from_case(case, oracle, DEV)
    base = model.shared_step(batch, noise)
    base["loss"].backward()
    g0 = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    model.zero_grad()
    model.scan_rows_per_block, model.scan_threads = rows, threads
    out = model.shared_step(batch, noise)
    out["loss"].backward()
    for k in base:
        np.testing.assert_allclose(float(out[k]), float(base[k]), rtol=1e-6, err_msg=k)
    for k, p in model.named_parameters():
        if p.grad is not None:
            # conv weight gradients are summed with fp32 atomics: two runs differ by their arrival order (~3e-6 of the max)
            np.testing.assert_allclose(_np(p.grad), _np(g0[k]), rtol=1e-4, atol=1e-5 * (float(g0[k].abs().max()) + 1e-9), err_msg=k)


# ---------------------------------------------------------------------------------------------
# BASELINE config-2 / config-3 sizes (B=64, T=50): size-independent properties
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["mrssm_cfg2dims", "mmtrssm_cfg3dims"])
def test_full_size_properties(name: str, lib_loaded: None) -> None:
    case = with_sizes(CASES[name], 64, 50)
    oracle = build_model(case)
    model = product_from_case(case, oracle, DEV)
    batch = tuple(b.to(DEV) for b in build_batch(case))
    noise = _to(build_noise(case), DEV)
    d = case.dims
    with torch.no_grad():
        state0 = model.initial_state((batch[1][:, 0], batch[2][:, 0]), noise)
        post, prior = model.rollout_representation(actions=batch[0], observations=(batch[1], batch[2]), prev_state=state0, noise=noise)
        # (1) rows are independent: a sub-batch reproduces its rows of the full batch bit for bit
        rows = slice(7, 19)
        sub_noise = {k: v[rows] for k, v in noise.items()}
        post_sub, _ = model.rollout_representation(actions=batch[0][rows], observations=(batch[1][rows], batch[2][rows]),
                                                   prev_state=state0[rows], noise=sub_noise)
        if case.kind == "mrssm":
            assert torch.equal(post_sub.deter, post.deter[rows])
            assert torch.equal(post_sub.stoch, post.stoch[rows])
            _assert_onehot(post.stoch, d.cats, d.classes)
            _assert_onehot(prior.stoch, d.cats, d.classes)
            probs = [post.distribution.probs, prior.distribution.probs]
            kls = [post.kl_per_step]
        else:
            assert torch.equal(post_sub.deter_l, post.deter_l[rows])
            assert torch.equal(post_sub.stoch_h, post.stoch_h[rows])
            _assert_onehot(post.stoch_l, d.ls_cats, d.ls_classes)
            _assert_onehot(post.stoch_h, d.hs_cats, d.hs_classes)
            probs = [post.distribution_l.probs, post.distribution_h.probs, prior.distribution_l.probs]
            kls = [post.kl_per_step, post.kl_h_per_step]
        # (2) every categorical is a distribution; (3) KL >= 0; (4) a prefix of the sequence is a shorter rollout
        for p in probs:
            np.testing.assert_allclose(_np(p.sum(-1)), 1.0, atol=1e-5)
        for kl in kls:
            assert float(kl.min()) > -1e-6
            assert torch.isfinite(kl).all()
        t_half = 20
        half_noise = {k: (v[:, :t_half] if v.dim() == 3 else v) for k, v in noise.items()}
        post_half, _ = model.rollout_representation(actions=batch[0][:, :t_half], observations=(batch[1][:, :t_half], batch[2][:, :t_half]),
                                                    prev_state=state0, noise=half_noise)
        if case.kind == "mrssm":
            assert torch.equal(post_half.deter, post.deter[:, :t_half])
        else:
            assert torch.equal(post_half.deter_h, post.deter_h[:, :t_half])
    # (5) full-size train step against the oracle on a row subset (the oracle finishes 8 rows in seconds)
    sub = slice(0, 8)
    sub_batch = tuple(b[sub] for b in batch)
    sub_noise = {k: v[sub] for k, v in noise.items()}
    out = model.shared_step(sub_batch, sub_noise)
    ref = oracle.shared_step(tuple(b.cpu() for b in sub_batch), {k: v.cpu() for k, v in sub_noise.items()})
    for k in out:
        np.testing.assert_allclose(float(out[k]), float(ref[k]), rtol=1e-4, err_msg=k)


# ---------------------------------------------------------------------------------------------
# the EXACT model bench.py times (BASELINE configs[1] / configs[2] at their stated frame sizes, T = 50)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["mrssm_bench", "mmtrssm_bench", "mrssm_large_bench"])
def test_bench_model_matches_oracle(name: str, lib_loaded: None) -> None:
    """bench.py's models -- configs[1] / configs[2] (1x128x32 + 1x64x64 frames, channels [8,16,32], 3 residual blocks, deter 200,
    stoch 6x5, T = 50) and the "Large" configs[4] model of `--model large` (deter = hidden = embed = 1024, stoch 16x8, T = 100) -- on
    two sequences against the oracle: losses <= 1e-4 relative (north_star), posterior / prior probabilities <= 1e-5, every
    one-hot sample exact, EVERY gradient <= 2e-4 of its tensor's largest entry.  The noise seed is screened so that no draw
    sits within 1e-4 of a CDF edge (two fp32 implementations differ by ~1e-6 there)."""
    torch.set_num_threads(8)
    case = CASES[name]
    oracle = build_model(case)
    batch = build_batch(case)
    # 16 categoricals x 100 steps: a seed that keeps 1e-4 from all ~20 000 CDF edges does not turn up in 40 tries; 2e-5 is
    # still 20 x the distance two fp32 implementations land apart
    want_margin = 2e-5 if "large" in name else 1e-4
    noise, margin, _seed = screened_noise(case, oracle, batch, margin=want_margin)
    assert margin >= want_margin, margin
    ref = oracle.shared_step(batch, noise)
    ref["loss"].backward()
    model = product_from_case(case, oracle, DEV)
    gbatch, gnoise = tuple(b.to(DEV) for b in batch), _to(noise, DEV)
    out = model.shared_step(gbatch, gnoise)
    out["loss"].backward()
    torch.cuda.synchronize()
    assert set(out) == {k for k in ref if not k.startswith("_")}
    for k in out:
        np.testing.assert_allclose(float(out[k]), float(ref[k]), rtol=1e-4, err_msg=k)
    want = {k: p.grad for k, p in oracle.named_parameters() if p.grad is not None}
    got = dict(model.named_parameters())
    assert len(want) > 90
    for k, g in want.items():
        scale = float(g.abs().max()) + 1e-12
        np.testing.assert_allclose(_np(got[k].grad), g.numpy(), rtol=2e-4, atol=2e-4 * scale, err_msg=f"grad {k}")
    d = case.dims
    with torch.no_grad():
        state0 = model.initial_state((gbatch[1][:, 0], gbatch[2][:, 0]), gnoise)
        post, prior = model.rollout_representation(actions=gbatch[0], observations=(gbatch[1], gbatch[2]), prev_state=state0, noise=gnoise)
    if case.kind == "mrssm":
        pairs = [(post.distribution.probs, ref["_post_logits"], post.stoch, ref["_post_stoch"], d.cats, d.classes),
                 (prior.distribution.probs, ref["_prior_logits"], prior.stoch, ref["_prior_stoch"], d.cats, d.classes)]
        np.testing.assert_allclose(_np(post.deter), ref["_deter"].detach().numpy(), atol=1e-5)
    else:
        pairs = [(post.distribution_l.probs, ref["_post_logits_l"], post.stoch_l, ref["_post_stoch_l"], d.ls_cats, d.ls_classes),
                 (post.distribution_h.probs, ref["_post_logits_h"], post.stoch_h, ref["_post_stoch_h"], d.hs_cats, d.hs_classes),
                 (prior.distribution_l.probs, ref["_prior_logits_l"], prior.stoch_l, ref["_prior_stoch_l"], d.ls_cats, d.ls_classes),
                 (prior.distribution_h.probs, ref["_prior_logits_h"], prior.stoch_h, ref["_prior_stoch_h"], d.hs_cats, d.hs_classes)]
        np.testing.assert_allclose(_np(post.deter_l), ref["_deter_l"].detach().numpy(), atol=1e-5)
        np.testing.assert_allclose(_np(post.deter_h), ref["_deter_h"].detach().numpy(), atol=1e-5)
    for probs, ref_logits, stoch, ref_stoch, cats, classes in pairs:
        _, want_p = cat_probs(ref_logits.detach(), cats, classes)
        np.testing.assert_allclose(_np(probs), want_p.numpy(), atol=1e-5)
        assert (_index(stoch, cats, classes) == _index(ref_stoch.detach(), cats, classes)).all()


@pytest.mark.parametrize("name", ["mrssm_bench", "mmtrssm_bench"])
def test_bench_model_full_batch_properties(name: str, lib_loaded: None) -> None:
    """The same model at the bench's batch (B = 64, T = 50, real frame sizes): a 12-row sub-batch reproduces its rows of the
    full batch bit for bit (forward AND the loss terms' per-row inputs), a prefix of the sequence is the shorter rollout, and
    the train step's loss equals the mean over two 32-row halves run with the global rows' noise -- what data-parallel
    sharding relies on."""
    case = with_sizes(CASES[name], 64, 50)
    model = product_from_case(case, build_model(case), DEV)
    batch = tuple(b.to(DEV) for b in build_batch(case))
    noise = _to(build_noise(case), DEV)
    with torch.no_grad():
        state0 = model.initial_state((batch[1][:, 0], batch[2][:, 0]), noise)
        post, _ = model.rollout_representation(actions=batch[0], observations=(batch[1], batch[2]), prev_state=state0, noise=noise)
        rows = slice(21, 33)
        sub_noise = {k: v[rows] for k, v in noise.items()}
        post_sub, _ = model.rollout_representation(actions=batch[0][rows], observations=(batch[1][rows], batch[2][rows]),
                                                   prev_state=state0[rows], noise=sub_noise)
        half_noise = {k: (v[:, :20] if v.dim() == 3 else v) for k, v in noise.items()}
        post_half, _ = model.rollout_representation(actions=batch[0][:, :20], observations=(batch[1][:, :20], batch[2][:, :20]),
                                                    prev_state=state0, noise=half_noise)
        a, b, c = (("deter", "stoch", "deter") if case.kind == "mrssm" else ("deter_l", "stoch_h", "deter_h"))
        assert torch.equal(getattr(post_sub, a), getattr(post, a)[rows])
        assert torch.equal(getattr(post_sub, b), getattr(post, b)[rows])
        assert torch.equal(getattr(post_half, c), getattr(post, c)[:, :20])
        full = model.shared_step(batch, noise)
        halves = [model.shared_step(tuple(x[h] for x in batch), {k: v[h] for k, v in noise.items()}) for h in (slice(0, 32), slice(32, 64))]
    for k in full:
        # equal up to the fp32 summation order of 13 M squared errors (atomics over workgroup partials): a few 1e-6
        np.testing.assert_allclose(float(full[k]), 0.5 * (float(halves[0][k]) + float(halves[1][k])), rtol=1e-5, err_msg=k)


# ---------------------------------------------------------------------------------------------
# conv kernels (MFMA implicit GEMM) vs torch CPU convolutions
# ---------------------------------------------------------------------------------------------
CONV_CASES = [
    # n, cin, h, w, cout, k, s, p, pre_act, coords
    (3, 1, 16, 8, 8, 3, 2, 1, False, True),     # first encoder layer: coord channels, thin
    (5, 8, 9, 7, 16, 3, 2, 1, True, False),     # odd plane, ragged pixel tile
    (2, 32, 8, 8, 64, 3, 1, 1, True, False),    # residual 3x3, Cout = 64 (two MFMA tiles)
    (2, 64, 8, 8, 128, 3, 1, 1, True, False),   # decoder residual 3x3, Cout = 128 (two workgroup rows)
    (4, 128, 4, 4, 64, 1, 1, 0, True, False),   # 1x1
    (1, 20, 5, 6, 40, 3, 1, 1, False, False),   # Cin, Cout not multiples of the tile sizes
    (3, 16, 16, 16, 32, 3, 2, 1, True, False),  # stride-2 patch, 8x8 output: one frame per 64-pixel group
    (2, 8, 32, 32, 16, 3, 2, 1, True, False),   # 16x16 output: four-row groups
    (2, 3, 64, 64, 8, 3, 2, 1, False, True),    # 32x32 output: two-row groups, coordinate channels
    (9, 32, 4, 4, 64, 3, 1, 1, True, False),    # 4x4 planes: four frames per group, ragged last group
    (3, 64, 16, 4, 64, 3, 1, 1, True, False),   # 16x4 audio plane
    (4, 1, 64, 64, 8, 3, 2, 1, False, True),    # first vision layer: thin gathered weight gradient (conv_weight_grad_thin_split_kernel), 27 columns
    (3, 1, 128, 32, 8, 3, 2, 1, False, True),   # ... first audio layer (64 x 16 output plane)
    (5, 2, 16, 16, 6, 3, 2, 1, True, False),    # ... with activation, 18 columns, 8 x 8 output plane
    (5, 24, 8, 8, 40, 3, 1, 1, False, False),   # 3x3 on 8-wide planes: register-direct weight gradient (conv3x3_weight_grad_split_kernel), ragged channels
    (11, 64, 2, 8, 32, 3, 1, 1, True, False),   # ... one k-step per frame (2 x 8 plane): every window row but two is outside
    (70, 16, 16, 4, 64, 3, 1, 1, True, False),  # ... 4-wide audio plane, many frames per workgroup slice
    (7, 64, 8, 8, 64, 1, 1, 0, True, False),    # 1x1 layers: register-direct weight gradient (conv1x1_weight_grad_split_kernel), 2 x 2 tiles
    (5, 24, 16, 4, 48, 1, 1, 0, False, False),  # ... ragged channel counts (1 x 2 tiles), no activation
    (3, 40, 4, 4, 20, 1, 1, 0, True, False),    # ... one-frame k-steps, 2 x 1 tiles
    (70, 16, 8, 8, 64, 1, 1, 0, True, False),   # ... more k-steps than one per workgroup slice boundary (280 k-steps)
    (6, 64, 8, 8, 128, 1, 1, 0, True, False),   # ... decoder residual 1x1 (64 -> 128): 4 x 2 tiles, 8 waves
    (3, 128, 4, 4, 100, 1, 1, 0, True, False),  # ... 4 x 4 tiles, 16 waves, ragged Cout
    (6, 128, 8, 8, 64, 1, 1, 0, True, False),   # 1x1, 128 -> 64: its backward-data (64 -> 128 with act') runs conv1x1_stream_kernel<64, 128, false>; (7, 64, 8, 8, 64, 1, ...) above <64, 64, false>
    (6, 64, 8, 8, 64, 3, 1, 1, True, False),    # residual 3x3 on 8x8 planes: weight-resident gather (conv3x3_resident_kernel<64, 2, 1>)
    (7, 64, 8, 8, 128, 3, 1, 1, True, False),   # ... decoder residual 64 -> 128 (<64, 4, 1>); its backward-data is <128, 2, 2> (K split over wave pairs)
    (5, 128, 8, 8, 64, 3, 1, 1, False, False),  # ... 128 -> 64 forward (<128, 2, 2>), backward-data <64, 4, 1>, no activation
    (10, 32, 8, 8, 64, 3, 1, 1, True, False),   # ... encoder 32 -> 64 (<32, 2, 1>)
    (6, 64, 16, 4, 64, 3, 1, 1, True, False),   # ... 16x4 audio plane (18 x 6 haloed image); (3, 64, 16, 4, ...) above is the same shape
    (5, 64, 4, 16, 128, 3, 1, 1, True, False),  # ... 4x16 plane
    (1300, 64, 8, 8, 64, 3, 1, 1, True, False), # ... more tiles than workgroups (650 on 256): the persistent loop, double-buffered images
    (7, 64, 8, 8, 64, 3, 1, 1, True, False),    # ... odd frame count: no whole tiles, the patch-staged kernel takes it
    (1, 64, 8, 8, 64, 3, 1, 1, True, False),    # staged-operand weight gradient (conv3x3_wgrad_resident_kernel): ONE frame, one workgroup
    (700, 64, 16, 4, 128, 3, 1, 1, True, False),  # ... three frames per workgroup (odd run), two co groups, 16x4 plane; 1300 above: six / five
    (5, 64, 8, 8, 64, 3, 1, 1, False, False),   # ... no activation
    (3, 8, 64, 16, 16, 3, 2, 1, True, False),   # second encoder layer, audio plane (32x8 output): staged stride-2 weight gradient (conv3x3s2_wgrad_staged_kernel<2, 8>); (2, 8, 32, 32, 16, ...) above is the vision plane (<2, 16>)
    (1500, 8, 32, 32, 16, 3, 2, 1, True, False),  # ... six frames per workgroup: the three raw register sets and both image buffers go round; forward: conv3x3s2_band_kernel<8, 0> with three tiles per workgroup (both register sets of requests in use); backward-data: convt_quad_resident_kernel<16, 8, 256, true>
    (700, 8, 64, 16, 16, 3, 2, 1, False, False),  # ... three frames per workgroup, no activation
    (700, 1, 64, 64, 8, 3, 2, 1, True, True),   # first encoder layer (forward: conv3x3s2_band_kernel<1, 2>, 2800 band tiles, coordinate channels from the frame-independent planes), staged (conv3x3s2_thin_wgrad_staged_kernel<2, 32>): three frames per workgroup, activation on frame and coordinate channels; (4, 1, 64, 64, 8, ...) / (3, 1, 128, 32, 8, ...) above are its one-frame cases
    (300, 1, 128, 32, 8, 3, 2, 1, False, True),  # ... audio plane (<2, 16>), two frames per workgroup
    (700, 16, 16, 16, 32, 3, 2, 1, True, False),  # third encoder layer, staged (conv3x3s2c_wgrad_staged_kernel<2, 8>): three frames per workgroup; (3, 16, 16, 16, 32, ...) above is its one-frame case
    (300, 16, 32, 8, 32, 3, 2, 1, False, False),  # ... audio plane (16x4 output, <2, 4>), two frames per workgroup, no activation
    (5, 16, 32, 8, 32, 3, 2, 1, True, False),   # ... one frame per workgroup
]


@pytest.fixture(params=["bf16x2", "bf16x3", "f32"])
def fp32_grade_mode(request):  # noqa: ANN001, ANN201
    """The MFMA operand formats of the conv kernels that meet the parity tolerances (conv.set_mfma_mode): the default
    "bf16x2" (two bf16 pieces per operand, three bf16 MFMA products, fp32 accumulation), "bf16x3" (three pieces, six
    products) and the fp32 MFMA kernels; same tolerances for all three."""
    from multimodal_mtrssm_amd import conv

    before = conv.mfma_mode()
    conv.set_mfma_mode(request.param)
    yield request.param
    conv.set_mfma_mode(before)


@pytest.mark.parametrize(("n", "cin", "h", "w", "cout", "k", "s", "p", "pre", "coords"), CONV_CASES)
def test_conv2d_kernel(n, cin, h, w, cout, k, s, p, pre, coords, fp32_grade_mode: str, lib_loaded: None) -> None:  # noqa: ANN001, PLR0913
    import torch.nn.functional as F  # noqa: N812

    from multimodal_mtrssm_amd.conv import conv2d

    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, cin, h, w, generator=g).requires_grad_()
    cc = torch.randn(2, h, w, generator=g) if coords else None
    wt = (torch.randn(cout, cin + (2 if coords else 0), k, k, generator=g) * 0.2).requires_grad_()
    b = torch.randn(cout, generator=g).requires_grad_()
    xin = F.elu(x) if pre else x
    if coords:
        cin_full = torch.cat([xin, (F.elu(cc) if pre else cc).unsqueeze(0).expand(n, -1, -1, -1)], 1)
    else:
        cin_full = xin
    want = F.conv2d(cin_full, wt, b, s, p)
    gout = torch.randn(want.shape, generator=g)
    want.backward(gout)
    xg = x.detach().to(DEV).requires_grad_()
    wg = wt.detach().to(DEV).requires_grad_()
    bg = b.detach().to(DEV).requires_grad_()
    got = conv2d(xg, wg, bg, stride=s, padding=p, pre_act=pre, act=2, coords=None if cc is None else cc.to(DEV))
    got.backward(gout.to(DEV))
    # tolerances relative to each tensor's scale (weight gradients are sums over all pixels: O(30) here): 2e-5 of the max
    # covers the default two-piece operands (measured 8e-6) and is ~10x what the three-piece / fp32 kernels need
    np.testing.assert_allclose(_np(got), want.detach().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(_np(xg.grad), x.grad.numpy(), rtol=1e-4, atol=max(1e-4, 1e-5 * float(x.grad.abs().max())))
    np.testing.assert_allclose(_np(wg.grad), wt.grad.numpy(), rtol=1e-4, atol=max(2e-4, 2e-5 * float(wt.grad.abs().max())))
    np.testing.assert_allclose(_np(bg.grad), b.grad.numpy(), rtol=1e-4, atol=max(2e-4, 2e-5 * float(b.grad.abs().max())))


DECONV_CASES = [
    # n, cin, h, w, cout, k, s, p, op, pre_act
    (3, 64, 4, 2, 32, 4, 2, 1, 0, True),
    (2, 16, 8, 8, 1, 4, 2, 1, 0, True),    # last decoder layer: one output channel
    (2, 8, 5, 3, 4, 3, 2, 1, 1, False),    # odd kernel, output_padding
    (2, 12, 6, 5, 7, 3, 1, 1, 0, True),    # stride 1
    (3, 64, 8, 8, 32, 4, 2, 1, 0, True),   # decoder shapes: patch-staged weight gradient with 16 taps
    (2, 32, 16, 16, 16, 4, 2, 1, 0, True),
    (2, 16, 32, 32, 1, 4, 2, 1, 0, True),
    (5, 64, 16, 4, 32, 4, 2, 1, 0, True),  # audio decoder plane
    (3, 16, 13, 37, 1, 4, 2, 1, 0, True),  # last-layer kernel (convt_k4s2_thin): ragged tile edges
    (2, 5, 8, 40, 2, 4, 2, 1, 0, False),   # two output channels, no activation
    (2, 16, 64, 16, 1, 4, 2, 1, 0, True),  # audio plane 128 x 32 output
    (3, 32, 32, 8, 16, 4, 2, 1, 0, True),  # second decoder layer, audio plane: staged weight gradient (convt4s2_wgrad_staged_kernel<2, 8>); (2, 32, 16, 16, 16, ...) above is the vision plane (<2, 16>)
    (700, 32, 16, 16, 16, 4, 2, 1, 0, True),  # ... three frames per workgroup: the raw register sets go round
    (300, 32, 32, 8, 16, 4, 2, 1, 0, False),  # ... two frames per workgroup (the last ones one), no activation
    (700, 64, 8, 8, 32, 4, 2, 1, 0, True),    # first decoder layer (convt4s2b_wgrad_staged_kernel<2, 8>): three frames per workgroup; (3, 64, 8, 8, 32, ...) and (5, 64, 16, 4, 32, ...) above are its one-frame cases (<2, 8>, <2, 4>)
    (300, 64, 16, 4, 32, 4, 2, 1, 0, False),  # ... audio plane, two frames per workgroup, no activation
    (700, 16, 32, 32, 1, 4, 2, 1, 0, True),   # last decoder layer, staged weight gradient (convt4s2_thin_wgrad_staged_kernel<2, 32>): three frames per workgroup; (2, 16, 32, 32, 1, ...) and (2, 16, 64, 16, 1, ...) above are its one-frame cases
    (300, 16, 64, 16, 1, 4, 2, 1, 0, False),  # ... audio plane (<2, 16>), two frames per workgroup, no activation
]


@pytest.mark.parametrize(("n", "cin", "h", "w", "cout", "k", "s", "p", "op", "pre"), DECONV_CASES)
def test_conv_transpose2d_kernel(n, cin, h, w, cout, k, s, p, op, pre, fp32_grade_mode: str, lib_loaded: None) -> None:  # noqa: ANN001, PLR0913
    import torch.nn.functional as F  # noqa: N812

    from multimodal_mtrssm_amd.conv import conv_transpose2d

    g = torch.Generator().manual_seed(12)
    x = torch.randn(n, cin, h, w, generator=g).requires_grad_()
    wt = (torch.randn(cin, cout, k, k, generator=g) * 0.2).requires_grad_()
    b = torch.randn(cout, generator=g).requires_grad_()
    want = F.conv_transpose2d(F.elu(x) if pre else x, wt, b, s, p, op)
    gout = torch.randn(want.shape, generator=g)
    want.backward(gout)
    xg = x.detach().to(DEV).requires_grad_()
    wg = wt.detach().to(DEV).requires_grad_()
    bg = b.detach().to(DEV).requires_grad_()
    got = conv_transpose2d(xg, wg, bg, stride=s, padding=p, output_padding=op, pre_act=pre, act=2)
    got.backward(gout.to(DEV))
    assert got.shape == want.shape
    # tolerances relative to each tensor's scale (weight gradients are sums over all pixels: O(30) here): 2e-5 of the max
    # covers the default two-piece operands (measured 8e-6) and is ~10x what the three-piece / fp32 kernels need
    np.testing.assert_allclose(_np(got), want.detach().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(_np(xg.grad), x.grad.numpy(), rtol=1e-4, atol=max(1e-4, 1e-5 * float(x.grad.abs().max())))
    np.testing.assert_allclose(_np(wg.grad), wt.grad.numpy(), rtol=1e-4, atol=max(2e-4, 2e-5 * float(wt.grad.abs().max())))
    np.testing.assert_allclose(_np(bg.grad), b.grad.numpy(), rtol=1e-4, atol=max(2e-4, 2e-5 * float(b.grad.abs().max())))


@pytest.mark.parametrize("name", ["mrssm_default", "mrssm_nonsquare"])
def test_encoder_decoder_match_oracle(name: str, fp32_grade_mode: str, lib_loaded: None) -> None:
    """Build-defined conv stacks (cnn is absent upstream: parity unpinned): HIP build vs oracle/ref_cnn.py on CPU."""
    import multimodal_mtrssm_amd as mt
    from oracle.ref_cnn import Decoder, Encoder

    d = CASES[name].dims
    torch.manual_seed(5)
    for cfg, ref_cls, mine_cls, shape in ((d.enc_audio, Encoder, mt.Encoder, (2, 3, *CASES[name].audio_shape)),
                                           (d.dec_vision, Decoder, mt.Decoder, (2, 3, d.deter + d.stoch))):
        ref = ref_cls(cfg)
        mine = mine_cls(cfg)
        mine.load_state_dict(ref.state_dict())
        mine = mine.to(DEV)
        x = torch.randn(shape).requires_grad_()
        y = ref(x)
        gy = torch.randn(y.shape)
        y.backward(gy)
        xg = x.detach().to(DEV).requires_grad_()
        yg = mine(xg)
        yg.backward(gy.to(DEV))
        np.testing.assert_allclose(_np(yg), y.detach().numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(_np(xg.grad), x.grad.numpy(), rtol=1e-3, atol=1e-5 * float(x.grad.abs().max() + 1))
        for (k, pr), (_, pm) in zip(ref.named_parameters(), mine.named_parameters(), strict=True):
            scale = float(pr.grad.abs().max()) + 1e-12
            np.testing.assert_allclose(_np(pm.grad), pr.grad.numpy(), rtol=1e-3, atol=2e-4 * scale, err_msg=k)
        # a single frame gives the same embedding as that frame inside a [B,T] batch
        np.testing.assert_allclose(_np(mine(xg[:, 0])), _np(yg[:, 0]), rtol=1e-5, atol=1e-6)


def test_paired_stacks_of_different_shape_fall_back(lib_loaded: None) -> None:
    """cnn.encode_pair / decode_pair with two stacks that do NOT match layer for layer (other channel counts, other number
    of residual blocks): the walk falls back to single launches where shapes differ and gives what the modules give alone."""
    import copy

    import multimodal_mtrssm_amd as mt
    from multimodal_mtrssm_amd import cnn

    d = CASES["mrssm_default"].dims
    torch.manual_seed(9)
    cfg_a = copy.deepcopy(dict(d.enc_audio))
    cfg_b = copy.deepcopy(dict(d.enc_vision))
    cfg_b["channels"] = [c // 2 for c in cfg_b["channels"]]
    cfg_b["num_residual_blocks"] = cfg_b.get("num_residual_blocks", 0) + 1
    ea, eb = mt.Encoder(cfg_a).to(DEV), mt.Encoder(cfg_b).to(DEV)
    xa = torch.randn(2, 3, *CASES["mrssm_default"].audio_shape).to(DEV).requires_grad_()
    xb = torch.randn(2, 3, *CASES["mrssm_default"].vision_shape).to(DEV).requires_grad_()
    ya1, yb1 = ea(xa), eb(xb)  # the first call builds the layers on the input's device
    (ya1.sum() + yb1.square().sum()).backward()
    want = [ya1.detach(), yb1.detach(), xa.grad.clone(), xb.grad.clone(), *(p.grad.clone() for p in (*ea.parameters(), *eb.parameters()))]
    xa.grad = xb.grad = None
    for p in (*ea.parameters(), *eb.parameters()):
        p.grad = None
    ya2, yb2 = cnn.encode_pair(ea, eb, xa, xb)
    (ya2.sum() + yb2.square().sum()).backward()
    got = [ya2.detach(), yb2.detach(), xa.grad, xb.grad, *(p.grad for p in (*ea.parameters(), *eb.parameters()))]
    for i, (g, w) in enumerate(zip(got, want, strict=True)):
        # weight gradients meet in fp32 atomics (conv partial tiles, split GEMM reductions): equal up to their arrival order
        np.testing.assert_allclose(_np(g), _np(w), rtol=1e-4, atol=5e-6 * float(w.abs().max() + 1e-12), err_msg=str(i))

    da_cfg, db_cfg = copy.deepcopy(dict(d.dec_audio)), copy.deepcopy(dict(d.dec_vision))
    db_cfg["num_residual_blocks"] = 0
    da, db = mt.Decoder(da_cfg).to(DEV), mt.Decoder(db_cfg).to(DEV)
    f = torch.randn(2, 3, d.deter + d.stoch).to(DEV)
    ra, rb = da(f), db(f)
    pa, pb = cnn.decode_pair(da, db, f, f)
    np.testing.assert_allclose(_np(pa), _np(ra), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(_np(pb), _np(rb), rtol=1e-6, atol=1e-6)


def test_one_pass_parity_class_kernels_match_the_per_class_launches(lib_loaded: None) -> None:
    """convt_quad_resident_kernel (all four output parity classes of a k = 4, s = 2 ConvTranspose2d in one pass, single and
    paired, and -- with the act' operand -- the backward-data of a k = 3, s = 2 Conv2d) and conv_tgather_thin_kernel (the thin
    transposed gather) against the general path they replace (one gather launch per parity class): the same products, summed
    in another order."""
    from multimodal_mtrssm_amd import conv

    gen = torch.Generator(device="cpu").manual_seed(23)

    def rnd(*shape: int, scale: float = 1.0) -> torch.Tensor:
        return (torch.randn(*shape, generator=gen) * scale).to(DEV).requires_grad_(True)

    def run() -> list[torch.Tensor]:
        gen.manual_seed(23)
        outs = []
        # decoder shapes: 64 -> 32 on 8x8 (vision) and 16x4 (audio) planes, paired; 32 -> 16 on 16x16, single
        xa, xv = rnd(6, 64, 16, 4), rnd(6, 64, 8, 8)
        wa, wv, ba, bv = rnd(64, 32, 4, 4, scale=0.05), rnd(64, 32, 4, 4, scale=0.05), rnd(32, scale=0.1), rnd(32, scale=0.1)
        ya, yv = conv.conv_transpose2d_pair((xa, wa, ba, 2, 1, 0, True, 2), (xv, wv, bv, 2, 1, 0, True, 2))
        x2, w2, b2 = rnd(5, 32, 16, 16), rnd(32, 16, 4, 4, scale=0.05), rnd(16, scale=0.1)
        y2 = conv.conv_transpose2d(x2, w2, b2, stride=2, padding=1, output_padding=0, pre_act=True, act=2)
        # encoder shapes: the backward-data of 16 -> 32 (k3 s2, quad kernel with act') and of 8 -> 16 (thin transposed gather)
        x3, w3, b3 = rnd(4, 16, 16, 16), rnd(32, 16, 3, 3, scale=0.1), rnd(32, scale=0.1)
        y3 = conv.conv2d(x3, w3, b3, stride=2, padding=1, pre_act=True, act=2)
        x4, w4, b4 = rnd(4, 8, 32, 32), rnd(16, 8, 3, 3, scale=0.1), rnd(16, scale=0.1)
        y4 = conv.conv2d(x4, w4, b4, stride=2, padding=1, pre_act=True, act=2)
        (ya.square().sum() + yv.sin().sum() + y2.square().sum() + y3.sin().sum() + y4.square().sum()).backward()
        torch.cuda.synchronize()
        outs += [ya.detach(), yv.detach(), y2.detach(), xa.grad, xv.grad, x2.grad, x3.grad, x4.grad]
        return outs

    res = {}
    for one_pass in (False, True):
        conv.CONVT_QUAD = conv.TGATHER_THIN = conv.CONVT_QUAD_BWD = one_pass
        try:
            res[one_pass] = run()
        finally:
            conv.CONVT_QUAD = conv.TGATHER_THIN = True
            conv.CONVT_QUAD_BWD = True
    for i, (a, b) in enumerate(zip(res[True], res[False], strict=True)):
        # (x4.grad: the per-class path is the fp32 VALU kernel, the one-pass path two bf16 pieces per operand: 16 significant bits)
        np.testing.assert_allclose(_np(a), _np(b), rtol=2e-5, atol=(5e-6 if i == 7 else 2e-6) * float(b.abs().max()), err_msg=str(i))


@pytest.mark.parametrize(("na", "nv", "plane_a", "plane_v", "act", "mid"),
                         [(1, 0, (8, 8), None, 2, 128), (37, 50, (8, 8), (16, 4), 2, 128), (300, 700, (4, 16), (8, 8), 2, 128),
                          (260, 0, (8, 8), None, 1, 128), (2, 0, (8, 8), None, 2, 64), (38, 50, (16, 4), (8, 8), 2, 64),
                          (700, 300, (8, 8), (4, 16), 2, 64), (520, 0, (8, 8), None, 1, 64)])
def test_fused_residual_block_matches_two_launches_and_float64(lib_loaded: None, na: int, nv: int, plane_a: tuple, plane_v: tuple | None,  # noqa: PLR0913
                                                               act: int, mid: int) -> None:
    """mtrssm_residual_block_fwd (3x3 -> act -> 1x1 + skip in one launch, 64 channels / 128 or 64 intermediate channels on 64-pixel
    planes: conv3x3_resident_kernel<64, 4 | 2, 1, false, true>) against the two-launch path (same two-piece products, another summation order in the 1x1) and against float64 on
    the CPU (`oracle/ref_cnn.py:ResidualBlock` arithmetic), values and every gradient; fewer frames than CUs, uneven pairs
    and several tiles per workgroup."""
    import torch.nn.functional as F  # noqa: N812

    from multimodal_mtrssm_amd import _lib, conv

    gen = torch.Generator(device="cpu").manual_seed(5 + na)
    def rnd(*shape: int, scale: float = 1.0) -> torch.Tensor:
        return (torch.randn(*shape, generator=gen) * scale).to(DEV).requires_grad_(True)
    def params() -> tuple:
        return (rnd(mid, 64, 3, 3, scale=0.05), rnd(mid, scale=0.1), rnd(64, mid, 1, 1, scale=0.1), rnd(64, scale=0.1))
    xa, pa = rnd(na, 64, *plane_a), params()
    xv, pv = (rnd(nv, 64, *plane_v), params()) if nv else (None, None)
    leaves = [xa, *pa] + ([xv, *pv] if nv else [])

    def run(fused: bool) -> list[torch.Tensor]:
        conv.RESBLOCK_FUSE = fused
        conv.invalidate_packs()
        for t in leaves:
            t.grad = None
        try:
            with torch.no_grad():  # which kernel the forward is
                conv.residual_block(xa, *pa, act=act)
            last = _lib.load().mtrssm_last_kernel().decode()
            assert last.endswith("false, true>") == fused, last
            if nv:
                ya, yv = conv.residual_block_pair(xa, pa, xv, pv, act=act)
                (ya.square().sum() + yv.sin().sum()).backward()
                outs = [ya.detach(), yv.detach()]
            else:
                ya = conv.residual_block(xa, *pa, act=act)
                ya.square().sum().backward()
                outs = [ya.detach()]
            torch.cuda.synchronize()
        finally:
            conv.RESBLOCK_FUSE = True
        return outs + [t.grad.clone() for t in leaves]

    two, one = run(False), run(True)
    for i, (a, b) in enumerate(zip(one, two, strict=True)):
        np.testing.assert_allclose(_np(a), _np(b), rtol=2e-5, atol=3e-6 * float(b.abs().max()), err_msg=str(i))
    # float64 on the CPU
    def ref(x: torch.Tensor, p: tuple) -> torch.Tensor:
        w3, b3, w1, b1 = p
        fn = F.elu if act == 2 else F.relu  # _lib.ACT_IDS
        return x + F.conv2d(fn(F.conv2d(fn(x), w3, b3, 1, 1)), w1, b1)
    cpu = [t.detach().double().cpu().requires_grad_(True) for t in leaves]
    ya64 = ref(cpu[0], tuple(cpu[1:5]))
    loss = ya64.square().sum()
    outs64 = [ya64]
    if nv:
        yv64 = ref(cpu[5], tuple(cpu[6:10]))
        loss = loss + yv64.sin().sum()
        outs64.append(yv64)
    loss.backward()
    for i, (a, b) in enumerate(zip(one, [o.detach() for o in outs64] + [t.grad for t in cpu], strict=True)):
        if act == 1 and i >= len(outs64):
            break  # ReLU: an intermediate value within rounding of 0 switches its gradient on or off; the values are compared
        np.testing.assert_allclose(_np(a), b.numpy(), rtol=1e-4, atol=4e-5 * float(b.abs().max()), err_msg=f"float64 {i}")


def test_paired_launches_change_nothing(lib_loaded: None) -> None:
    """conv.paired (the audio and the vision stack's equal layers in one launch, the default) against one launch per
    layer: the gathers are the same arithmetic per output tile, so forward values are bit-identical; gradients are equal
    up to the arrival order of the weight-gradient kernels' fp32 atomics."""
    from multimodal_mtrssm_amd import conv

    # (1) the merged kernel itself, at a shape the split MFMA kernels take (64 -> 64 channels, 3x3 and 1x1)
    gen = torch.Generator(device="cpu").manual_seed(11)
    def rnd(*shape: int, scale: float = 1.0) -> torch.Tensor:
        return (torch.randn(*shape, generator=gen) * scale).to(DEV).requires_grad_(True)
    res = {}
    for paired in (False, True):
        conv.PAIR_LAUNCH = paired
        try:
            gen.manual_seed(11)
            xa, xv = rnd(96, 64, 16, 16), rnd(96, 64, 16, 16)
            pa = (rnd(32, 64, 3, 3, scale=0.05), rnd(32, scale=0.1), rnd(64, 32, 1, 1, scale=0.1), rnd(64, scale=0.1))
            pv = (rnd(32, 64, 3, 3, scale=0.05), rnd(32, scale=0.1), rnd(64, 32, 1, 1, scale=0.1), rnd(64, scale=0.1))
            ya, yv = conv.residual_block_pair(xa, pa, xv, pv, act=1)
            (ya.square().sum() + yv.sin().sum()).backward()
            torch.cuda.synchronize()
            res[paired] = [ya.detach(), yv.detach(), xa.grad, xv.grad, *(t.grad for t in pa), *(t.grad for t in pv)]
        finally:
            conv.PAIR_LAUNCH = True
    for i, (a, b) in enumerate(zip(res[True], res[False], strict=True)):
        if i < 4:
            assert torch.equal(a, b), i
        else:
            np.testing.assert_allclose(_np(a), _np(b), rtol=1e-4, atol=1e-5 * float(b.abs().max()), err_msg=str(i))
    # and against the unpaired single-block node
    ya1 = conv.residual_block(xa.detach(), *(t.detach() for t in pa), act=1)
    assert torch.equal(ya1, res[True][0])

    # (2) the whole train step
    case = CASES["mrssm_default"]
    fx = load_golden("mrssm_default")
    batch, noise = tuple(b.to(DEV) for b in golden_batch(fx)), _to(golden_noise(fx), DEV)
    runs = {}
    for paired in (False, True):
        conv.PAIR_LAUNCH = paired
        try:
            model = product_from_case(case, build_model(case), DEV)
            model.zero_grad(set_to_none=True)
            out = model.shared_step(batch, noise)
            out["loss"].backward()
            torch.cuda.synchronize()
            runs[paired] = ({k: float(v) for k, v in out.items()}, {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
        finally:
            conv.PAIR_LAUNCH = True
    for k, v in runs[False][0].items():
        np.testing.assert_allclose(runs[True][0][k], v, rtol=2e-6, err_msg=k)
    for k, g in runs[False][1].items():
        scale = float(g.abs().max()) + 1e-12
        np.testing.assert_allclose(_np(runs[True][1][k]), _np(g), rtol=1e-4, atol=1e-5 * scale, err_msg=k)


def test_step_pack_plan_changes_nothing(lib_loaded: None) -> None:
    """conv._PackPlan (every conv weight of a step packed by one launch at the start of shared_step, the default) against
    one pack launch per use: packing is a pure re-layout, so the first step's loss agrees to the last bits (fp32 atomics in the NLL sum); also a
    weight edited between begin_step and its use must be picked up (version counter), as must the optimizer's raw writes."""
    from multimodal_mtrssm_amd import conv
    from multimodal_mtrssm_amd.optim import FlatAdamW, FlatParameters

    case = CASES["mrssm_default"]
    fx = load_golden("mrssm_default")
    batch, noise = tuple(b.to(DEV) for b in golden_batch(fx)), _to(golden_noise(fx), DEV)
    losses = {}
    for plan in (False, True):
        conv.PACK_PLAN = plan
        try:
            model = product_from_case(case, build_model(case), DEV)
            flat = FlatParameters(model)
            opt = FlatAdamW(flat, lr=1e-3)
            trace = []
            for _ in range(4):
                opt.zero_grad()
                out = model.shared_step(batch, noise)
                out["loss"].backward()
                opt.step()
                trace.append(float(out["loss"]))
            with torch.no_grad():  # an edit autograd can see, after the step's pack launch
                conv.begin_step(torch.device(DEV))
                model.audio_encoder.res[0].conv3.weight.mul_(0.5)
                trace.append(float(model.shared_step(batch, noise)["loss"]))
            losses[plan] = trace
        finally:
            conv.PACK_PLAN = True
    np.testing.assert_allclose(losses[True][0], losses[False][0], rtol=1e-6)  # same packed values; the NLL sum uses fp32 atomics
    # later steps differ only by the arrival order of the weight-gradient atomics feeding the optimizer
    np.testing.assert_allclose(losses[True], losses[False], rtol=2e-5)
    assert losses[True][1] < losses[True][0]  # and the optimizer's updates were seen by the next step's packs
    if conv.PACK_PLAN:
        assert conv._PLAN.entries, "the plan was never used"  # noqa: SLF001


def test_fp32_mfma_mode_train_step_matches_golden(lib_loaded: None) -> None:
    """The whole train step with the fp32 MFMA conv kernels ("f32" mode; every other GPU test of the step runs in the
    default "bf16x2" mode): same golden losses, same tolerance."""
    from multimodal_mtrssm_amd import conv

    conv.set_mfma_mode("f32")
    try:
        case = CASES["mrssm_default"]
        fx = load_golden("mrssm_default")
        model = product_from_case(case, build_model(case), DEV)
        out = model.shared_step(tuple(b.to(DEV) for b in golden_batch(fx)), _to(golden_noise(fx), DEV))
        for k in out:
            np.testing.assert_allclose(float(out[k].detach()), float(fx[f"loss/{k}"]), rtol=2e-5, err_msg=k)
    finally:
        conv.set_mfma_mode("bf16x2")


def test_bf16_mode_stays_within_its_stated_tolerance(lib_loaded: None) -> None:
    """"bf16" conv mode (plain bf16 MFMA operands, fp32 accumulation; BASELINE configs name bf16): NOT fp32 parity.
    Stated tolerance: every loss within 2e-3 relative of the golden fp32 value, conv outputs within 2 % of the tensor's
    max (bf16 has 8 significant bits).  The fp32 bar (1e-4 ELBO) is met by the default mode, not by this one."""
    import torch.nn.functional as F  # noqa: N812

    from multimodal_mtrssm_amd import conv

    conv.set_mfma_mode("bf16")
    try:
        case = CASES["mrssm_default"]
        fx = load_golden("mrssm_default")
        model = product_from_case(case, build_model(case), DEV)
        out = model.shared_step(tuple(b.to(DEV) for b in golden_batch(fx)), _to(golden_noise(fx), DEV))
        for k in out:
            np.testing.assert_allclose(float(out[k].detach()), float(fx[f"loss/{k}"]), rtol=2e-3, err_msg=k)
        g = torch.Generator().manual_seed(21)
        x = torch.randn(4, 64, 8, 8, generator=g)
        wt = torch.randn(64, 64, 3, 3, generator=g) * 0.1
        want = F.conv2d(F.elu(x), wt, None, 1, 1)
        got = conv.conv2d(x.to(DEV), wt.to(DEV), None, stride=1, padding=1, pre_act=True, act=2)
        err = float((got.cpu() - want).abs().max() / want.abs().max())
        assert 1e-5 < err < 2e-2, err  # really bf16 operands (not silently fp32), within the stated bound
    finally:
        conv.set_mfma_mode("bf16x2")


# ---------------------------------------------------------------------------------------------
# fp32-MFMA GEMM (csrc/gemm.hip) and the Linear layers built on it
# ---------------------------------------------------------------------------------------------
GEMM_CASES = [
    # M, N, R, a_rmajor, b_rmajor, extras
    (3200, 256, 4096, False, False, "bias+act_a"),   # encoder head forward: act(X) W^T + b
    (3200, 200, 4, False, False, "bias"),            # action projection: 4-wide reduction, unaligned rows (lda = 4 ok, K tail)
    (64, 200, 256, False, False, "bias"),            # init_proj on B rows
    (130, 70, 33, False, False, "views"),            # ragged everything, operands are column slices of wider matrices
    (3200, 4096, 256, False, True, "zgrad"),         # encoder head data gradient: (dY W) * act'(X)
    (100, 230, 64, False, True, ""),                 # decoder stem data gradient
    (200, 230, 3200, True, True, "colsum+acc"),      # scan weight gradient + bias gradient, split reduction (atomics)
    (30, 200, 3200, True, True, "colsum+acc"),       # narrow head weight gradient
    (256, 4096, 3200, True, True, "acc+act_b"),      # encoder head weight gradient with the fused activation on X
    (70, 50, 129, True, True, "views+acc"),          # ragged, strided views, accumulate into a running target
    (65, 33, 70, True, False, ""),                   # the fourth layout (unused by the model, same kernel)
    (3200, 4096, 64, False, False, "bias"),          # decoder stem forward: two k-steps, 800 tiles
    (3200, 64, 4096, False, True, "zgrad"),          # decoder stem data gradient: 64-column tiles, split reduction with the last-arriver epilogue
    (4096, 64, 3200, True, True, "colsum+acc"),      # decoder stem weight gradient
    (1024, 3072, 1024, True, False, "bias"),         # the fused GRU input matrix of the large model, fourth layout on full tiles
    (128, 1024, 3200, True, True, "acc"),            # a 128-row weight gradient: one row of tiles, reduction split 4 ways
]


@pytest.mark.parametrize("pieces", [0, 2, 3])
@pytest.mark.parametrize(("m", "n", "r", "a_rm", "b_rm", "extras"), GEMM_CASES)
def test_gemm_kernel(m: int, n: int, r: int, a_rm: bool, b_rm: bool, extras: str, pieces: int, lib_loaded: None) -> None:  # noqa: PLR0913
    """mtrssm_gemm against float64 matmul: every layout, ragged tiles, strided views, fused activations / bias / act' / column sums /
    accumulation, on the fp32 MFMA kernel (pieces = 0: tolerance 2e-6 of the result's scale -- an fp32 fma chain over <= 4096
    terms; the reference's fp32 nn.Linear has the same) and on the split-bf16 kernel the large Linear layers of the conv stacks use
    (pieces = 2: operands carry 16 significant bits, 3e-5 of the scale -- the conv kernels' default arithmetic) resp. the tile kernel
    of csrc/gemm_tile.h on full 128 x {128, 64} tiles (pieces = 2 | 3; 3 = three bf16 pieces per operand, exact to 2^-24: the fp32
    tolerance; shapes it does not take fall back to the split kernel resp. the fp32 MFMA kernel)."""
    import torch.nn.functional as F  # noqa: N812

    from multimodal_mtrssm_amd.linear import gemm

    g = torch.Generator().manual_seed(m * 7 + n * 3 + r)
    pad = 5 if "views" in extras else 0
    a_full = torch.randn((r, m + pad) if a_rm else (m, r + pad), generator=g)
    b_full = torch.randn((r, n + pad) if b_rm else (n, r + pad), generator=g)
    a = a_full[:, pad:] if pad else a_full
    b = b_full[:, pad:] if pad else b_full
    a_ir = a.t() if a_rm else a   # [M, R]
    b_jr = b.t() if b_rm else b   # [N, R]
    if "act_a" in extras:
        a_ir = F.elu(a_ir)
    if "act_b" in extras:
        b_jr = F.elu(b_jr)
    want = a_ir.double() @ b_jr.double().t()
    bias = torch.randn(n, generator=g) if "bias" in extras else None
    if bias is not None:
        want = want + bias.double()
    z = torch.randn(m, n, generator=g) if "zgrad" in extras else None
    if z is not None:
        want = want * torch.where(z > 0, torch.ones_like(z), z.exp()).double()
    c0 = torch.randn(m, n + pad, generator=g) if "acc" in extras else torch.full((m, n + pad), float("nan"))
    if "acc" in extras:
        want = want + c0[:, pad:].double()
    c = c0.to(DEV)
    colsum0 = torch.randn(m, generator=g) if "colsum" in extras else None
    colsum = None if colsum0 is None else colsum0.to(DEV)
    ag, bg = a_full.to(DEV), b_full.to(DEV)
    gemm(ag[:, pad:] if pad else ag, bg[:, pad:] if pad else bg, c[:, pad:] if pad else c, a_rmajor=a_rm, b_rmajor=b_rm,
         bias=None if bias is None else bias.to(DEV), zgrad=None if z is None else z.to(DEV), colsum=colsum,
         act_a=2 if "act_a" in extras else 0, act_b=2 if "act_b" in extras else 0, act_z=2 if z is not None else 0,
         accumulate="acc" in extras, mfma_split=pieces)
    got = c[:, pad:] if pad else c
    scale = float(want.abs().max())
    rtol, tol = (1e-4, 3e-5) if pieces == 2 else (1e-5, 2e-6)
    np.testing.assert_allclose(_np(got), want.float().numpy(), rtol=rtol, atol=tol * scale)
    if pad and "acc" not in extras:
        assert torch.isnan(c[:, :pad]).all()  # nothing outside the view was written
    if colsum is not None:
        want_cs = colsum0.double() + (a.t() if a_rm else a).double().sum(1)
        np.testing.assert_allclose(_np(colsum), want_cs.float().numpy(), rtol=rtol, atol=tol * float(want_cs.abs().max()))


def test_grouped_gemm_equals_the_single_launches(lib_loaded: None) -> None:
    """mtrssm_gemm_group (independent problems of one layout in one grid; the scan's weight gradients go through it): the same
    arithmetic per problem as mtrssm_gemm -- bitwise when the reduction is not split, to the arrival order of the fp32 atomics
    when it is -- over ragged and full tiles, both weight-gradient style (r-major operands, accumulate, column sums) and plain
    forward problems in ONE call (two layouts = two launches), and more problems than one launch holds (24)."""
    from multimodal_mtrssm_amd import linear

    g = torch.Generator().manual_seed(3)
    problems, singles = [], []
    shapes = [(200, 200, 3200), (30, 200, 3200), (600, 200, 640), (64, 128, 256), (7, 5, 33)] * 6  # 30 weight-gradient problems
    for i, (m, n, r) in enumerate(shapes):
        a = torch.randn(r, m, generator=g).to(DEV)
        b = torch.randn(r, n, generator=g).to(DEV)
        c0 = torch.randn(m, n, generator=g).to(DEV)
        cs0 = torch.randn(m, generator=g).to(DEV)
        c1, c2, cs1, cs2 = c0.clone(), c0.clone(), cs0.clone(), cs0.clone()
        kw = dict(a_rmajor=True, b_rmajor=True, accumulate=True)
        problems.append(((a, b, c1), dict(kw, colsum=cs1 if i % 2 == 0 else None)))
        singles.append(((a, b, c2), dict(kw, colsum=cs2 if i % 2 == 0 else None), cs1, cs2))
    for m, n, r in ((3200, 200, 256), (100, 64, 96)):  # forward-layout problems in the same call
        a = torch.randn(m, r, generator=g).to(DEV)
        b = torch.randn(n, r, generator=g).to(DEV)
        bias = torch.randn(n, generator=g).to(DEV)
        c1, c2 = torch.empty(m, n, device=DEV), torch.empty(m, n, device=DEV)
        problems.append(((a, b, c1), dict(a_rmajor=False, b_rmajor=False, bias=bias)))
        singles.append(((a, b, c2), dict(a_rmajor=False, b_rmajor=False, bias=bias), None, None))
    # two problems that accumulate into the SAME matrix (the prior head's weights get a gradient from the initial state and one from
    # the scan): they must not share a launch (an unsplit reduction accumulates by plain read-modify-write) -- successive rounds
    a = torch.randn(40, 64, generator=g).to(DEV)
    b = torch.randn(40, 32, generator=g).to(DEV)
    shared1, shared2 = torch.zeros(64, 32, device=DEV), torch.zeros(64, 32, device=DEV)
    for _ in range(2):
        problems.append(((a, b, shared1), dict(a_rmajor=True, b_rmajor=True, accumulate=True)))
        singles.append(((a, b, shared2), dict(a_rmajor=True, b_rmajor=True, accumulate=True), None, None))
    assert len(linear._rounds(problems)) == 2  # noqa: SLF001
    linear.gemm_group(problems)
    for (abc, kw, _cs1, _cs2) in singles:
        linear.gemm(*abc, **kw)
    torch.cuda.synchronize()
    np.testing.assert_allclose(_np(shared1), 2.0 * _np(a.t() @ b), rtol=1e-5, atol=1e-4)
    for ((_, _, c1), _), ((a, _, c2), kw, cs1, cs2) in zip(problems, singles, strict=True):
        scale = float(c2.abs().max()) + 1e-9
        np.testing.assert_allclose(_np(c1), _np(c2), rtol=0, atol=2e-6 * scale)
        if kw.get("colsum") is not None:
            np.testing.assert_allclose(_np(cs1), _np(cs2), rtol=0, atol=2e-5 * (float(cs2.abs().max()) + 1e-9))


def test_linear_function_and_gradient_sink(lib_loaded: None) -> None:
    """linear.linear (= F.linear(act(x), W, b)) forward / backward against torch on the CPU, once with plain parameters (gradients
    returned to autograd) and once inside a FlatParameters module (weight / bias gradients accumulated by the GEMM straight into
    the flat buffer: .grad views filled, the autograd node hands back None, the parameters are marked touched)."""
    import torch.nn.functional as F  # noqa: N812

    import multimodal_mtrssm_amd as mt
    from multimodal_mtrssm_amd.linear import linear
    from multimodal_mtrssm_amd.optim import FlatParameters

    torch.manual_seed(3)
    ref = mt.MLP(37, 11, 50, 1, torch.nn.ELU)
    x = torch.randn(6, 9, 37, requires_grad=True)
    y = ref(x)  # CPU: nn.Sequential's path
    gy = torch.randn(y.shape)
    y.backward(gy)
    for flat_mode in (False, True):
        net = mt.MLP(37, 11, 50, 1, torch.nn.ELU)
        net.load_state_dict(ref.state_dict())
        net = net.to(DEV)
        flat = FlatParameters(net) if flat_mode else None
        xg = x.detach().to(DEV).requires_grad_()
        yg = net(xg)
        yg.backward(gy.to(DEV))
        np.testing.assert_allclose(_np(yg), y.detach().numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(_np(xg.grad), x.grad.numpy(), rtol=1e-5, atol=1e-6)
        for (k, p), q in zip(net.named_parameters(), ref.parameters(), strict=True):
            np.testing.assert_allclose(_np(p.grad), q.grad.numpy(), rtol=1e-5, atol=2e-6 * float(q.grad.abs().max()), err_msg=k)
        if flat is not None:
            flat.check_views()
            assert all(flat.touched)
            # a second backward accumulates (as AccumulateGrad would)
            net(xg).backward(gy.to(DEV))
            for p, q in zip(net.parameters(), ref.parameters(), strict=True):
                np.testing.assert_allclose(_np(p.grad), 2 * q.grad.numpy(), rtol=1e-5, atol=4e-6 * float(q.grad.abs().max()))
    # a slice of a weight (the hoisted halves of the scan's first layers) and an activation on the input
    w = torch.randn(20, 30, requires_grad=True)
    b = torch.randn(20, requires_grad=True)
    xx = torch.randn(17, 12, requires_grad=True)
    want = F.linear(F.elu(xx), w[:, 18:], b)
    want.square().sum().backward()
    wg, bg, xg = (t.detach().to(DEV).requires_grad_() for t in (w, b, xx))
    got = linear(xg, wg[:, 18:], bg, pre_act=2)
    got.square().sum().backward()
    np.testing.assert_allclose(_np(got), want.detach().numpy(), rtol=1e-5, atol=1e-5)
    for a_, b_ in ((wg, w), (bg, b), (xg, xx)):
        np.testing.assert_allclose(_np(a_.grad), b_.grad.numpy(), rtol=1e-5, atol=2e-6 * float(b_.grad.abs().max()))


# ---------------------------------------------------------------------------------------------
# the small streaming kernels
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(3, 5, 1, 8, 8), (2, 7, 1, 5, 3), (4, 1, 3, 16, 16)])
def test_gaussian_nll_kernel(shape: tuple[int, ...], lib_loaded: None) -> None:
    import multimodal_mtrssm_amd as mt
    from oracle.ref_model import gaussian_nll

    g = torch.Generator().manual_seed(3)
    pred = torch.randn(shape, generator=g).requires_grad_()
    tgt = torch.randn(shape, generator=g)
    want = gaussian_nll(pred, tgt, 3)
    want.backward()
    p = pred.detach().to(DEV).requires_grad_()
    got = mt.likelihood(prediction=p, target=tgt.to(DEV), event_ndims=3)
    (got * 1.0).backward()
    np.testing.assert_allclose(float(got), float(want), rtol=2e-6)
    np.testing.assert_allclose(_np(p.grad), pred.grad.numpy(), rtol=1e-6, atol=1e-8)
    with pytest.raises(ValueError, match="same shape"):
        mt.likelihood(prediction=p, target=tgt.to(DEV)[1:], event_ndims=3)
    # the decoder's Tanh folded into the kernel: likelihood(raw, out_act=Tanh) == likelihood(tanh(raw))
    raw = torch.randn(shape, generator=g).requires_grad_()
    want_t = gaussian_nll(torch.tanh(raw), tgt, 3)
    want_t.backward()
    r = raw.detach().to(DEV).requires_grad_()
    got_t = mt.likelihood(prediction=r, target=tgt.to(DEV), event_ndims=3, out_act=3)
    (got_t * 1.0).backward()
    np.testing.assert_allclose(float(got_t), float(want_t), rtol=2e-6)
    np.testing.assert_allclose(_np(r.grad), raw.grad.numpy(), rtol=2e-5, atol=1e-7)


class _WithDeadLayer(torch.nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self.a = torch.nn.Linear(13, 7)
        self.dead = torch.nn.Linear(5, 5)  # registered, never called: MMTRSSM's l_posterior / dummy transition
        self.b = torch.nn.Linear(7, 3)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.b(torch.tanh(self.a(x)))


def test_flat_adamw_matches_torch(lib_loaded: None) -> None:
    """Fused clip + AdamW over the flat buffer against torch.optim.AdamW + clip_grad_norm_, including torch's rule that a
    parameter whose .grad is None is skipped (no weight decay): the dead layer must stay bit-identical to its initial value
    (mmtrssm/mopoe_mmtrssm/core.py:143-151,188), and a scheduler's lr change must reach the device-resident rate."""
    import multimodal_mtrssm_amd as mt
    from multimodal_mtrssm_amd.optim import FlatParameters

    torch.manual_seed(0)
    net, twin = _WithDeadLayer().to(DEV), _WithDeadLayer().to(DEV)
    twin.load_state_dict(net.state_dict())
    dead0 = [p.detach().clone() for p in net.dead.parameters()]
    flat = FlatParameters(net)
    opt = mt.FlatAdamW(flat, lr=1e-2, clip_norm=0.5)
    ref = torch.optim.AdamW(twin.parameters(), lr=1e-2)
    x = torch.randn(32, 13, device=DEV)
    for it in range(6):
        if it == 3:  # what ReduceLROnPlateau does
            opt.param_groups[0]["lr"] = 5e-3
            ref.param_groups[0]["lr"] = 5e-3
        opt.zero_grad()
        net(x).square().sum().backward()
        opt.step()
        ref.zero_grad()
        twin(x).square().sum().backward()
        torch.nn.utils.clip_grad_norm_(twin.parameters(), 0.5)
        ref.step()
    for (k, a), b in zip(net.named_parameters(), twin.parameters(), strict=True):
        np.testing.assert_allclose(_np(a), _np(b), rtol=2e-5, atol=1e-6, err_msg=k)
    for p, p0, q in zip(net.dead.parameters(), dead0, twin.dead.parameters(), strict=True):
        assert torch.equal(p, p0) and torch.equal(q, p0)  # neither optimizer decays a parameter that never had a gradient
    assert float(opt.state[1]) == 6.0 and abs(float(opt.state[0]) - 5e-3) < 1e-9
    # a stock zero_grad() (set_to_none=True) breaks the view contract: the next step must refuse, not step on zeros
    net.zero_grad()
    net(x).square().sum().backward()
    with pytest.raises(RuntimeError, match="no longer aliases"):
        opt.step()


@pytest.mark.parametrize("name", ["mrssm_default", "mmtrssm_default", "mrssm_cfg2dims"])
def test_captured_train_step_matches_eager(name: str, lib_loaded: None) -> None:
    """graph.CapturedTrainStep (zero_grad + shared_step + backward + clip + AdamW recorded once as a hipGraph, replayed per
    step; uniforms drawn outside into fixed buffers) against the same steps enqueued eagerly: same losses step by step and the
    same parameters after five steps (up to the arrival order of fp32 atomics), the device-side step count advanced, the
    cluster scans' status words clean."""
    import multimodal_mtrssm_amd as mt
    from multimodal_mtrssm_amd import scan
    from multimodal_mtrssm_amd.graph import CapturedTrainStep
    from multimodal_mtrssm_amd.optim import FlatParameters

    case = with_sizes(CASES[name], 6, 9)
    oracle = build_model(case)
    batch = tuple(b.to(DEV) for b in build_batch(case))
    results = {}
    for mode in ("eager", "graph"):
        model = product_from_case(case, oracle, DEV)
        flat = FlatParameters(model, extra=8)
        dp = mt.FlatDataParallel(flat)
        # a small learning rate: with 1e-3 eight steps of a model with discrete samples amplify the arrival order of the fp32
        # atomics into flipped one-hots and the two trajectories part by 1e-3 (seen in full-suite runs, not in isolation)
        opt = mt.FlatAdamW(flat, lr=1e-5, clip_norm=10.0)
        source = dp.noise_source(seed=11)
        shapes = model.noise_shapes(6, 9)
        losses = []
        start = flat.param.clone()
        if mode == "eager":
            for _ in range(5):  # the capture's warm-up steps are undone: the graph run IS the eager run, step for step
                noise = source.draw(shapes)
                opt.zero_grad()
                out = model.shared_step(batch, noise)
                out["loss"].backward()
                dp.sync({k: out[k] for k in out})
                opt.step(grad_scale=dp.grad_scale)
                losses.append(float(out["loss"]))
        else:
            cap = CapturedTrainStep(model, flat, opt, dp, batch, source, warmup=3)
            for _ in range(5):
                losses.append(float(cap.step()["loss"]))
            assert float(opt.state[1]) == 5.0 and opt.steps == 5  # parameters, moments, step count and noise stream were restored
            cap.close()
        scan.check_cluster_status()
        moved = (flat.param - start).abs()
        assert float(moved.max()) > 3e-5  # five Adam steps of 1e-5 were applied (at most lr per step and element)
        results[mode] = (losses, flat.param.clone())
    # same uniforms, same arithmetic up to the arrival order of the fp32 atomics.  A skipped or doubled optimizer step would move
    # EVERY parameter by ~1e-5 (the mean catches it); single elements whose gradient is rounding noise may take Adam's +-lr step in
    # opposite directions in the two runs (seen: one element off by 7.7e-6 after eight steps), hence the loose maximum.
    np.testing.assert_allclose(results["graph"][0], results["eager"][0], rtol=1e-4)
    diff = (results["graph"][1] - results["eager"][1]).abs()
    assert float(diff.max()) < 2e-4 and float(diff.mean()) < 2e-7, (float(diff.max()), float(diff.mean()))


def test_set_status_word_stops_the_optimizer_and_raises(lib_loaded: None) -> None:
    """A cooperative scan launch that gave up leaves its STICKY status word set: the fused AdamW launches of that step see it on
    the device and leave parameters, moments and the step count alone; the host raises at the next poll (one step later: the
    copy is asynchronous) and `check_cluster_status` raises at once.  After `STATUS.reset()` training goes on."""
    import multimodal_mtrssm_amd as mt
    from multimodal_mtrssm_amd import scan
    from multimodal_mtrssm_amd.optim import FlatParameters

    case = with_sizes(CASES["mrssm_default"], 3, 5)  # D = H = 32: the four-CU cluster scan (a cooperative kernel with a workspace)
    model = product_from_case(case, build_model(case), DEV)
    flat = FlatParameters(model, extra=8)
    opt = mt.FlatAdamW(flat, lr=1e-3)
    batch = tuple(b.to(DEV) for b in build_batch(case))
    noise = _to(build_noise(case), DEV)

    def step() -> None:
        opt.zero_grad()
        model.shared_step(batch, noise)["loss"].backward()
        opt.step()

    step()
    torch.cuda.synchronize()
    word = scan.status_word(torch.device(DEV))
    assert word is not None and int(word.item()) == 0
    before, m_before, count = flat.param.clone(), opt.exp_avg.clone(), float(opt.state[1])
    word.fill_(7)  # what a timed-out exchange would have stored
    try:
        step()  # the device skips the update; this step's post() carries the bad word to the host
        torch.cuda.synchronize()
        assert torch.equal(flat.param, before) and torch.equal(opt.exp_avg, m_before) and float(opt.state[1]) == count
        with pytest.raises(mt._lib.MtrssmLibraryError, match="status 7"):  # noqa: SLF001
            scan.check_cluster_status()
        with pytest.raises(mt._lib.MtrssmLibraryError, match="optimizer skipped its update"):  # noqa: SLF001
            step()
    finally:
        scan.STATUS.reset()
    step()
    torch.cuda.synchronize()
    assert not torch.equal(flat.param, before) and float(opt.state[1]) == count + 1
    scan.check_cluster_status()


def test_backward_that_raises_does_not_break_later_conv_gradients(lib_loaded: None) -> None:
    """ADVICE r2 (conv.py:440): a backward that raises half-way leaves the conv gradient sink armed and partly filled; the next
    step must produce exactly the gradients of a clean run."""
    from multimodal_mtrssm_amd.optim import FlatParameters

    case = with_sizes(CASES["mrssm_nonsquare"], 3, 4)
    oracle = build_model(case)
    batch = tuple(b.to(DEV) for b in build_batch(case))
    noise = _to(build_noise(case), DEV)
    grads = {}
    for broken in (False, True):
        model = product_from_case(case, oracle, DEV)
        flat = FlatParameters(model, extra=8)
        if broken:
            out = model.shared_step(batch, noise)

            def boom(_g: torch.Tensor) -> torch.Tensor:
                raise RuntimeError("injected failure in the middle of backward")

            handle = out["kl"].register_hook(boom)  # recon's branch (decoders: conv weight gradients) runs, then this raises
            with pytest.raises(RuntimeError, match="injected failure"):
                out["loss"].backward()
            handle.remove()
        flat.zero_grad()
        out = model.shared_step(batch, noise)
        out["loss"].backward()
        torch.cuda.synchronize()
        grads[broken] = flat.grad.clone()
    diff = (grads[True] - grads[False]).abs().max()
    assert float(diff) <= 1e-5 * float(grads[False].abs().max()), float(diff)


def test_cpu_tensors_are_refused(lib_loaded: None) -> None:
    """No silent fallback: the product path raises on CPU inputs."""
    import multimodal_mtrssm_amd as mt

    case = CASES["mrssm_nonsquare"]
    model = product_from_case(case, build_model(case), "cpu")
    with pytest.raises(mt._lib.MtrssmLibraryError):  # noqa: SLF001
        model.shared_step(build_batch(case), build_noise(case))
