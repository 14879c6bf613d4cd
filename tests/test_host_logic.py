"""CPU: host-side types and argument handling (the parts of the boundary that need no GPU)."""

from __future__ import annotations

import numpy as np
import pytest
import torch

import multimodal_mtrssm_amd as mt
from oracle import ref_dists
from oracle.cases import CASES, build_batch, build_model, build_noise
from tests.conftest import product_from_case


def test_distribution_matches_oracle_definition() -> None:
    g = torch.Generator().manual_seed(0)
    logits_q, logits_p = torch.randn(3, 5, 12, generator=g), torch.randn(3, 5, 12, generator=g)
    u = torch.rand(3, 5, 3, generator=g)
    q, p = mt.MultiOneHotFactory(class_size=4, category_size=3)(logits_q), mt.MultiOneHotFactory(4, 3)(logits_p)
    oq, op = ref_dists.MultiOneHotFactory(4, 3)(logits_q), ref_dists.MultiOneHotFactory(4, 3)(logits_p)
    assert q.probs.shape == (3, 5, 3, 4)
    torch.testing.assert_close(q.probs, oq.probs)
    for bal in (True, False):
        torch.testing.assert_close(mt.kl_divergence(q.independent(1), p.independent(1), use_balancing=bal),
                                   ref_dists.kl_divergence(oq.independent(1), op.independent(1), use_balancing=bal))
    ref_dists.TAPE.clear()
    ref_dists.TAPE.push(u)
    with mt.inject_uniforms([u]):
        got = q.rsample()
    want = oq.rsample()
    assert torch.equal(got, want)
    assert got.shape == (3, 5, 12)
    assert torch.equal(got.reshape(3, 5, 3, 4).sum(-1), torch.ones(3, 5, 3))


def test_straight_through_gradient() -> None:
    logits = torch.randn(2, 6, requires_grad=True)
    d = mt.MultiOneHotFactory(class_size=3, category_size=2)(logits)
    w = torch.randn(6)
    (d.rsample() * w).sum().backward()
    probs = torch.softmax(logits.detach().reshape(2, 2, 3), -1)
    gw = w.reshape(1, 2, 3)
    want = probs * (gw - (probs * gw).sum(-1, keepdim=True))
    torch.testing.assert_close(logits.grad, want.reshape(2, 6))


def test_state_api() -> None:
    f = mt.MultiOneHotFactory(class_size=4, category_size=3)
    s = mt.State(torch.randn(2, 5, 7), f(torch.randn(2, 5, 12)))
    assert s.feature.shape == (2, 5, 19)
    assert torch.equal(s.feature, torch.cat([s.deter, s.stoch], -1))
    one = s[:, 2]
    assert one.deter.shape == (2, 7) and one.distribution.probs.shape == (2, 3, 4)
    assert len(list(iter(s))) == 2
    assert s.unsqueeze(0).deter.shape == (1, 2, 5, 7) and s.unsqueeze(0).squeeze(0).stoch.shape == (2, 5, 12)
    assert s.detach().deter.requires_grad is False and s.clone().deter.data_ptr() != s.deter.data_ptr()
    joined = mt.cat_states([s[:, :2], s[:, 2:]], dim=1)
    assert torch.equal(joined.deter, s.deter) and torch.equal(joined.distribution.probs, s.distribution.probs)
    stacked = mt.stack_states([s[:, t] for t in range(5)], dim=1)
    assert torch.equal(stacked.stoch, s.stoch) and torch.equal(stacked.distribution.logits, s.distribution.logits)


def test_mtstate_api() -> None:
    fl, fh = mt.MultiOneHotFactory(4, 4), mt.MultiOneHotFactory(class_size=2, category_size=8)
    kw = dict(deter_h=torch.randn(2, 5, 6), deter_l=torch.randn(2, 5, 7), distribution_h=fh(torch.randn(2, 5, 16)),
              distribution_l=fl(torch.randn(2, 5, 16)), hidden_h=torch.randn(2, 5, 6), hidden_l=torch.randn(2, 5, 7))
    s = mt.MTState(**kw)
    assert s.feature.shape == (2, 5, 6 + 16 + 7 + 16)
    assert torch.equal(s.feature, torch.cat([s.deter_h, s.stoch_h, s.deter_l, s.stoch_l], -1))  # mmtrssm/state.py:51
    assert s[:, 1].hidden_l.shape == (2, 7) and s[:, 1].distribution_h.probs.shape == (2, 8, 2)
    c = s.clone()
    assert torch.equal(c.distribution_h.probs, s.distribution_h.probs)  # upstream clones distribution_l here (bug, state.py:133)
    joined = mt.cat_mtstates([s[:, :3], s[:, 3:]], dim=1)
    assert torch.equal(joined.deter_l, s.deter_l) and joined.hidden_h.shape == (2, 2, 6)  # keeps the LAST hidden (state.py:237)
    stacked = mt.stack_mtstates([s[:, t] for t in range(5)], dim=1)
    assert torch.equal(stacked.stoch_h, s.stoch_h) and torch.equal(stacked.hidden_l, s.hidden_l)
    vec = mt.MTState(**{**kw, "hidden_h": torch.zeros(6), "hidden_l": torch.zeros(7)})
    assert vec[:, 0].hidden_h.shape == (6,)  # 1-d hidden is passed through untouched (state.py:78-79)


def test_constructor_validation() -> None:
    with pytest.raises(ValueError, match="2 elements"):
        mt.Representation(deterministic_size=4, hidden_size=4, obs_embed_size=4, distribution_config=[2, 2, 2])
    with pytest.raises(ValueError, match="2 elements"):
        mt.Transition(deterministic_size=4, hidden_size=4, action_size=2, distribution_config=[2], activation_name="ELU")
    with pytest.raises(AssertionError, match="tau"):
        mt.MTRNN(4, 4, tau=1.0)
    t = mt.Transition(deterministic_size=4, hidden_size=6, action_size=2, distribution_config=(3, 2), activation_name="ELU")
    assert t.rnn_cell.weight_ih.shape == (12, 6) and t.action_state_projector[0].weight.shape == (6, 8)
    assert t.distribution_factory.class_size == 3 and t.distribution_factory.category_size == 2


@pytest.mark.parametrize("name", ["mrssm_default", "mmtrssm_default"])
def test_state_dict_names_match_the_reference(name: str) -> None:
    """checkpoints interchange by name (the oracle's names were checked against the
    reference's own modules with strict=True when the fixtures were generated)."""
    case = CASES[name]
    oracle = build_model(case)
    model = product_from_case(case, oracle, "cpu")
    assert list(model.state_dict()) == list(oracle.state_dict())
    assert [k for k, _ in model.named_parameters()] == [k for k, _ in oracle.named_parameters()]
    sd = model.state_dict()
    assert sd["representation.rnn_to_post_projector.0.weight"].data_ptr() == sd["audio_representation.rnn_to_post_projector.0.weight"].data_ptr()
    assert "transition.rnn_cell.weight_ih" in sd
    if case.kind == "mmtrssm":
        assert sd["transition.action_state_projector.0.weight"].shape == (32, 2)  # the dummy Transition(A=1, S=1)
        assert {"l_rnn._d2h.weight", "h_rnn._input2h.bias", "l_posterior.0.weight", "h_posterior.2.bias"} <= set(sd)


def test_rollout_argument_errors_and_no_cpu_fallback() -> None:
    case = CASES["mrssm_nonsquare"]
    model = product_from_case(case, build_model(case), "cpu")
    batch, noise = build_batch(case), build_noise(case)
    with pytest.raises(TypeError, match="tuple"):
        model.rollout_representation(actions=batch[0], observations=batch[1], prev_state=None)
    with pytest.raises(mt._lib.MtrssmLibraryError, match="no CPU fallback"):  # noqa: SLF001
        model.shared_step(batch, noise)
    assert model.get_observations_from_batch(batch)[1] is batch[2]
    assert model.get_targets_from_batch(batch)["recon/audio"] is batch[4]
    a0, v0 = model.get_initial_observation((batch[1], batch[2]))
    assert a0.shape == batch[1][:, 0].shape and v0.shape == batch[2][:, 0].shape


def test_encoder_decoder_share_the_oracle_architecture() -> None:
    """Build-defined conv stacks: same parameters (names, shapes) as oracle/ref_cnn.py; numerics are a GPU test."""
    from oracle.ref_cnn import Decoder, Encoder

    for case in (CASES["mrssm_default"], CASES["mrssm_nonsquare"]):
        for mine, ref in ((mt.Encoder(case.dims.enc_audio), Encoder(case.dims.enc_audio)),
                          (mt.Decoder(case.dims.dec_vision), Decoder(case.dims.dec_vision))):
            got = {k: tuple(v.shape) for k, v in mine.state_dict().items()}
            want = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
            assert got == want
    enc = mt.Encoder(CASES["mrssm_default"].dims.enc_audio)
    with pytest.raises(mt._lib.MtrssmLibraryError, match="no CPU fallback"):  # noqa: SLF001
        enc(torch.zeros(1, 1, 32, 32))


def test_reference_checkpoint_round_trip(tmp_path) -> None:  # noqa: ANN001
    """A Lightning-style checkpoint ({"state_dict": ...}) written from the oracle model (the reference's parameter
    names, checked with strict=True when the fixtures were generated) loads into the product classes, aliases included."""
    from multimodal_mtrssm_amd.optim import load_reference_checkpoint

    case = CASES["mmtrssm_default"]
    oracle = build_model(case)
    path = tmp_path / "last.ckpt"
    torch.save({"state_dict": oracle.state_dict(), "epoch": 7, "global_step": 123}, path)
    torch.manual_seed(1)
    model = product_from_case(case, build_model(case), "cpu")
    with torch.no_grad():
        for p in model.parameters():
            p.add_(1.0)  # make sure the load really writes
    rest = load_reference_checkpoint(model, str(path))
    assert rest["epoch"] == 7 and rest["missing_keys"] == [] and rest["unexpected_keys"] == []
    for (k, a), (_, b) in zip(model.state_dict().items(), oracle.state_dict().items(), strict=True):
        assert torch.equal(a, b), k
    bare = tmp_path / "weights.pt"
    torch.save(oracle.state_dict(), bare)
    assert load_reference_checkpoint(model, str(bare))["missing_keys"] == []


def test_plateau_scheduler_follows_torch() -> None:
    """optim.ReduceLROnPlateau on a FlatAdamW-like optimizer = torch's scheduler on a torch optimizer (mode min,
    factor 0.5 as in default.yaml:109-114; a short patience so that the trace shows several reductions)."""
    from multimodal_mtrssm_amd.optim import ReduceLROnPlateau

    class _Opt:
        param_groups = [{"lr": 1e-3}]

    mine = ReduceLROnPlateau(_Opt(), mode="min", factor=0.5, patience=2)
    w = torch.nn.Parameter(torch.zeros(1))
    ref_opt = torch.optim.AdamW([w], lr=1e-3)
    ref = torch.optim.lr_scheduler.ReduceLROnPlateau(ref_opt, mode="min", factor=0.5, patience=2)
    g = torch.Generator().manual_seed(0)
    trace = [5.0, 4.0, 4.0, 4.0, 4.0, 3.9999, 3.0, 3.1, 3.2, 3.3, 3.4, 3.5, 3.6, 2.0] + (3 + torch.rand(20, generator=g)).tolist()
    for v in trace:
        mine.step(v)
        ref.step(v)
        assert mine.optimizer.param_groups[0]["lr"] == pytest.approx(ref_opt.param_groups[0]["lr"], rel=1e-12)
    assert mine.optimizer.param_groups[0]["lr"] < 1e-3


@pytest.mark.parametrize(("kind", "name"), [("mrssm", "mrssm_bench"), ("mmtrssm", "mmtrssm_bench")])
def test_bench_model_is_the_oracle_bench_case(kind: str, name: str) -> None:
    """bench.build_model() and oracle.cases' `*_bench` case describe the SAME network (every state-dict key and shape), so
    tests/test_gpu_parity.py::test_bench_model_matches_oracle is a test of the model bench.py times."""
    import bench

    mine = bench.build_model("cpu", kind)
    oracle = build_model(CASES[name])
    a = {k: tuple(v.shape) for k, v in mine.state_dict().items()}
    b = {k: tuple(v.shape) for k, v in oracle.state_dict().items()}
    assert a == b
    case = CASES[name]
    w = bench.WORKLOAD
    assert (case.steps, case.audio_shape, case.vision_shape) == (w["steps"], w["audio"], w["vision"])
    shapes = mine.noise_shapes(w["batch_per_gpu"], w["steps"])
    if kind == "mrssm":
        assert shapes == {"u_init": (64, 6), "u_post": (64, 50, 6)}
    else:
        assert shapes == {"u_init_h": (64, 6), "u_init_l": (64, 6), "u_post_l": (64, 50, 6), "u_post_h": (64, 50, 6)}


def test_global_row_noise_is_rank_count_invariant() -> None:
    """B = 8 on one rank sees, row for row, the uniforms 2 x 4 and 4 x 2 ranks see."""
    from multimodal_mtrssm_amd.parallel import GlobalRowNoise

    def draws(world: int) -> list[dict[str, torch.Tensor]]:
        srcs = [GlobalRowNoise(5, world, r, "cpu") for r in range(world)]
        out = []
        for _ in range(3):
            parts = [s.draw({"u_post": (8 // world, 7, 3), "u_init": (8 // world, 3)}) for s in srcs]
            out.append({k: torch.cat([p[k] for p in parts]) for k in parts[0]})
        return out

    one, two, four = draws(1), draws(2), draws(4)
    for a, b, c in zip(one, two, four, strict=True):
        for k in a:
            assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k])
    assert not torch.equal(one[0]["u_post"], one[1]["u_post"])  # successive steps draw fresh numbers
    fixed = {"u_post": torch.zeros(4, 7, 3), "u_init": torch.zeros(4, 3)}
    got = GlobalRowNoise(5, 2, 1, "cpu").draw({"u_post": (4, 7, 3), "u_init": (4, 3)}, out=fixed)
    assert got["u_post"] is fixed["u_post"] and torch.equal(fixed["u_post"], one[0]["u_post"][4:])


def test_flat_parameters_track_untouched_parameters() -> None:
    """torch.optim.AdamW skips `.grad is None` parameters; with pre-set gradient views the flat optimizer needs its own
    record of which parameters autograd ever reached (MMTRSSM: l_posterior and the dummy transition never are)."""
    from multimodal_mtrssm_amd.optim import FlatParameters

    case = CASES["mmtrssm_default"]
    model = product_from_case(case, build_model(case), "cpu")
    flat = FlatParameters(model)
    assert flat.active_mask() is not None and int(flat.active_mask().sum()) == 0
    names = {id(p): k for k, p in model.named_parameters()}
    for i, p in enumerate(flat.params):  # what backward would do on the GPU: every parameter but the dead ones
        if not names[id(p)].startswith(("l_posterior.", "transition.")):
            flat.mark_touched(i)
    mask = flat.active_mask()
    dead = sum(p.numel() for p in flat.params if names[id(p)].startswith(("l_posterior.", "transition.")))
    assert dead > 0 and int((mask == 0).sum()) >= dead
    for p, off in zip(flat.params, flat.offsets, strict=True):
        want = 0 if names[id(p)].startswith(("l_posterior.", "transition.")) else 1
        assert int(mask[off : off + p.numel()].min()) == want == int(mask[off : off + p.numel()].max())
    flat.prune_hooks()
    flat.check_views()
    model.zero_grad()
    with pytest.raises(RuntimeError, match="no longer aliases"):
        flat.check_views()


def test_failed_cooperative_scan_raises_from_the_optimizer_step() -> None:
    """VERDICT r2 item 4a / ADVICE r2 (scan.py:57): the cooperative scan kernels leave a STICKY status word; the training
    path must not go on silently.  ``FlatAdamW.step`` polls the words (the value an earlier step posted asynchronously) before
    it enqueues anything and raises ``MtrssmLibraryError``; ``check_cluster_status`` raises at once.  The device side of the
    same contract (``mtrssm_adamw_apply`` skips the update while the word is set) is tested on the GPU."""
    from multimodal_mtrssm_amd import _lib, scan
    from multimodal_mtrssm_amd.optim import FlatAdamW, FlatParameters

    monitor = scan.StatusMonitor()
    ws = torch.zeros(4, dtype=torch.int64)  # a CPU stand-in for a workspace: first int32 = the status word
    monitor.watch(("cpu", 0), ws)
    monitor.post()
    monitor.poll()  # clean: nothing raised
    ws[:1].view(torch.int32)[0] = 5  # a launch of step n gave up on an exchange
    monitor.post()  # ... step n's opt.step() posts the copy
    with pytest.raises(_lib.MtrssmLibraryError, match="timed out"):
        monitor.poll()  # ... and step n + 1's opt.step() raises
    with pytest.raises(_lib.MtrssmLibraryError, match="status 5"):
        monitor.check()
    monitor.reset()
    monitor.check()
    assert int(ws[0]) == 0

    # through the optimizer: the module-level monitor, a CPU model (the poll comes before any library call)
    net = torch.nn.Linear(3, 2)
    opt = FlatAdamW(FlatParameters(net))
    saved = (scan.STATUS._words, scan.STATUS._pending)  # noqa: SLF001
    try:
        scan.STATUS._words, scan.STATUS._pending = {("cpu", 0): ws}, []  # noqa: SLF001
        ws[:1].view(torch.int32)[0] = 9
        scan.STATUS.post()
        before = opt.flat.param.clone()
        with pytest.raises(_lib.MtrssmLibraryError, match="optimizer skipped its update"):
            opt.step()
        assert torch.equal(opt.flat.param, before) and opt.steps == 0
    finally:
        scan.STATUS._words, scan.STATUS._pending = saved  # noqa: SLF001


def test_conv_grad_sink_recovers_from_a_backward_that_raised() -> None:
    """ADVICE r2 (conv.py:440): when a backward pass raises, autograd drops the queued flush callback; the sink's one-shot
    ``pending`` flag must not stay set (later backwards would never flush: the conv weights would silently stop training)
    and the half-finished packed sums must not leak into the next step."""
    from multimodal_mtrssm_amd import conv

    sink = conv._ConvGradSink()  # noqa: SLF001
    packed = torch.ones(2, 3, 4)
    owner = torch.nn.Linear(1, 1)
    import weakref

    sink.entries[("k",)] = (packed, torch.zeros(24), weakref.ref(owner))
    sink.pending = True  # what a backward that raised leaves behind
    sink.discard()
    assert sink.pending is False and float(packed.abs().max()) == 0.0
    packed.fill_(2.0)
    sink.discard()  # nothing pending: a no-op (gradient accumulation across several backwards keeps its sums)
    assert float(packed.min()) == 2.0


def test_scalar_epilogue_and_sampled_head_host_paths_follow_the_reference_arithmetic() -> None:
    """``core._elbo`` and ``core._sampled_head`` off the GPU (the fused launches exist for CUDA tensors only): the eager arithmetic of
    ``core.py:187-221`` / ``mmtrssm core.py:563-606`` and ``factory(logits)`` + the straight-through inverse-CDF sample."""
    from multimodal_mtrssm_amd import core

    g = torch.Generator().manual_seed(3)
    nll_a, nll_v = torch.rand((), generator=g) * 100, torch.rand((), generator=g) * 100
    kl0, kl1 = torch.rand(4, 7, generator=g), torch.rand(4, 7, generator=g)
    recon, k0, k1, loss = core._elbo(nll_a, nll_v, kl0, 0.3, kl1, 0.6)  # noqa: SLF001
    torch.testing.assert_close(recon, nll_a + nll_v)
    torch.testing.assert_close(k0, kl0.mean().mul(0.3))
    torch.testing.assert_close(k1, kl1.mean().mul(0.6))
    torch.testing.assert_close(loss, nll_a + nll_v + kl0.mean().mul(0.3) + kl1.mean().mul(0.6))
    recon1, k01, _, loss1 = core._elbo(nll_a, nll_v, kl0, 0.3)  # noqa: SLF001
    torch.testing.assert_close(loss1, recon1 + k01)

    factory = mt.MultiOneHotFactory(class_size=5, category_size=6)
    logits = torch.randn(3, 30, generator=g, requires_grad=True)
    u = torch.rand(3, 6, generator=g)
    dist, stoch = core._sampled_head(factory, logits, u)  # noqa: SLF001
    ref = ref_dists.MultiOneHotFactory(5, 6)(logits.detach())
    torch.testing.assert_close(dist.probs, ref.probs)
    ref_dists.TAPE.clear()
    ref_dists.TAPE.push(u)
    torch.testing.assert_close(stoch.detach(), ref.rsample())
    weights = torch.randn(3, 30, generator=g)
    (stoch * weights).sum().backward()  # the straight-through gradient reaches the logits through the probabilities
    probs = dist.probs.detach()
    gw = weights.reshape(3, 6, 5)
    want = (probs * (gw - (probs * gw).sum(-1, keepdim=True))).reshape(3, 30)
    torch.testing.assert_close(logits.grad, want)


def test_pack_descriptor_of_a_view_that_fills_the_leading_taps_of_its_grid() -> None:
    """``conv._pack_row``: the sub-kernels of a 3 x 3 kernel run as a zero-padded 4 x 4 transposed convolution are strided views of
    the parameter with 2x2 / 2x1 / 1x2 / 1x1 taps, packed into a 2 x 2 tap grid -- descriptor words 5, 6 = the grid, 14, 15 = the
    taps the view holds (0, 0 = all of them: an ordinary weight)."""
    from multimodal_mtrssm_amd import conv

    w = torch.randn(32, 16, 3, 3)
    sub = w[:, :, 1::2, 0::2].permute(1, 0, 2, 3)   # one tap row, two tap columns
    wp, wq = torch.zeros(32, 4, 32), None
    row = conv._pack_row(sub, wp, wq, (2, 2))  # noqa: SLF001
    assert row[3:7] == [16, 32, 2, 2] and row[14:16] == [1, 2] and len(row) == 16
    assert row[7:11] == list(sub.stride())
    full = conv._pack_row(w, torch.zeros(32, 9, 16), None, None)  # noqa: SLF001
    assert full[5:7] == [3, 3] and full[14:16] == [0, 0]
