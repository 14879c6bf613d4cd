"""autograd bindings of the HIP scan kernels (``include/mtrssm.h``).

Division of labour:

* the serial T-step recurrence runs in ONE persistent HIP launch forward and ONE backward
  (``mtrssm_*_rollout_fwd`` / ``_bwd``);
* contractions that do not depend on the recurrence -- the action / observation-embedding halves of the
  first layers -- are hoisted out of the loop as ``[B*T, in] x [in, out]`` GEMMs (``linear.linear``: the hand-written
  fp32-MFMA kernel of ``csrc/gemm.hip``);
* every weight gradient is formed AFTER the backward scan as one ``[out, B*T] x [B*T, in]`` GEMM of the same kernel
  from the per-step pre-activation gradients the scan emits, accumulated straight into the flat gradient buffer when the
  parameters live in one (``linear.grad_target``), bias gradients as column sums inside the same launch.

Replaces the loop bodies of ``mrssm/mopoe_mrssm/core.py:221-256`` and
``mmtrssm/mopoe_mmtrssm/core.py:405-490`` and their autograd graphs (~870 eager ops per step).
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch
from torch import Tensor

from multimodal_mtrssm_amd import _lib
from multimodal_mtrssm_amd.distributions import KL_BALANCE_ALPHA
from multimodal_mtrssm_amd import linear as _linear
from multimodal_mtrssm_amd.linear import gemm, gemm_group, grad_target, linear


@dataclass(frozen=True)
class ScanConfig:
    cats: int
    classes: int
    act: int
    balancing: bool = True
    rows_per_block: int = 0
    threads: int = 0

    @property
    def kl_weights(self) -> tuple[float, float]:
        return (1.0 - KL_BALANCE_ALPHA, KL_BALANCE_ALPHA) if self.balancing else (1.0, 1.0)


KERNEL_TIMERS = _lib.TIMERS
DEFER_SCAN_GRADS = True  # the scan's weight gradients join the end-of-backward group of linear._DeferredWeightGrads

# Two families of COOPERATIVE scan kernels (their workgroups wait for each other inside one launch, so every workgroup must be
# resident: a GPU shared with another process must switch them off):
# * square, mid-sized models (D = H in {32, 64, 128, 200}: BASELINE configs[1]): one batch row on a CLUSTER of four CUs with all
#   weights resident (csrc/mrssm_cluster.hip) -- MTRSSM_SCAN_CLUSTER=0 switches it off;
# * large models (D or H >= 256: BASELINE configs[4]): ALL CUs on one tile of 32 batch rows, MFMA products, a grid barrier
#   between the layers of a timestep (csrc/mrssm_wide.hip) -- MTRSSM_SCAN_WIDE=0 switches it off, MTRSSM_WIDE_PIECES = 2 | 3
#   picks the bf16 pieces per fp32 operand: 2 (default: 16 significant bits, the conv kernels' arithmetic; against the one-CU
#   fp32 scan at full size -- tools/wide_pieces_error.py -- posterior probabilities differ by <= 2.4e-6, losses by 1.2e-6
#   relative, gradients by 1.2e-5 of a tensor's max, samples identical; the stated tolerances are 1e-5 / 1e-4 / 2e-4) or 3 (exact
#   to 2^-24: 1.8e-7 / 3e-7 / 8e-7, 7 % slower at configs[4] dims).
# Otherwise one CU per row streams the weights from L2 every step (csrc/mrssm_scan.hip).
# Every cooperative launch takes its workspace from `_workspace()`: ONE buffer per (device, stream) whose first int32 is a
# STICKY status word -- a launch whose exchange gave up stores a non-zero code there, no launch clears it.  The fused AdamW
# step is handed that word and skips the update when it is set (`status_word`), and `STATUS.poll()` -- called by
# FlatAdamW.step / CapturedTrainStep.step with the previous step's asynchronously copied value, and by
# `check_cluster_status()` synchronously -- raises.
import os as _os  # noqa: E402

CLUSTER_SCAN = _os.environ.get("MTRSSM_SCAN_CLUSTER", "1") != "0"
WIDE_SCAN = _os.environ.get("MTRSSM_SCAN_WIDE", "1") != "0"
WIDE_BWD = _os.environ.get("MTRSSM_SCAN_WIDE_BWD", "1") != "0"
WIDE_PIECES = int(_os.environ.get("MTRSSM_WIDE_PIECES", "2"))
_WS: dict[tuple, Tensor] = {}
_WS_RETIRED: list[Tensor] = []


def _workspace(device: torch.device, nbytes: int) -> Tensor:
    """The cooperative kernels' workspace of the current stream of ``device`` (int64 words, >= nbytes, 256-byte aligned),
    grown on demand; the sticky status word at its head survives the growth."""
    key = (torch.device(device).index or 0, _lib.stream_ptr(device))
    ws = _WS.get(key)
    if ws is None or ws.numel() * 8 < nbytes:
        new = torch.zeros((nbytes + 7) // 8 + 32, device=device, dtype=torch.int64)
        if ws is not None:
            new[:2].copy_(ws[:2])
            _WS_RETIRED.append(ws)  # a captured hipGraph may still hold the old address
        ws = _WS[key] = new
        STATUS.watch(key, ws)
    return ws


def status_word(device: torch.device) -> Tensor | None:
    """int32[1] view of the sticky status word of the current stream's workspace (None: no cooperative launch was made yet)."""
    ws = _WS.get((torch.device(device).index or 0, _lib.stream_ptr(device)))
    return None if ws is None else ws[:1].view(torch.int32)[:1]


class StatusMonitor:
    """Host side of the sticky status words.  ``post()`` starts an asynchronous copy of every watched word into pinned host
    memory (no synchronisation); ``poll()`` raises if a copy that has COMPLETED shows a non-zero word -- so a failure surfaces
    one ``opt.step()`` late, without ever stalling the enqueueing host (the device already skipped the update on its own);
    ``check()`` synchronises and raises right away."""

    def __init__(self) -> None:
        self._words: dict[tuple, Tensor] = {}
        self._pending: list[tuple[tuple, Tensor, object]] = []

    def watch(self, key: tuple, ws: Tensor) -> None:
        self._words[key] = ws

    @staticmethod
    def _raise(key: tuple, code: int) -> None:
        msg = (f"cooperative scan kernel (workspace {key}): an exchange between workgroups timed out (status {code}); the results of "
               "that step are invalid and the optimizer skipped its update.  The GPU was probably shared, partitioned or CU-masked: "
               "set MTRSSM_SCAN_CLUSTER=0 MTRSSM_SCAN_WIDE=0")
        raise _lib.MtrssmLibraryError(msg)

    def post(self) -> None:
        for key, ws in self._words.items():
            word = ws[:1].view(torch.int32)[:1]
            if word.is_cuda:
                host = torch.empty(1, dtype=torch.int32, pin_memory=True)
                host.copy_(word, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(word.device))
            else:  # CPU tensors stand in for the device words in the host-logic tests
                host, ev = word.clone(), None
            self._pending.append((key, host, ev))

    def poll(self) -> None:
        keep = []
        for key, host, ev in self._pending:
            if ev is not None and not ev.query():
                keep.append((key, host, ev))
                continue
            code = int(host[0])
            if code:
                self._pending = []
                self._raise(key, code)
        self._pending = keep

    def check(self) -> None:
        for key, ws in self._words.items():
            code = int(ws[:1].view(torch.int32)[0].item())
            if code:
                self._raise(key, code)

    def reset(self) -> None:
        """Clear every status word (after the caller dealt with a reported failure)."""
        for ws in self._words.values():
            ws[:1].zero_()
        self._pending = []


STATUS = StatusMonitor()


def check_cluster_status() -> None:
    """Synchronises and raises if any cooperative scan launch so far gave up on an exchange (its outputs are then invalid)."""
    STATUS.check()


def _c(t: Tensor) -> Tensor:
    return t.contiguous()


def _flat2(t: Tensor) -> Tensor:
    return t.reshape(-1, t.shape[-1])


def _new(like: Tensor, *shape: int) -> Tensor:
    return torch.empty(shape, device=like.device, dtype=torch.float32)


def _opt(t: Tensor | None) -> Tensor | None:
    return None if t is None else _c(t)


class _WeightGrads:
    """Collects the ``dW[:, cols] += gy^T x`` / ``db += colsum(gy)`` problems of one backward scan; ``result(w)`` is what autograd
    gets for ``w``: None when the gradient goes straight into a flat gradient buffer, else the tensor it was accumulated in.
    The problems are independent of each other: ``flush()`` launches them side by side."""

    def __init__(self) -> None:
        self.own: dict[int, Tensor] = {}
        self.pending: list = []  # (problem, lands in a flat gradient buffer)

    def _target(self, w: Tensor, rows: slice | None, cols: slice | None) -> tuple[Tensor, bool]:
        view = w
        if rows is not None:
            view = view[rows]
        if cols is not None:
            view = view[:, cols]
        tgt = grad_target(view)
        if tgt is not None:
            return tgt, True
        full = self.own.get(id(w))
        if full is None:
            full = self.own[id(w)] = torch.zeros_like(w)
        out = full
        if rows is not None:
            out = out[rows]
        if cols is not None:
            out = out[:, cols]
        return out, False

    def add(self, gy: Tensor, x: Tensor, w: Tensor, *, cols: slice | None = None, bias: Tensor | None = None) -> None:
        """``w.grad[:, cols] += gy^T x`` with ``gy [B*T, out]``, ``x [B*T, in]`` (strided column views welcome)."""
        target, sunk = self._target(w, None, cols)
        colsum = None
        if bias is not None:
            colsum, bias_sunk = self._target(bias, None, None)
            sunk = sunk and bias_sunk
        self.pending.append((((gy, x, target), dict(a_rmajor=True, b_rmajor=True, colsum=colsum, accumulate=True)), sunk))

    def flush(self) -> None:
        """Problems that land in a flat gradient buffer join the end-of-backward group of ``linear._DeferredWeightGrads``; the
        others (their results are handed back to autograd) go here, as one grouped launch (``mtrssm_gemm_group``)."""
        now = [problem for problem, sunk in self.pending if not (DEFER_SCAN_GRADS and sunk and _linear._DEFERRED.add(problem))]  # noqa: SLF001
        gemm_group(now)
        self.pending = []

    def result(self, w: Tensor) -> Tensor | None:
        if self.pending:
            self.flush()
        return self.own.get(id(w))


def _expect(**named: tuple[Tensor | None, tuple[int, ...]]) -> None:
    """Host-side shape check of every operand before a launch (a wrong extent would fault the GPU)."""
    for name, (t, shape) in named.items():
        if t is not None and tuple(t.shape) != tuple(shape):
            msg = f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}"
            raise ValueError(msg)


# ============================================================================================
# MRSSM
# ============================================================================================
class _MrssmScan(torch.autograd.Function):
    """Posterior rollout.  Inputs: hoisted projections, initial state, uniforms, raw parameters."""

    @staticmethod
    def forward(  # noqa: PLR0913, PLR0914
        ctx, cfg: ScanConfig, xa, pa, pv, deter0, stoch0, u_post, u_prior,  # noqa: ANN001
        w1, w2, b2, wih, bih, whh, bhh, w3, b3, w4, b4, wa1, wa2, ba2, wv1, wv2, bv2,  # noqa: ANN001
    ):
        # outputs the loss does not touch get NO gradient tensor (autograd would materialise zeros: a fill kernel per unused output
        # and a read of it in the backward scan); the kernels take null pointers for absent gradients
        ctx.set_materialize_grads(False)
        lib = _lib.load()
        B, T, H = xa.shape
        D = deter0.shape[1]
        S = cfg.cats * cfg.classes
        A = w1.shape[1] - S
        K = cfg.cats
        _expect(pa=(pa, (B, T, H)), pv=(pv, (B, T, H)), deter0=(deter0, (B, D)), stoch0=(stoch0, (B, S)),
                u_post=(u_post, (B, T, K)), u_prior=(u_prior, (B, T, K)), w1=(w1, (H, A + S)), w2=(w2, (H, H)), b2=(b2, (H,)),
                wih=(wih, (3 * D, H)), bih=(bih, (3 * D,)), whh=(whh, (3 * D, D)), bhh=(bhh, (3 * D,)), w3=(w3, (H, D)),
                b3=(b3, (H,)), w4=(w4, (S, H)), b4=(b4, (S,)), wa2=(wa2, (S, H)), ba2=(ba2, (S,)), wv2=(wv2, (S, H)),
                bv2=(bv2, (S,)), wa1_rows=(wa1[:, :1], (H, 1)), wv1_rows=(wv1[:, :1], (H, 1)))
        if A < 0 or wa1.shape[1] <= D or wv1.shape[1] <= D:
            msg = "weight shapes are inconsistent with (deter, stoch, action, embed)"
            raise ValueError(msg)
        xa, pa, pv, deter0, stoch0, u_post = map(_c, (xa, pa, pv, deter0, stoch0, u_post))
        u_prior = _opt(u_prior)
        # forward layouts (include/mtrssm.h: wide outputs stream W^T, narrow outputs stream W)
        w1s_t = _c(w1[:, A:].t())
        wh1 = torch.cat([w3, wa1[:, :D], wv1[:, :D]], dim=0)  # [3H, D]
        whh_t, wh1_t = _c(whh.t()), _c(wh1.t())
        deter = _new(xa, B, T, D)
        prior_logits, post_logits, post_stoch = (_new(xa, B, T, S) for _ in range(3))
        prior_stoch = _new(xa, B, T, S) if u_prior is not None else None
        kl = _new(xa, B, T)
        need_grad = any(ctx.needs_input_grad)
        sv = dict(sv_h1=None, sv_h2=None, sv_gates=None, sv_heads=None, sv_la=None, sv_lv=None)
        if need_grad:
            sv = dict(sv_h1=_new(xa, B, T, H), sv_h2=_new(xa, B, T, H), sv_gates=_new(xa, B, T, 4 * D),
                      sv_heads=_new(xa, B, T, 3 * H), sv_la=_new(xa, B, T, S), sv_lv=_new(xa, B, T, S))
        io = _lib.fill(
            _lib.MrssmFwdIO(), xa=xa, pa=pa, pv=pv, deter0=deter0, stoch0=stoch0, u_post=u_post, u_prior=u_prior,
            deter=deter, prior_logits=prior_logits, prior_stoch=prior_stoch, post_logits=post_logits, post_stoch=post_stoch,
            kl=kl, **sv,
        )
        wp, wq = cfg.kl_weights
        dims = _lib.MrssmDims(B, T, D, H, cfg.cats, cfg.classes, cfg.act, 1, wp, wq, cfg.rows_per_block, cfg.threads)
        # algorithmic work: per (b,t) the scan reads xa,pa,pv + two uniform rows and writes deter, three
        # S-wide rows, kl and (training) the saved activations; FLOPs = 2 x every weight element touched per row-step
        per_bt = 3 * H + 2 * K + D + 3 * S + 1 + (2 * H + 4 * D + 3 * H + 2 * S if need_grad else 0)
        macs = S * H + H * H + 3 * D * H + 3 * D * D + 3 * H * D + 3 * S * H
        plain = cfg.rows_per_block == 0 and cfg.threads == 0
        cluster = CLUSTER_SCAN and plain and bool(lib.mtrssm_mrssm_cluster_supported(C.byref(dims)))
        wide = WIDE_SCAN and plain and not cluster and bool(lib.mtrssm_mrssm_wide_supported(C.byref(dims), WIDE_PIECES))
        ctx.scan_kind = "cluster" if cluster else ("wide" if wide else None)
        if cluster or wide:
            # fused GRU input path: gi = (W_ih W2) h1 + (W_ih b2 + b_ih); two small GEMMs per launch (weights change every step)
            wf_t = _new(xa, H, 3 * D)
            gemm(w2, wih, wf_t, a_rmajor=True, b_rmajor=False)            # wf_t[k][j] = sum_h W2[h][k] W_ih[j][h]
            bf = _new(xa, 1, 3 * D)
            gemm(b2.reshape(1, H), wih, bf, a_rmajor=False, b_rmajor=False, bias=_c(bih))
            cw = _lib.fill(_lib.MrssmClusterWeights(), w1s_t=w1s_t, wf_t=wf_t, bf=bf.reshape(-1), whh_t=whh_t, bhh=_c(bhh),
                           wh1_t=wh1_t, b3=_c(b3), w4=_c(w4), b4=_c(b4), wa2=_c(wa2), ba2=_c(ba2), wv2=_c(wv2), bv2=_c(bv2))
            io.sv_h2 = None  # not produced on this path: recomputed in the backward where dW_ih needs it
            stream = _lib.stream_ptr(xa.device)
            if cluster:
                ws = _workspace(xa.device, int(lib.mtrssm_mrssm_cluster_workspace_bytes(C.byref(dims))))
                _lib.check(_lib.TIMERS.call("mtrssm_mrssm_rollout_fwd_cluster", lib.mtrssm_mrssm_rollout_fwd_cluster, C.byref(dims), C.byref(cw),
                                            C.byref(io), _lib.raw_ptr(ws), ws.numel() * 8, stream,
                                            flops=2.0 * B * T * macs, nbytes=4.0 * B * T * per_bt), "mtrssm_mrssm_rollout_fwd_cluster")
            else:
                ws = _workspace(xa.device, int(lib.mtrssm_mrssm_wide_workspace_bytes(C.byref(dims), WIDE_PIECES)))
                _lib.check(_lib.TIMERS.call("mtrssm_mrssm_rollout_fwd_wide", lib.mtrssm_mrssm_rollout_fwd_wide, C.byref(dims), C.byref(cw),
                                            C.byref(io), WIDE_PIECES, _lib.raw_ptr(ws), ws.numel() * 8, stream,
                                            flops=2.0 * B * T * macs, nbytes=4.0 * B * T * per_bt), "mtrssm_mrssm_rollout_fwd_wide")
            sv["sv_h2"] = None
            ctx.cluster_w = (wf_t, whh_t, wh1_t)
        else:
            ctx.cluster_w = None
            # (the one-CU kernel takes W2 and W_ih themselves, transposed: two copies the fused-path kernels never need)
            fw = _lib.fill(
                _lib.MrssmFwdWeights(), w1s_t=w1s_t, w2_t=_c(w2.t()), b2=_c(b2), wih_t=_c(wih.t()), bih=_c(bih), whh_t=whh_t,
                bhh=_c(bhh), wh1_t=wh1_t, b3=_c(b3), w4=_c(w4), b4=_c(b4), wa2=_c(wa2), ba2=_c(ba2), wv2=_c(wv2), bv2=_c(bv2),
            )
            _lib.check(_lib.TIMERS.call("mtrssm_mrssm_rollout_fwd", lib.mtrssm_mrssm_rollout_fwd, C.byref(dims), C.byref(fw), C.byref(io),
                                        _lib.stream_ptr(xa.device), flops=2.0 * B * T * macs, nbytes=4.0 * B * T * per_bt),
                       "mtrssm_mrssm_rollout_fwd")
        if need_grad:
            ctx.cfg, ctx.A = cfg, A
            ctx.biases, ctx.w3 = (b2, bih, bhh, b3, b4, ba2, bv2), w3
            ctx.save_for_backward(deter0, stoch0, deter, prior_logits, post_logits, post_stoch, sv["sv_h1"], sv["sv_h2"],
                                  sv["sv_gates"], sv["sv_heads"], sv["sv_la"], sv["sv_lv"], w1s_t, wh1,
                                  w1, w2, wih, whh, w4, wa1, wa2, wv1, wv2)
        ctx.has_prior_stoch = prior_stoch is not None
        outs = (deter, prior_logits, post_logits, post_stoch, prior_stoch if prior_stoch is not None else deter.new_empty(()), kl)  # (placeholder, never read: no fill launch)
        return outs

    @staticmethod
    def backward(ctx, g_deter, g_prior_logits, g_post_logits, g_post_stoch, g_prior_stoch, g_kl):  # noqa: ANN001, PLR0913, PLR0914
        lib = _lib.load()
        (deter0, stoch0, deter, prior_logits, post_logits, post_stoch, sv_h1, sv_h2, sv_gates, sv_heads, sv_la, sv_lv,
         w1s_t, wh1, w1, w2, wih, whh, w4, wa1, wa2, wv1, wv2) = ctx.saved_tensors
        cfg, A = ctx.cfg, ctx.A
        B, T, D = deter.shape
        H = sv_h1.shape[-1]
        S = cfg.cats * cfg.classes
        if not ctx.has_prior_stoch:
            g_prior_stoch = None
        _expect(g_deter=(g_deter, (B, T, D)), g_prior_logits=(g_prior_logits, (B, T, S)), g_post_logits=(g_post_logits, (B, T, S)),
                g_post_stoch=(g_post_stoch, (B, T, S)), g_prior_stoch=(g_prior_stoch, (B, T, S)), g_kl=(g_kl, (B, T)))
        bw = _lib.fill(_lib.MrssmBwdWeights(), w1s_t=w1s_t, w2=_c(w2), wih=_c(wih), whh=_c(whh), wh1=wh1, w4=_c(w4), wa2=_c(wa2),
                       wv2=_c(wv2))
        g_deter0, g_stoch0 = _new(deter, B, D), _new(deter, B, S)
        d_z1, d_h2 = _new(deter, B, T, H), _new(deter, B, T, H)
        d_gi, d_gh = _new(deter, B, T, 3 * D), _new(deter, B, T, 3 * D)
        d_zh = _new(deter, B, T, 3 * H)
        d_lp, d_la, d_lv = (_new(deter, B, T, S) for _ in range(3))
        io = _lib.fill(
            _lib.MrssmBwdIO(), deter0=deter0, deter=deter, prior_logits=prior_logits, post_logits=post_logits, sv_h1=sv_h1,
            sv_h2=sv_h2, sv_gates=sv_gates, sv_heads=sv_heads, sv_la=sv_la, sv_lv=sv_lv,
            g_deter=_opt(g_deter), g_post_stoch=_opt(g_post_stoch), g_prior_stoch=_opt(g_prior_stoch),
            g_post_logits=_opt(g_post_logits), g_prior_logits=_opt(g_prior_logits), g_kl=_opt(g_kl),
            g_deter0=g_deter0, g_stoch0=g_stoch0, d_z1=d_z1, d_h2=d_h2, d_gi=d_gi, d_gh=d_gh, d_zh=d_zh, d_lp=d_lp, d_la=d_la,
            d_lv=d_lv,
        )
        wp, wq = cfg.kl_weights
        dims = _lib.MrssmDims(B, T, D, H, cfg.cats, cfg.classes, cfg.act, 1, wp, wq, cfg.rows_per_block, cfg.threads)
        per_bt = (2 * H + 4 * D + 3 * H + 2 * S) + D + 2 * S + (D + S + 1) + 2 * H + 6 * D + 3 * H + 3 * S
        macs = S * H + H * H + 3 * D * H + 3 * D * D + 3 * H * D + 3 * S * H
        cw3 = getattr(ctx, "cluster_w", None)
        kind = getattr(ctx, "scan_kind", None)
        use_cluster = kind == "cluster" and CLUSTER_SCAN and cfg.classes * cfg.cats <= 32  # noqa: PLR2004
        use_wide = kind == "wide" and WIDE_SCAN and WIDE_BWD
        if cw3 is not None and (use_cluster or use_wide):
            # the forward ran cooperatively: so does the reverse scan (same fused input path, csrc/mrssm_cluster.hip / mrssm_wide.hip)
            wf_t, whh_t, wh1_t = cw3
            cw = _lib.fill(_lib.MrssmClusterWeights(), w1s_t=w1s_t, wf_t=wf_t, whh_t=whh_t, wh1_t=wh1_t, w4=_c(w4), wa2=_c(wa2), wv2=_c(wv2))
            stream = _lib.stream_ptr(deter.device)
            if use_cluster:
                ws = _workspace(deter.device, int(lib.mtrssm_mrssm_cluster_bwd_workspace_bytes(C.byref(dims))))
                _lib.check(_lib.TIMERS.call("mtrssm_mrssm_rollout_bwd_cluster", lib.mtrssm_mrssm_rollout_bwd_cluster, C.byref(dims), C.byref(cw),
                                            C.byref(io), _lib.raw_ptr(ws), ws.numel() * 8, stream,
                                            flops=2.0 * B * T * macs, nbytes=4.0 * B * T * per_bt), "mtrssm_mrssm_rollout_bwd_cluster")
            else:
                ws = _workspace(deter.device, int(lib.mtrssm_mrssm_wide_bwd_workspace_bytes(C.byref(dims), WIDE_PIECES)))
                _lib.check(_lib.TIMERS.call("mtrssm_mrssm_rollout_bwd_wide", lib.mtrssm_mrssm_rollout_bwd_wide, C.byref(dims), C.byref(cw),
                                            C.byref(io), WIDE_PIECES, _lib.raw_ptr(ws), ws.numel() * 8, stream,
                                            flops=2.0 * B * T * macs, nbytes=4.0 * B * T * per_bt), "mtrssm_mrssm_rollout_bwd_wide")
            # d_h2 = d_gi . W_ih for all (b, t) at once (the fused input path has no h2 inside the scan)
            gemm(_flat2(d_gi), wih, _flat2(d_h2), a_rmajor=False, b_rmajor=True)
        else:
            _lib.check(_lib.TIMERS.call("mtrssm_mrssm_rollout_bwd", lib.mtrssm_mrssm_rollout_bwd, C.byref(dims), C.byref(bw), C.byref(io),
                                        _lib.stream_ptr(deter.device), flops=2.0 * B * T * macs, nbytes=4.0 * B * T * per_bt),
                       "mtrssm_mrssm_rollout_bwd")

        # ---- weight gradients: one [out, B*T] x [B*T, in] GEMM each (csrc/gemm.hip), bias gradients in the same launches -----
        prev_stoch = _flat2(torch.cat([stoch0.unsqueeze(1), post_stoch[:, :-1]], dim=1))
        prev_deter = _flat2(torch.cat([deter0.unsqueeze(1), deter[:, :-1]], dim=1))
        fz1, fh2, fgi, fgh, fzh = map(_flat2, (d_z1, d_h2, d_gi, d_gh, d_zh))
        flp, fla, flv = map(_flat2, (d_lp, d_la, d_lv))
        (b2, bih, bhh, b3, b4, ba2, bv2) = ctx.biases
        if sv_h2 is None:  # cluster forward (fused GRU input path): h2 = W2 h1 + b2 for all (b, t) at once
            sv_h2 = _new(deter, B, T, H)
            gemm(_flat2(sv_h1), w2, _flat2(sv_h2), a_rmajor=False, b_rmajor=False, bias=_c(b2))
        fdet, fh1, fsh2, fheads = map(_flat2, (deter, sv_h1, sv_h2, sv_heads))
        wg = _WeightGrads()
        wg.add(fz1, prev_stoch, w1, cols=slice(A, None))
        wg.add(fh2, fh1, w2, bias=b2)
        wg.add(fgi, fsh2, wih, bias=bih)
        wg.add(fgh, prev_deter, whh, bias=bhh)
        w3 = ctx.w3
        wg.add(fzh[:, :H], fdet, w3, bias=b3)  # prior.0 | audio.0[:, :D] | vision.0[:, :D]
        wg.add(fzh[:, H : 2 * H], fdet, wa1, cols=slice(0, D))
        wg.add(fzh[:, 2 * H :], fdet, wv1, cols=slice(0, D))
        wg.add(flp, fheads[:, :H], w4, bias=b4)
        wg.add(fla, fheads[:, H : 2 * H], wa2, bias=ba2)
        wg.add(flv, fheads[:, 2 * H :], wv2, bias=bv2)
        g_pa, g_pv = d_zh[..., H : 2 * H], d_zh[..., 2 * H :]
        r = wg.result
        return (None, d_z1, g_pa, g_pv, g_deter0, g_stoch0, None, None,
                r(w1), r(w2), r(b2), r(wih), r(bih), r(whh), r(bhh), r(w3), r(b3), r(w4), r(b4), r(wa1), r(wa2), r(ba2), r(wv1), r(wv2),
                r(bv2))


def mrssm_posterior_rollout(  # noqa: PLR0913
    transition, audio_rep, vision_rep, actions: Tensor, audio_embed: Tensor, vision_embed: Tensor,  # noqa: ANN001
    deter0: Tensor, stoch0: Tensor, u_post: Tensor, u_prior: Tensor | None, *, balancing: bool, rows_per_block: int = 0,
    threads: int = 0,
) -> dict[str, Tensor]:
    """Run ``mrssm/mopoe_mrssm/core.py:221-256`` for all T steps on the GPU."""
    factory = transition.distribution_factory
    K, Cc = factory.category_size, factory.class_size
    D = transition.deterministic_size
    A = actions.shape[-1]
    l1, l2 = transition.action_state_projector.two_layer()
    p1, p2 = transition.rnn_to_prior_projector.two_layer()
    a1, a2 = audio_rep.rnn_to_post_projector.two_layer()
    v1, v2 = vision_rep.rnn_to_post_projector.two_layer()
    act_names = {transition.action_state_projector.activation_name, transition.rnn_to_prior_projector.activation_name,
                 audio_rep.rnn_to_post_projector.activation_name, vision_rep.rnn_to_post_projector.activation_name}
    if len(act_names) != 1:
        msg = f"the HIP scan uses one activation for all heads, got {sorted(act_names)}"
        raise NotImplementedError(msg)
    cfg = ScanConfig(K, Cc, _lib.ACT_IDS[act_names.pop()], balancing, rows_per_block, threads)
    # hoisted, recurrence-independent halves of the first layers: plain library GEMMs
    xa = linear(actions, l1.weight[:, :A], l1.bias)
    pa = linear(audio_embed, a1.weight[:, D:], a1.bias)
    pv = linear(vision_embed, v1.weight[:, D:], v1.bias)
    cell = transition.rnn_cell
    deter, prior_logits, post_logits, post_stoch, prior_stoch, kl = _MrssmScan.apply(
        cfg, xa, pa, pv, deter0, stoch0, u_post, u_prior,
        l1.weight, l2.weight, l2.bias, cell.weight_ih, cell.bias_ih, cell.weight_hh, cell.bias_hh,
        p1.weight, p1.bias, p2.weight, p2.bias, a1.weight, a2.weight, a2.bias, v1.weight, v2.weight, v2.bias,
    )
    return {"deter": deter, "prior_logits": prior_logits, "post_logits": post_logits, "post_stoch": post_stoch,
            "prior_stoch": prior_stoch if u_prior is not None else None, "kl": kl}


def _st_sample(logits: Tensor, cats: int, classes: int, u: Tensor) -> tuple[Tensor, Tensor]:
    """Straight-through one-hot sample of K C-way categoricals from flat logits and given uniforms (inverse CDF on the detached
    probabilities, ``onehot + probs - probs.detach()``: distributions.MultiOneHot.rsample)."""
    from multimodal_mtrssm_amd.distributions import onehot_from_uniforms  # noqa: PLC0415

    probs = torch.softmax(logits.reshape(*logits.shape[:-1], cats, classes), dim=-1)
    onehot = onehot_from_uniforms(probs.detach(), u)
    return (onehot + (probs - probs.detach())).flatten(start_dim=-2), probs


def _mrssm_prior_rollout_differentiable(transition, actions: Tensor, deter0: Tensor, stoch0: Tensor,  # noqa: ANN001
                                        u_prior: Tensor | None) -> dict[str, Tensor]:
    """``rollout_transition`` with autograd on (``core.py:170-185`` = T times ``Transition.forward``, ``networks.py:150-173``):
    MLP([action, stoch]) -> GRUCell -> MLP -> straight-through sample, step by step.  Every Linear runs on the library's own
    GEMM with its fused epilogues (``linear.py``; the MLPs through ``networks.MLP.forward``), the GRU gate arithmetic and the
    sample are elementwise torch ops, and autograd chains them -- a composed path for the rare differentiable use (the
    reference itself only calls this method from callbacks, without gradients); the fused scan kernel serves inference."""
    factory = transition.distribution_factory
    cats, classes = factory.category_size, factory.class_size
    B, T, _ = actions.shape
    cell = transition.rnn_cell
    if u_prior is None:
        u_prior = torch.rand(B, T, cats, device=actions.device)
    actions, deter, stoch, u_prior = actions.float(), deter0.float(), stoch0.float(), u_prior.float()
    deters, logits_all, stochs = [], [], []
    for t in range(T):
        x = transition.action_state_projector(_c(torch.cat([actions[:, t], stoch], dim=-1)))
        gi = linear(x, cell.weight_ih, cell.bias_ih)
        gh = linear(_c(deter), cell.weight_hh, cell.bias_hh)
        i_r, i_z, i_n = gi.chunk(3, dim=-1)
        h_r, h_z, h_n = gh.chunk(3, dim=-1)
        r, z = torch.sigmoid(i_r + h_r), torch.sigmoid(i_z + h_z)
        deter = (1.0 - z) * torch.tanh(i_n + r * h_n) + z * deter   # torch.nn.GRUCell's update
        logits = transition.rnn_to_prior_projector(_c(deter))
        stoch, _ = _st_sample(logits, cats, classes, u_prior[:, t])
        deters.append(deter)
        logits_all.append(logits)
        stochs.append(stoch)
    return {"deter": torch.stack(deters, 1), "prior_logits": torch.stack(logits_all, 1), "prior_stoch": torch.stack(stochs, 1)}


def mrssm_prior_rollout(transition, actions: Tensor, deter0: Tensor, stoch0: Tensor, u_prior: Tensor | None,  # noqa: ANN001
                        *, rows_per_block: int = 0, threads: int = 0) -> dict[str, Tensor]:
    """``BaseRSSM.rollout_transition`` (``core.py:170-185``): the prior-only scan.  Under ``torch.no_grad()`` (how the reference's
    callbacks call it, mrssm/callback.py:184) ONE fused scan kernel; with autograd on, the step-by-step differentiable form."""
    lib = _lib.load()
    if torch.is_grad_enabled() and (actions.requires_grad or deter0.requires_grad or stoch0.requires_grad
                                    or any(p.requires_grad for p in transition.parameters())):
        return _mrssm_prior_rollout_differentiable(transition, actions, deter0, stoch0, u_prior)
    with torch.no_grad():
        factory = transition.distribution_factory
        K, Cc = factory.category_size, factory.class_size
        S = K * Cc
        B, T, A = actions.shape
        D, H = transition.deterministic_size, transition.hidden_size
        l1, l2 = transition.action_state_projector.two_layer()
        p1, p2 = transition.rnn_to_prior_projector.two_layer()
        cell = transition.rnn_cell
        if u_prior is None:
            u_prior = torch.rand(B, T, K, device=actions.device)
        xa = _c(linear(actions, l1.weight[:, :A], l1.bias))
        tensors = dict(w1s_t=_c(l1.weight[:, A:].t()), w2_t=_c(l2.weight.t()), b2=_c(l2.bias), wih_t=_c(cell.weight_ih.t()),
                       bih=_c(cell.bias_ih), whh_t=_c(cell.weight_hh.t()), bhh=_c(cell.bias_hh), wh1_t=_c(p1.weight.t()),
                       b3=_c(p1.bias), w4=_c(p2.weight), b4=_c(p2.bias))
        fw = _lib.fill(_lib.MrssmFwdWeights(), **tensors)
        deter, prior_logits, prior_stoch = _new(xa, B, T, D), _new(xa, B, T, S), _new(xa, B, T, S)
        deter0, stoch0, u_prior = _c(deter0.float()), _c(stoch0.float()), _c(u_prior.float())
        _expect(deter0=(deter0, (B, D)), stoch0=(stoch0, (B, S)), u_prior=(u_prior, (B, T, K)), xa=(xa, (B, T, H)),
                w1=(l1.weight, (H, A + S)), wih=(cell.weight_ih, (3 * D, H)), whh=(cell.weight_hh, (3 * D, D)),
                w3=(p1.weight, (H, D)), w4=(p2.weight, (S, H)))
        io = _lib.fill(_lib.MrssmFwdIO(), xa=xa, deter0=deter0, stoch0=stoch0, u_prior=u_prior, deter=deter,
                       prior_logits=prior_logits, prior_stoch=prior_stoch)
        act = _lib.ACT_IDS[transition.action_state_projector.activation_name]
        dims = _lib.MrssmDims(B, T, D, H, K, Cc, act, 0, 1.0, 1.0, rows_per_block, threads)
        _lib.check(lib.mtrssm_mrssm_rollout_fwd(C.byref(dims), C.byref(fw), C.byref(io), _lib.stream_ptr(xa.device)),
                   "mtrssm_mrssm_rollout_fwd(prior-only)")
        del tensors
    return {"deter": deter, "prior_logits": prior_logits, "prior_stoch": prior_stoch}


# ============================================================================================
# MMTRSSM
# ============================================================================================
@dataclass(frozen=True)
class MTScanConfig:
    kl_cats: int
    kl_classes: int
    kh_cats: int
    kh_classes: int
    act: int
    tau_l: float
    tau_h: float
    balancing: bool = True
    rows_per_block: int = 0
    threads: int = 0

    @property
    def kl_weights(self) -> tuple[float, float]:
        return (1.0 - KL_BALANCE_ALPHA, KL_BALANCE_ALPHA) if self.balancing else (1.0, 1.0)

    def dims(self, B: int, T: int, LD: int, HD: int, H: int, post: int) -> C.Structure:  # noqa: N803, PLR0913
        wp, wq = self.kl_weights
        return _lib.MmtrssmDims(B, T, LD, HD, H, self.kl_cats, self.kl_classes, self.kh_cats, self.kh_classes, self.act, post,
                                self.tau_l, self.tau_h, 1.0 - 1.0 / self.tau_l, 1.0 - 1.0 / self.tau_h, wp, wq,
                                self.rows_per_block, self.threads)


class _MmtrssmScan(torch.autograd.Function):
    @staticmethod
    def forward(  # noqa: PLR0913, PLR0914
        ctx, cfg: MTScanConfig, xl, pa, pv, deter_l0, deter_h0, hidden_l0, hidden_h0, stoch_l0, stoch_h0,  # noqa: ANN001
        u_post_l, u_post_h, u_prior_l, u_prior_h,  # noqa: ANN001
        wxl, wdl, wxh, wdh, bh, wlp1, blp1, wlp2, blp2, wa1, wa2, ba2, wv1, wv2, bv2, whp1, bhp1, whp2, bhp2, whq1, bhq1, whq2, bhq2,  # noqa: ANN001
    ):
        # outputs the loss does not touch get NO gradient tensor (autograd would materialise zeros: a fill kernel per unused output
        # and a read of it in the backward scan); the kernels take null pointers for absent gradients
        ctx.set_materialize_grads(False)
        lib = _lib.load()
        B, T, LD = xl.shape
        HD = deter_h0.shape[1]
        H = pa.shape[-1]
        LS, HS = cfg.kl_cats * cfg.kl_classes, cfg.kh_cats * cfg.kh_classes
        A = wxl.shape[1] - LS - HS
        _expect(pa=(pa, (B, T, H)), pv=(pv, (B, T, H)), deter_l0=(deter_l0, (B, LD)), deter_h0=(deter_h0, (B, HD)),
                hidden_l0=(hidden_l0, (B, LD)), hidden_h0=(hidden_h0, (B, HD)), stoch_l0=(stoch_l0, (B, LS)),
                stoch_h0=(stoch_h0, (B, HS)), u_post_l=(u_post_l, (B, T, cfg.kl_cats)), u_post_h=(u_post_h, (B, T, cfg.kh_cats)),
                u_prior_l=(u_prior_l, (B, T, cfg.kl_cats)), u_prior_h=(u_prior_h, (B, T, cfg.kh_cats)),
                wxl=(wxl, (LD, A + LS + HS)), wdl=(wdl, (LD, LD)), wxh=(wxh, (HD, HS)), wdh=(wdh, (HD, HD)), bh=(bh, (HD,)),
                wlp1=(wlp1, (H, LD)), blp1=(blp1, (H,)), wlp2=(wlp2, (LS, H)), blp2=(blp2, (LS,)), wa2=(wa2, (LS, H)),
                ba2=(ba2, (LS,)), wv2=(wv2, (LS, H)), bv2=(bv2, (LS,)), whp1=(whp1, (H, HD)), bhp1=(bhp1, (H,)),
                whp2=(whp2, (HS, H)), bhp2=(bhp2, (HS,)), whq1=(whq1, (H, LD + HD)), bhq1=(bhq1, (H,)), whq2=(whq2, (HS, H)),
                bhq2=(bhq2, (HS,)), wa1_rows=(wa1[:, :1], (H, 1)), wv1_rows=(wv1[:, :1], (H, 1)))
        if A < 0 or wa1.shape[1] <= LD or wv1.shape[1] <= LD:
            msg = "weight shapes are inconsistent with (ld, ls, hs, action, embed)"
            raise ValueError(msg)
        (xl, pa, pv, deter_l0, deter_h0, hidden_l0, hidden_h0, stoch_l0, stoch_h0, u_post_l, u_post_h) = map(
            _c, (xl, pa, pv, deter_l0, deter_h0, hidden_l0, hidden_h0, stoch_l0, stoch_h0, u_post_l, u_post_h))
        u_prior_l, u_prior_h = _opt(u_prior_l), _opt(u_prior_h)
        wxl_s_t = _c(wxl[:, A:].t())  # [LS+HS, LD]
        wxh_t = _c(wxh.t())  # [HS, HD]
        wl1 = torch.cat([wlp1, wa1[:, :LD], wv1[:, :LD], whq1[:, :LD]], dim=0)  # [4H, LD]
        wh1 = torch.cat([whp1, whq1[:, LD:]], dim=0)  # [2H, HD]
        bh1 = torch.cat([bhp1, bhq1], dim=0)
        tensors = dict(wxl_s_t=wxl_s_t, wdl_t=_c(wdl.t()), wxh_t=wxh_t, wdh_t=_c(wdh.t()), bh=_c(bh), wl1_t=_c(wl1.t()),
                       bl1=_c(blp1), wh1_t=_c(wh1.t()), bh1=bh1, wlp2=_c(wlp2), blp2=_c(blp2), wa2=_c(wa2), ba2=_c(ba2),
                       wv2=_c(wv2), bv2=_c(bv2), whp2=_c(whp2), bhp2=_c(bhp2), whq2=_c(whq2), bhq2=_c(bhq2))
        fw = _lib.fill(_lib.MmtrssmFwdWeights(), **tensors)
        o = dict(
            deter_l=_new(xl, B, T, LD), deter_h=_new(xl, B, T, HD), hidden_l=_new(xl, B, T, LD), hidden_h=_new(xl, B, T, HD),
            prior_logits_l=_new(xl, B, T, LS), prior_logits_h=_new(xl, B, T, HS),
            prior_stoch_l=_new(xl, B, T, LS) if u_prior_l is not None else None,
            prior_stoch_h=_new(xl, B, T, HS) if u_prior_h is not None else None,
            post_logits_l=_new(xl, B, T, LS), post_logits_h=_new(xl, B, T, HS),
            post_stoch_l=_new(xl, B, T, LS), post_stoch_h=_new(xl, B, T, HS), kl_l=_new(xl, B, T), kl_h=_new(xl, B, T),
        )
        need_grad = any(ctx.needs_input_grad)
        sv = dict(sv_l1=None, sv_h1=None, sv_la=None, sv_lv=None)
        if need_grad:
            sv = dict(sv_l1=_new(xl, B, T, 4 * H), sv_h1=_new(xl, B, T, H), sv_la=_new(xl, B, T, LS), sv_lv=_new(xl, B, T, LS))
        io = _lib.fill(
            _lib.MmtrssmFwdIO(), xl=xl, pa=pa, pv=pv, deter_l0=deter_l0, deter_h0=deter_h0, hidden_l0=hidden_l0,
            hidden_h0=hidden_h0, stoch_l0=stoch_l0, stoch_h0=stoch_h0, u_post_l=u_post_l, u_post_h=u_post_h,
            u_prior_l=u_prior_l, u_prior_h=u_prior_h, **o, **sv,
        )
        dims = cfg.dims(B, T, LD, HD, H, 1)
        macs = (LS + HS) * LD + LD * LD + HS * HD + HD * HD + 4 * H * LD + 2 * H * HD + 3 * LS * H + 2 * HS * H
        per_bt = LD + 2 * H + 2 * (LD + HD) + 2 * (LS + HS) + 2 * (LS + HS) + 2 + (5 * H + 2 * LS if need_grad else 0)
        wide = (WIDE_SCAN and cfg.rows_per_block == 0 and cfg.threads == 0
                and bool(lib.mtrssm_mmtrssm_wide_supported(C.byref(dims), WIDE_PIECES)))
        ctx.scan_kind = "wide" if wide else None
        if wide:
            ws = _workspace(xl.device, int(lib.mtrssm_mmtrssm_wide_workspace_bytes(C.byref(dims), WIDE_PIECES)))
            _lib.check(_lib.TIMERS.call("mtrssm_mmtrssm_rollout_fwd_wide", lib.mtrssm_mmtrssm_rollout_fwd_wide, C.byref(dims), C.byref(fw),
                                         C.byref(io), WIDE_PIECES, _lib.raw_ptr(ws), ws.numel() * 8, _lib.stream_ptr(xl.device),
                                         flops=2.0 * B * T * macs, nbytes=4.0 * B * T * per_bt), "mtrssm_mmtrssm_rollout_fwd_wide")
        else:
            _lib.check(_lib.TIMERS.call("mtrssm_mmtrssm_rollout_fwd", lib.mtrssm_mmtrssm_rollout_fwd, C.byref(dims), C.byref(fw), C.byref(io),
                                         _lib.stream_ptr(xl.device), flops=2.0 * B * T * macs, nbytes=4.0 * B * T * per_bt),
                       "mtrssm_mmtrssm_rollout_fwd")
        del tensors
        if need_grad:
            ctx.cfg, ctx.A = cfg, A
            ctx.biases, ctx.first_layers = (bh, blp1, blp2, ba2, bv2, bhp1, bhp2, bhq1, bhq2), (wlp1, whp1)
            ctx.save_for_backward(
                deter_l0, deter_h0, stoch_l0, stoch_h0, o["deter_l"], o["deter_h"], o["prior_logits_l"], o["prior_logits_h"],
                o["post_logits_l"], o["post_logits_h"], o["post_stoch_l"], o["post_stoch_h"], sv["sv_l1"], sv["sv_h1"],
                sv["sv_la"], sv["sv_lv"], wxl_s_t, wxh_t, wl1, wh1, wxl, wdl, wxh, wdh, wlp2, wa1, wa2, wv1, wv2, whp2, whq1, whq2)
        ctx.has_prior = (o["prior_stoch_l"] is not None, o["prior_stoch_h"] is not None)
        z = o["deter_l"].new_empty(())  # placeholder for absent prior samples, never read: no fill launch
        return (o["deter_l"], o["deter_h"], o["hidden_l"], o["hidden_h"], o["prior_logits_l"], o["prior_logits_h"],
                o["post_logits_l"], o["post_logits_h"], o["post_stoch_l"], o["post_stoch_h"],
                o["prior_stoch_l"] if o["prior_stoch_l"] is not None else z,
                o["prior_stoch_h"] if o["prior_stoch_h"] is not None else z, o["kl_l"], o["kl_h"])

    @staticmethod
    def backward(ctx, g_dl, g_dh, g_hl, g_hh, g_pll, g_plh, g_qll, g_qlh, g_qsl, g_qsh, g_psl, g_psh, g_kll, g_klh):  # noqa: ANN001, PLR0913, PLR0914, PLR0915
        lib = _lib.load()
        (deter_l0, deter_h0, stoch_l0, stoch_h0, deter_l, deter_h, prior_logits_l, prior_logits_h, post_logits_l, post_logits_h,
         post_stoch_l, post_stoch_h, sv_l1, sv_h1, sv_la, sv_lv, wxl_s_t, wxh_t, wl1, wh1, wxl, wdl, wxh, wdh, wlp2, wa1, wa2,
         wv1, wv2, whp2, whq1, whq2) = ctx.saved_tensors
        cfg, A = ctx.cfg, ctx.A
        B, T, LD = deter_l.shape
        HD = deter_h.shape[-1]
        H = sv_h1.shape[-1]
        LS, HS = cfg.kl_cats * cfg.kl_classes, cfg.kh_cats * cfg.kh_classes
        if not ctx.has_prior[0]:
            g_psl = None
        if not ctx.has_prior[1]:
            g_psh = None
        bw = _lib.fill(_lib.MmtrssmBwdWeights(), wxl_s_t=wxl_s_t, wdl=_c(wdl), wxh_t=wxh_t, wdh=_c(wdh), wl1=wl1, wh1=wh1,
                       wlp2=_c(wlp2), wa2=_c(wa2), wv2=_c(wv2), whp2=_c(whp2), whq2=_c(whq2))
        g0 = dict(g_deter_l0=_new(deter_l, B, LD), g_deter_h0=_new(deter_l, B, HD), g_hidden_l0=_new(deter_l, B, LD),
                  g_hidden_h0=_new(deter_l, B, HD), g_stoch_l0=_new(deter_l, B, LS), g_stoch_h0=_new(deter_l, B, HS))
        d = dict(d_ul=_new(deter_l, B, T, LD), d_uh=_new(deter_l, B, T, HD), d_zl1=_new(deter_l, B, T, 4 * H),
                 d_zh1=_new(deter_l, B, T, H), d_lpl=_new(deter_l, B, T, LS), d_la=_new(deter_l, B, T, LS),
                 d_lv=_new(deter_l, B, T, LS), d_lph=_new(deter_l, B, T, HS), d_lqh=_new(deter_l, B, T, HS))
        io = _lib.fill(
            _lib.MmtrssmBwdIO(), deter_l0=deter_l0, deter_h0=deter_h0, deter_l=deter_l, deter_h=deter_h,
            prior_logits_l=prior_logits_l, prior_logits_h=prior_logits_h, post_logits_l=post_logits_l, post_logits_h=post_logits_h,
            sv_l1=sv_l1, sv_h1=sv_h1, sv_la=sv_la, sv_lv=sv_lv,
            g_deter_l=_opt(g_dl), g_deter_h=_opt(g_dh), g_hidden_l=_opt(g_hl), g_hidden_h=_opt(g_hh),
            g_post_stoch_l=_opt(g_qsl), g_post_stoch_h=_opt(g_qsh), g_prior_stoch_l=_opt(g_psl), g_prior_stoch_h=_opt(g_psh),
            g_post_logits_l=_opt(g_qll), g_post_logits_h=_opt(g_qlh), g_prior_logits_l=_opt(g_pll), g_prior_logits_h=_opt(g_plh),
            g_kl_l=_opt(g_kll), g_kl_h=_opt(g_klh), **g0, **d,
        )
        dims = cfg.dims(B, T, LD, HD, H, 1)
        macs = (LS + HS) * LD + LD * LD + HS * HD + HD * HD + 4 * H * LD + 2 * H * HD + 3 * LS * H + 2 * HS * H
        per_bt = 5 * H + 2 * LS + 4 * (LD + HD) + 4 * (LS + HS) + 2 * (LD + HD) + 5 * H + 3 * LS + 2 * HS
        if getattr(ctx, "scan_kind", None) == "wide" and WIDE_SCAN and WIDE_BWD:
            ws = _workspace(deter_l.device, int(lib.mtrssm_mmtrssm_wide_bwd_workspace_bytes(C.byref(dims), WIDE_PIECES)))
            _lib.check(_lib.TIMERS.call("mtrssm_mmtrssm_rollout_bwd_wide", lib.mtrssm_mmtrssm_rollout_bwd_wide, C.byref(dims), C.byref(bw),
                                         C.byref(io), WIDE_PIECES, _lib.raw_ptr(ws), ws.numel() * 8, _lib.stream_ptr(deter_l.device),
                                         flops=2.0 * B * T * macs, nbytes=4.0 * B * T * per_bt), "mtrssm_mmtrssm_rollout_bwd_wide")
        else:
            _lib.check(_lib.TIMERS.call("mtrssm_mmtrssm_rollout_bwd", lib.mtrssm_mmtrssm_rollout_bwd, C.byref(dims), C.byref(bw), C.byref(io),
                                         _lib.stream_ptr(deter_l.device), flops=2.0 * B * T * macs, nbytes=4.0 * B * T * per_bt),
                       "mtrssm_mmtrssm_rollout_bwd")

        prev_sl = torch.cat([stoch_l0.unsqueeze(1), post_stoch_l[:, :-1]], dim=1)
        prev_sh = torch.cat([stoch_h0.unsqueeze(1), post_stoch_h[:, :-1]], dim=1)
        prev_slh = _flat2(torch.cat([prev_sl, prev_sh], dim=-1))
        prev_dl = _flat2(torch.cat([deter_l0.unsqueeze(1), deter_l[:, :-1]], dim=1))
        prev_dh = _flat2(torch.cat([deter_h0.unsqueeze(1), deter_h[:, :-1]], dim=1))
        ful, fuh, fzl, fzh = map(_flat2, (d["d_ul"], d["d_uh"], d["d_zl1"], d["d_zh1"]))
        flpl, fla, flv, flph, flqh = map(_flat2, (d["d_lpl"], d["d_la"], d["d_lv"], d["d_lph"], d["d_lqh"]))
        fdl, fdh, fl1, fh1 = map(_flat2, (deter_l, deter_h, sv_l1, sv_h1))
        (bh, blp1, blp2, ba2, bv2, bhp1, bhp2, bhq1, bhq2) = ctx.biases
        wlp1, whp1 = ctx.first_layers
        # one [out, B*T] x [B*T, in] GEMM per weight block (csrc/gemm.hip), bias gradients as column sums in the same launch
        wg = _WeightGrads()
        wg.add(ful, prev_slh, wxl, cols=slice(A, None))
        wg.add(ful, prev_dl, wdl)
        wg.add(fuh, prev_slh[:, LS:], wxh, bias=bh)
        wg.add(fuh, prev_dh, wdh)
        wg.add(fzl[:, :H], fdl, wlp1, bias=blp1)
        wg.add(fzl[:, H : 2 * H], fdl, wa1, cols=slice(0, LD))
        wg.add(fzl[:, 2 * H : 3 * H], fdl, wv1, cols=slice(0, LD))
        fzq = fzl[:, 3 * H :]
        wg.add(fzq, fdl, whq1, cols=slice(0, LD), bias=bhq1)
        wg.add(fzq, fdh, whq1, cols=slice(LD, None))
        wg.add(fzh, fdh, whp1, bias=bhp1)
        wg.add(flpl, fl1[:, :H], wlp2, bias=blp2)
        wg.add(fla, fl1[:, H : 2 * H], wa2, bias=ba2)
        wg.add(flv, fl1[:, 2 * H : 3 * H], wv2, bias=bv2)
        wg.add(flqh, fl1[:, 3 * H :], whq2, bias=bhq2)
        wg.add(flph, fh1, whp2, bias=bhp2)
        zl = d["d_zl1"]
        r = wg.result
        return (None, d["d_ul"], zl[..., H : 2 * H], zl[..., 2 * H : 3 * H], g0["g_deter_l0"], g0["g_deter_h0"], g0["g_hidden_l0"],
                g0["g_hidden_h0"], g0["g_stoch_l0"], g0["g_stoch_h0"], None, None, None, None,
                r(wxl), r(wdl), r(wxh), r(wdh), r(bh), r(wlp1), r(blp1), r(wlp2), r(blp2), r(wa1), r(wa2), r(ba2), r(wv1), r(wv2), r(bv2),
                r(whp1), r(bhp1), r(whp2), r(bhp2), r(whq1), r(bhq1), r(whq2), r(bhq2))


def _mt_act(model) -> int:  # noqa: ANN001
    names = {model.l_prior.activation_name, model.h_prior.activation_name, model.h_posterior.activation_name,
             model.audio_representation.rnn_to_post_projector.activation_name,
             model.vision_representation.rnn_to_post_projector.activation_name}
    if len(names) != 1:
        msg = f"the HIP scan uses one activation for all heads, got {sorted(names)}"
        raise NotImplementedError(msg)
    return _lib.ACT_IDS[names.pop()]


def mmtrssm_posterior_rollout(model, actions: Tensor, audio_embed: Tensor, vision_embed: Tensor, state0: dict[str, Tensor],  # noqa: ANN001
                              noise: dict[str, Tensor | None], *, rows_per_block: int = 0, threads: int = 0) -> dict[str, Tensor]:
    """Run ``mmtrssm/mopoe_mmtrssm/core.py:405-490`` for all T steps on the GPU."""
    LD = model.ld_dim
    A = actions.shape[-1]
    cfg = MTScanConfig(model.l_dist.category_size, model.l_dist.class_size, model.h_dist.category_size, model.h_dist.class_size,
                       _mt_act(model), float(model.l_rnn.tau), float(model.h_rnn.tau), bool(model.use_kl_balancing),
                       rows_per_block, threads)
    lp1, lp2 = model.l_prior.two_layer()
    hp1, hp2 = model.h_prior.two_layer()
    hq1, hq2 = model.h_posterior.two_layer()
    a1, a2 = model.audio_representation.rnn_to_post_projector.two_layer()
    v1, v2 = model.vision_representation.rnn_to_post_projector.two_layer()
    lr, hr = model.l_rnn, model.h_rnn
    xl = linear(actions, lr._input2h.weight[:, :A], lr._input2h.bias + lr._d2h.bias)  # noqa: SLF001
    pa = linear(audio_embed, a1.weight[:, LD:], a1.bias)
    pv = linear(vision_embed, v1.weight[:, LD:], v1.bias)
    bh = hr._input2h.bias + hr._d2h.bias  # noqa: SLF001
    out = _MmtrssmScan.apply(
        cfg, xl, pa, pv, state0["deter_l"], state0["deter_h"], state0["hidden_l"], state0["hidden_h"], state0["stoch_l"],
        state0["stoch_h"], noise["u_post_l"], noise["u_post_h"], noise.get("u_prior_l"), noise.get("u_prior_h"),
        lr._input2h.weight, lr._d2h.weight, hr._input2h.weight, hr._d2h.weight, bh,  # noqa: SLF001
        lp1.weight, lp1.bias, lp2.weight, lp2.bias, a1.weight, a2.weight, a2.bias, v1.weight, v2.weight, v2.bias,
        hp1.weight, hp1.bias, hp2.weight, hp2.bias, hq1.weight, hq1.bias, hq2.weight, hq2.bias,
    )
    names = ("deter_l", "deter_h", "hidden_l", "hidden_h", "prior_logits_l", "prior_logits_h", "post_logits_l", "post_logits_h",
             "post_stoch_l", "post_stoch_h", "prior_stoch_l", "prior_stoch_h", "kl_l", "kl_h")
    res = dict(zip(names, out, strict=True))
    if noise.get("u_prior_l") is None:
        res["prior_stoch_l"] = None
    if noise.get("u_prior_h") is None:
        res["prior_stoch_h"] = None
    return res


def _mtrnn_step(rnn, x: Tensor, deter: Tensor, hidden: Tensor) -> tuple[Tensor, Tensor]:  # noqa: ANN001
    """``MTRNN.forward`` (``mmtrssm/mopoe_mmtrssm/core.py:59-60``): hidden = (1 - 1/tau) hidden + (W_d d + W_x x) / tau, d = tanh(hidden)."""
    pre = linear(_c(deter), rnn._d2h.weight, rnn._d2h.bias) + linear(_c(x), rnn._input2h.weight, rnn._input2h.bias)  # noqa: SLF001
    hidden = (1.0 - 1.0 / rnn.tau) * hidden + pre / rnn.tau
    return torch.tanh(hidden), hidden


def _mmtrssm_prior_rollout_differentiable(model, actions: Tensor, state0: dict[str, Tensor],  # noqa: ANN001
                                          noise: dict[str, Tensor | None]) -> dict[str, Tensor]:
    """``MoPoE_MMTRSSM.rollout_transition`` with autograd on (``mmtrssm core.py:496-544``), step by step on the library's GEMMs --
    the composed counterpart of ``_mrssm_prior_rollout_differentiable``."""
    kl, cl = model.l_dist.category_size, model.l_dist.class_size
    kh, ch = model.h_dist.category_size, model.h_dist.class_size
    B, T, _ = actions.shape
    u_l, u_h = noise.get("u_prior_l"), noise.get("u_prior_h")
    u_l = torch.rand(B, T, kl, device=actions.device) if u_l is None else u_l.float()
    u_h = torch.rand(B, T, kh, device=actions.device) if u_h is None else u_h.float()
    st = {k: state0[k].float() for k in ("deter_l", "deter_h", "hidden_l", "hidden_h", "stoch_l", "stoch_h")}
    deter_l, deter_h, hidden_l, hidden_h, stoch_l, stoch_h = (st[k] for k in ("deter_l", "deter_h", "hidden_l", "hidden_h", "stoch_l", "stoch_h"))
    names = ("deter_l", "deter_h", "hidden_l", "hidden_h", "prior_logits_l", "prior_logits_h", "prior_stoch_l", "prior_stoch_h")
    keep: dict[str, list[Tensor]] = {k: [] for k in names}
    actions = actions.float()
    for t in range(T):
        deter_l, hidden_l = _mtrnn_step(model.l_rnn, torch.cat([actions[:, t], stoch_l, stoch_h], dim=-1), deter_l, hidden_l)
        logits_l = model.l_prior(_c(deter_l))
        deter_h, hidden_h = _mtrnn_step(model.h_rnn, stoch_h, deter_h, hidden_h)
        logits_h = model.h_prior(_c(deter_h))
        stoch_h, _ = _st_sample(logits_h, kh, ch, u_h[:, t])
        stoch_l, _ = _st_sample(logits_l, kl, cl, u_l[:, t])
        for k, v in zip(names, (deter_l, deter_h, hidden_l, hidden_h, logits_l, logits_h, stoch_l, stoch_h), strict=True):
            keep[k].append(v)
    return {k: torch.stack(v, dim=1) for k, v in keep.items()}


def mmtrssm_prior_rollout(model, actions: Tensor, state0: dict[str, Tensor], noise: dict[str, Tensor | None],  # noqa: ANN001
                          *, rows_per_block: int = 0, threads: int = 0) -> dict[str, Tensor]:
    """``MoPoE_MMTRSSM.rollout_transition`` (``core.py:496-544``): prior-only scan; one fused kernel under ``torch.no_grad()``,
    the step-by-step differentiable form with autograd on."""
    lib = _lib.load()
    if torch.is_grad_enabled() and (actions.requires_grad or any(v is not None and v.requires_grad for v in state0.values())
                                    or any(p.requires_grad for m in (model.l_rnn, model.h_rnn, model.l_prior, model.h_prior)
                                           for p in m.parameters())):
        return _mmtrssm_prior_rollout_differentiable(model, actions, state0, noise)
    with torch.no_grad():
        LD, HD = model.ld_dim, model.hd_dim
        B, T, A = actions.shape
        cfg = MTScanConfig(model.l_dist.category_size, model.l_dist.class_size, model.h_dist.category_size,
                           model.h_dist.class_size, _mt_act(model), float(model.l_rnn.tau), float(model.h_rnn.tau), True,
                           rows_per_block, threads)
        LS, HS = cfg.kl_cats * cfg.kl_classes, cfg.kh_cats * cfg.kh_classes
        lp1, lp2 = model.l_prior.two_layer()
        hp1, hp2 = model.h_prior.two_layer()
        H = lp1.out_features
        lr, hr = model.l_rnn, model.h_rnn
        xl = _c(linear(actions, lr._input2h.weight[:, :A], lr._input2h.bias + lr._d2h.bias))  # noqa: SLF001
        tensors = dict(
            wxl_s_t=_c(lr._input2h.weight[:, A:].t()), wdl_t=_c(lr._d2h.weight.t()), wxh_t=_c(hr._input2h.weight.t()),  # noqa: SLF001
            wdh_t=_c(hr._d2h.weight.t()), bh=_c(hr._input2h.bias + hr._d2h.bias), wl1_t=_c(lp1.weight.t()), bl1=_c(lp1.bias),  # noqa: SLF001
            wh1_t=_c(hp1.weight.t()), bh1=_c(hp1.bias), wlp2=_c(lp2.weight), blp2=_c(lp2.bias), whp2=_c(hp2.weight), bhp2=_c(hp2.bias))
        fw = _lib.fill(_lib.MmtrssmFwdWeights(), **tensors)
        u_l = noise.get("u_prior_l")
        u_h = noise.get("u_prior_h")
        u_l = torch.rand(B, T, cfg.kl_cats, device=xl.device) if u_l is None else _c(u_l.float())
        u_h = torch.rand(B, T, cfg.kh_cats, device=xl.device) if u_h is None else _c(u_h.float())
        o = dict(deter_l=_new(xl, B, T, LD), deter_h=_new(xl, B, T, HD), hidden_l=_new(xl, B, T, LD), hidden_h=_new(xl, B, T, HD),
                 prior_logits_l=_new(xl, B, T, LS), prior_logits_h=_new(xl, B, T, HS), prior_stoch_l=_new(xl, B, T, LS),
                 prior_stoch_h=_new(xl, B, T, HS))
        st = {k: _c(state0[k].float()) for k in ("deter_l", "deter_h", "hidden_l", "hidden_h", "stoch_l", "stoch_h")}
        _expect(deter_l=(st["deter_l"], (B, LD)), deter_h=(st["deter_h"], (B, HD)), hidden_l=(st["hidden_l"], (B, LD)),
                hidden_h=(st["hidden_h"], (B, HD)), stoch_l=(st["stoch_l"], (B, LS)), stoch_h=(st["stoch_h"], (B, HS)),
                u_prior_l=(u_l, (B, T, cfg.kl_cats)), u_prior_h=(u_h, (B, T, cfg.kh_cats)),
                wxl=(lr._input2h.weight, (LD, A + LS + HS)), wdl=(lr._d2h.weight, (LD, LD)),  # noqa: SLF001
                wxh=(hr._input2h.weight, (HD, HS)), wdh=(hr._d2h.weight, (HD, HD)), wlp1=(lp1.weight, (H, LD)),  # noqa: SLF001
                wlp2=(lp2.weight, (LS, H)), whp1=(hp1.weight, (H, HD)), whp2=(hp2.weight, (HS, H)))
        io = _lib.fill(_lib.MmtrssmFwdIO(), xl=xl, deter_l0=st["deter_l"], deter_h0=st["deter_h"], hidden_l0=st["hidden_l"],
                       hidden_h0=st["hidden_h"], stoch_l0=st["stoch_l"], stoch_h0=st["stoch_h"], u_prior_l=u_l, u_prior_h=u_h, **o)
        dims = cfg.dims(B, T, LD, HD, H, 0)
        _lib.check(lib.mtrssm_mmtrssm_rollout_fwd(C.byref(dims), C.byref(fw), C.byref(io), _lib.stream_ptr(xl.device)),
                   "mtrssm_mmtrssm_rollout_fwd(prior-only)")
        del tensors
    return o
