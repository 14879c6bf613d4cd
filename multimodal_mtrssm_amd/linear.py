"""Linear layers of the rollout path on the hand-written fp32-MFMA GEMM (``mtrssm_gemm``, ``csrc/gemm.hip``).

Replaces the ``nn.Linear`` / ``torchrl.modules.MLP`` calls around the recurrence (``networks.py:57-64,130-145``,
``core.py:132-133``, the Linear layers of ``cnn.Encoder`` / ``cnn.Decoder``) and their autograd:

* ``linear(x, weight, bias, pre_act)``  =  ``F.linear(act(x), weight, bias)`` -- the stacks are "Linear -> act -> Linear", so
  the activation is applied to the operand while it is staged (never materialised) and the backward multiplies the data
  gradient by ``act'(x)`` in the GEMM's epilogue;
* weight gradients are ``dY^T act(X)`` accumulated by the GEMM **straight into the flat gradient buffer** when the weight is
  a view of an ``optim.FlatParameters`` buffer (``grad_target``), the bias gradient being the column sums of the same
  staged operand: no temporary, no AccumulateGrad add, no separate reduction.  The autograd node then returns ``None`` for
  that input (the gradient has already been accumulated where the optimizer reads it) -- the arrangement Megatron calls
  gradient-accumulation fusion.  Outside a ``FlatParameters`` module the gradients are returned as usual.

Numerics: with the reduction in one slice each output element is a k-ordered fp32 fma chain (``v_mfma_f32_32x32x2_f32``), the
reference's fp32 nn.Linear; the large layers run bf16-piece MFMA tiles (``SPLIT_PIECES`` = 2: 16 significant bits per operand,
fp32 accumulation, like the convolutions; 3 = operands exact to 2^-24).  NOT bitwise reproducible run to run
wherever the library splits the reduction (small grids, every weight gradient): the slices meet by fp32 atomics, whose order
varies, so such sums differ in the last bits between runs (~1e-7 relative; the parity tolerances are 2e-4 of a tensor's
largest entry).  ``mtrssm_gemm``'s ``split_r = 1`` forces one slice per tile -- ordered sums, at the cost of the grid fill.
"""

from __future__ import annotations

import bisect
import ctypes as C
import weakref

import torch
from torch import Tensor

from multimodal_mtrssm_amd import _lib

# ------------------------------------------------------------------------------------------------
# gradient sinks: flat gradient buffers registered by optim.FlatParameters
# ------------------------------------------------------------------------------------------------
_SINKS: list = []  # weak references to FlatParameters
SINK_ENABLED = True


def register_sink(flat) -> None:  # noqa: ANN001
    _SINKS.append(weakref.ref(flat))


def grad_target(t: Tensor) -> Tensor | None:
    """The view of a flat gradient buffer that mirrors ``t`` (a parameter held by a ``FlatParameters``, or a strided view into
    one), or None.  Marks the parameter as touched: the caller is about to accumulate its gradient there."""
    found = grad_target_owner(t)
    return None if found is None else found[0]


def grad_target_owner(t: Tensor):  # noqa: ANN201
    """``(view, owning FlatParameters)`` for ``grad_target``'s lookup, or None."""
    if not SINK_ENABLED or not t.is_cuda or torch.is_grad_enabled():
        return None  # (backward runs with grad mode off unless create_graph: then the usual path is taken)
    addr = t.data_ptr()
    if any(ref() is None for ref in _SINKS):
        _SINKS[:] = [ref for ref in _SINKS if ref() is not None]
    for ref in _SINKS:
        flat = ref()
        if flat is None or flat.param.device != t.device:
            continue
        base = flat.param.data_ptr()
        off = (addr - base) // 4
        if addr < base or off >= flat.numel:
            continue
        i = bisect.bisect_right(flat.offsets, off) - 1
        p = flat.params[i]
        extent = sum((n - 1) * s for n, s in zip(t.shape, t.stride(), strict=True)) + 1 if t.numel() else 0
        if off + extent > flat.offsets[i] + p.numel():
            return None  # not contained in one parameter
        flat.mark_touched(i)
        return flat.grad_full.as_strided(t.shape, t.stride(), off), flat
    return None


# ------------------------------------------------------------------------------------------------
# the kernel call
# ------------------------------------------------------------------------------------------------
_TICKETS: dict = {}
_N_TICKETS = 8192


def _ld(t: Tensor) -> int:
    if t.dim() != 2 or t.stride(1) != 1 or t.dtype != torch.float32 or not t.is_cuda:  # noqa: PLR2004
        msg = f"mtrssm_gemm operands are fp32 GPU matrices with unit column stride, got shape {tuple(t.shape)} stride {t.stride()} {t.dtype} {t.device}"
        raise _lib.MtrssmLibraryError(msg)
    return int(t.stride(0)) if t.shape[0] > 1 else max(int(t.stride(0)), int(t.shape[1]))


# MFMA operand format of the LARGE GEMMs (>= SPLIT_MIN_FLOPS) when the caller does not say: 2 (default, MTRSSM_GEMM_PIECES) =
# every fp32 operand as two bf16 pieces, three bf16 MFMA products, fp32 accumulation -- 16 significant bits per operand, the
# arithmetic of the convolutions and of the wide scans -- on the 128 x {128, 64}-tile kernel of csrc/gemm_tile.h where the shape
# is made of full tiles, else the fp32 MFMA kernel; 3 = three pieces, six products (exact to 2^-24, 1.5x the time); 0 = fp32
# MFMA always.  Measured at BASELINE configs[4] dims, B = 32, T = 100, against the one-CU fp32 scan with fp32-MFMA GEMMs
# (tools/wide_pieces_error.py; the tolerances are 1e-5 on posterior probabilities, 1e-4 on the losses, 2e-4 of a gradient
# tensor's largest entry): two pieces everywhere 3.9e-7 / 5.4e-7 / 5.5e-6 with identical samples; three pieces 1.6e-7 / 4.3e-7 /
# 4.0e-6.  Everything smaller -- the scan's projections at the base dims, init_proj, the prior head, the small weight
# gradients -- runs the fp32 MFMA kernel (latency-bound launches: the operand format is not what they wait for).
import os as _os  # noqa: E402

SPLIT_PIECES = int(_os.environ.get("MTRSSM_GEMM_PIECES", "2"))
SPLIT_MIN_FLOPS = 5.0e8


def _problem(a: Tensor, b: Tensor, c: Tensor, *, a_rmajor: bool, b_rmajor: bool, bias: Tensor | None = None, zgrad: Tensor | None = None,  # noqa: PLR0913
             colsum: Tensor | None = None, act_a: int = 0, act_b: int = 0, act_z: int = 0, accumulate: bool = False, split_r: int = 0,
             mfma_split: int | None = None) -> tuple[C.Structure, float, float]:
    """The ``MtrssmGemm`` of one problem (+ its FLOPs and algorithmic bytes); shapes and layouts are checked here."""
    m, r = (a.shape[1], a.shape[0]) if a_rmajor else (a.shape[0], a.shape[1])
    n, r2 = (b.shape[1], b.shape[0]) if b_rmajor else (b.shape[0], b.shape[1])
    if r != r2 or tuple(c.shape) != (m, n):
        msg = f"gemm shapes do not agree: a' [{m}, {r}], b' [{n}, {r2}], c {tuple(c.shape)}"
        raise ValueError(msg)
    if bias is not None and bias.numel() != n or colsum is not None and colsum.numel() != m or zgrad is not None and tuple(zgrad.shape) != (m, n):
        msg = "gemm: bias must have N entries, colsum M entries, zgrad the shape of c"
        raise ValueError(msg)
    g = _lib.Gemm()
    g.A, g.B, g.C = _lib.raw_ptr(a), _lib.raw_ptr(b), _lib.raw_ptr(c)
    g.bias, g.zgrad, g.colsum = _lib.raw_ptr(bias), _lib.raw_ptr(zgrad), _lib.raw_ptr(colsum)
    g.M, g.N, g.R = m, n, r
    g.lda, g.ldb, g.ldc = _ld(a), _ld(b), _ld(c)
    g.ldz = _ld(zgrad) if zgrad is not None else 0
    g.a_rmajor, g.b_rmajor = int(a_rmajor), int(b_rmajor)
    g.act_a, g.act_b, g.act_out, g.act_z = int(act_a), int(act_b), 0, int(act_z)
    g.accumulate, g.split_r = int(accumulate), int(split_r)
    if mfma_split is None:
        mfma_split = SPLIT_PIECES if 2.0 * m * n * r >= SPLIT_MIN_FLOPS else 0
    g.mfma_split = int(mfma_split)
    if zgrad is not None and not accumulate:  # a skinny data gradient may split its long reduction: last-arriver epilogue
        tk = _TICKETS.get(a.device)
        if tk is None:
            tk = _TICKETS[a.device] = torch.zeros(_N_TICKETS, device=a.device, dtype=torch.int32)
        g.tickets, g.n_tickets = tk.data_ptr(), _N_TICKETS
    for t in (bias, colsum):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
            msg = "gemm: bias / colsum must be contiguous fp32 vectors"
            raise _lib.MtrssmLibraryError(msg)
    return g, 2.0 * m * n * r, 4.0 * (m * r + n * r + m * n)


def gemm(a: Tensor, b: Tensor, c: Tensor, **kw) -> None:  # noqa: ANN003
    """``c[i][j] (+)= (bias[j] + sum_r actA(a'(i,r)) actB(b'(j,r))) * act_z'(zgrad[i][j])`` (``include/mtrssm.h: MtrssmGemm``).
    ``a`` is ``[M, R]`` (or ``[R, M]`` when ``a_rmajor``), ``b`` is ``[N, R]`` (or ``[R, N]`` when ``b_rmajor``), ``c`` is ``[M, N]``.
    Keywords: ``a_rmajor, b_rmajor, bias, zgrad, colsum, act_a, act_b, act_z, accumulate, split_r, mfma_split``."""
    g, flops, nbytes = _problem(a, b, c, **kw)
    lib = _lib.load()
    _lib.check(_lib.TIMERS.call("mtrssm_gemm", lib.mtrssm_gemm, C.byref(g), _lib.stream_ptr(a.device), flops=flops, nbytes=nbytes), "mtrssm_gemm")


GROUP_GEMMS = True  # False: every problem of a group as its own launch (A/B runs, tests)


def _span(t: Tensor) -> tuple[int, int]:
    """Byte range a (possibly strided) tensor's elements lie in."""
    lo = t.data_ptr()
    return lo, lo + 4 * (sum((n - 1) * st for n, st in zip(t.shape, t.stride(), strict=True)) + 1 if t.numel() else 0)


def _rounds(problems: list) -> list[list]:
    """Problems whose outputs (C, column sums) may touch the same memory must not share a launch (a problem whose reduction is
    not split accumulates by plain read-modify-write): e.g. the prior head's weights collect a gradient from the initial state
    AND one from the scan.  Greedy rounds by byte range; interleaved column views of one matrix land in different rounds too."""
    rounds: list[tuple[list, list]] = []
    for problem in problems:
        (_a, _b, c), kw = problem
        spans = [_span(c)] + ([_span(kw["colsum"])] if kw.get("colsum") is not None else [])
        for members, taken in rounds:
            if all(hi <= lo2 or hi2 <= lo for lo, hi in spans for lo2, hi2 in taken):
                members.append(problem)
                taken.extend(spans)
                break
        else:
            rounds.append(([problem], list(spans)))
    return [members for members, _ in rounds]


def gemm_group(problems: list[tuple[tuple[Tensor, Tensor, Tensor], dict]]) -> None:
    """Problems ``[((a, b, c), keywords of gemm), ...]`` whose outputs are not operands of each other, in as few launches as
    their operand layouts and output overlaps allow (``mtrssm_gemm_group``; problems accumulating into the SAME memory go to
    successive launches): what the weight gradients of one backward pass are."""
    if not problems:
        return
    if not GROUP_GEMMS or len(problems) == 1:
        for (a, b, c), kw in problems:
            gemm(a, b, c, **kw)
        return
    rounds = _rounds(problems)
    if len(rounds) > 1:
        for members in rounds:
            gemm_group(members)
        return
    built = [_problem(a, b, c, **{**kw, "mfma_split": 0}) for (a, b, c), kw in problems]
    arr = (_lib.Gemm * len(built))(*[g for g, _, _ in built])
    dev = problems[0][0][0].device
    lib = _lib.load()
    _lib.check(_lib.TIMERS.call("mtrssm_gemm_group", lib.mtrssm_gemm_group, arr, len(built), _lib.stream_ptr(dev),
                                flops=sum(f for _, f, _ in built), nbytes=sum(n for _, _, n in built)), "mtrssm_gemm_group")


class _DeferredWeightGrads:
    """Weight gradients that accumulate straight into a flat gradient buffer feed nothing else in the backward pass, so they
    need not run where autograd reaches their layer: they are collected and launched SIDE BY SIDE at the end of the pass (one
    ``mtrssm_gemm_group`` call from autograd's ``queue_callback``) -- two dozen latency-bound launches of 20-40 us each become
    one or two.  The operands (saved activations, gradient tensors) stay referenced until then.  As for ``conv._ConvGradSink``:
    a backward that raises never runs the callback, so ``discard()`` re-arms at the next step and ``flush()`` is also called
    by the consumers of the flat buffer (``FlatAdamW.step``, ``FlatDataParallel.reduce``)."""

    def __init__(self) -> None:
        self.pending: list = []
        self.armed = False

    def add(self, problem: tuple) -> bool:
        if not self.armed:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self.flush)  # noqa: SLF001
            except RuntimeError:  # not inside a backward pass: nothing would ever flush
                return False
            self.armed = True
        self.pending.append(problem)
        return True

    def flush(self) -> None:
        self.armed = False
        problems, self.pending = self.pending, []
        # big problems fill the chip on their own (grouped they only share its caches); the small ones go side by side
        small = [p for p in problems if 2.0 * p[0][0].numel() * p[0][1].shape[1] < GROUP_MAX_FLOPS]
        big = [p for p in problems if 2.0 * p[0][0].numel() * p[0][1].shape[1] >= GROUP_MAX_FLOPS]
        gemm_group(small)
        for (a, b, c), kw in big:
            gemm(a, b, c, **kw)

    def discard(self) -> None:
        self.armed = False
        self.pending = []


DEFER_WEIGHT_GRADS = True
GROUP_MAX_FLOPS = 1.0e9  # per problem (2 M N R): above it a GEMM is a chip-filling launch of its own (on the tile kernel)
_DEFERRED = _DeferredWeightGrads()


def flush_deferred() -> None:
    if _DEFERRED.pending:
        _DEFERRED.flush()


def discard_deferred() -> None:
    _DEFERRED.discard()


def _pick(pieces: int | None, m: int, n: int, r: int) -> int | None:
    """A layer's operand format applies to its LARGE GEMMs only (None = the module default, decided in ``_problem``)."""
    return None if pieces is None else (int(pieces) if 2.0 * m * n * r >= SPLIT_MIN_FLOPS else 0)


def weight_grad(gy: Tensor, x: Tensor, weight: Tensor, bias: Tensor | None, *, act_x: int = 0, want_weight: bool = True,
                want_bias: bool = True, pieces: int | None = None) -> tuple[Tensor | None, Tensor | None]:
    """``dW = gy^T act(x)`` (``[N, K]``), ``db = column sums of gy`` for ``y = act(x) W^T + b`` with ``gy [M, N]``, ``x [M, K]``.
    Accumulated into the flat gradient buffer when ``weight`` / ``bias`` live in one (then None is returned for that
    tensor), else returned.  One launch for both."""
    gw_ret = gb_ret = None
    if want_weight:
        gw = grad_target(weight)
        if gw is None:
            gw = gw_ret = torch.zeros(weight.shape, device=gy.device, dtype=torch.float32)
    gb = None
    if bias is not None and want_bias:
        gb = grad_target(bias)
        if gb is None:
            gb = gb_ret = torch.zeros(bias.shape, device=gy.device, dtype=torch.float32)
    if want_weight:
        problem = ((gy, x, gw), dict(a_rmajor=True, b_rmajor=True, act_b=act_x, colsum=gb, accumulate=True,
                                     mfma_split=_pick(pieces, gy.shape[1], x.shape[1], gy.shape[0])))
        sunk = gw_ret is None and gb_ret is None  # nothing is handed back to autograd: the launch can wait for the end of the pass
        if not (DEFER_WEIGHT_GRADS and sunk and _DEFERRED.add(problem)):
            gemm(gy, x, gw, **problem[1])
    elif gb is not None:
        gb.add_(gy.sum(0))
    return gw_ret, gb_ret


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Tensor | None, pre_act: int, pieces: int | None) -> Tensor:  # noqa: ANN001
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        if x2.stride(1) != 1:
            x2 = x2.contiguous()
        w = weight if weight.stride(1) == 1 else weight.contiguous()
        y = torch.empty(x2.shape[0], w.shape[0], device=x.device, dtype=torch.float32)
        gemm(x2, w, y, a_rmajor=False, b_rmajor=False, bias=None if bias is None else bias.contiguous(), act_a=pre_act,
             mfma_split=_pick(pieces, x2.shape[0], w.shape[0], x2.shape[1]))
        ctx.save_for_backward(x2, weight, bias if bias is not None else x2.new_zeros(0))
        ctx.pre_act, ctx.has_bias, ctx.lead, ctx.pieces = pre_act, bias is not None, lead, pieces
        return y.reshape(*lead, w.shape[0])

    @staticmethod
    def backward(ctx, gy: Tensor):  # noqa: ANN001, ANN205
        x2, weight, bias = ctx.saved_tensors
        bias = bias if ctx.has_bias else None
        gy2 = gy.reshape(-1, gy.shape[-1])
        if gy2.stride(1) != 1:
            gy2 = gy2.contiguous()
        gx = None
        if ctx.needs_input_grad[0]:
            w = weight if weight.stride(1) == 1 else weight.contiguous()
            gx = torch.empty_like(x2)
            gemm(gy2, w, gx, a_rmajor=False, b_rmajor=True, zgrad=x2 if ctx.pre_act else None, act_z=ctx.pre_act,
                 mfma_split=_pick(ctx.pieces, gy2.shape[0], x2.shape[1], gy2.shape[1]))
            gx = gx.reshape(*ctx.lead, x2.shape[1])
        gw, gb = weight_grad(gy2, x2, weight, bias, act_x=ctx.pre_act, want_weight=ctx.needs_input_grad[1],
                             want_bias=ctx.has_bias and ctx.needs_input_grad[2], pieces=ctx.pieces)
        return gx, gw, gb, None, None


def linear(x: Tensor, weight: Tensor, bias: Tensor | None = None, *, pre_act: int = 0, pieces: int | None = None) -> Tensor:
    """``F.linear(act(x), weight, bias)`` over any leading dims, on the hand-written GEMMs; GPU tensors only (no fallback).
    ``pieces``: MFMA operand format of this layer's three GEMMs when they are large (None = ``SPLIT_PIECES``)."""
    return _Linear.apply(x, weight, bias, int(pre_act), pieces)
