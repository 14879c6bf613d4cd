"""autograd bindings of the MFMA conv kernels (``mtrssm_conv_gather_gemm`` / ``_weight_grad``).

Two layer functions cover every convolution of the encoder / decoder stacks (``oracle/ref_cnn.py``):

* ``conv2d(x, weight, bias, stride, padding, pre_act, act, coords)``            = ``Conv2d(act?(x ++ coords))``
* ``conv_transpose2d(x, weight, bias, stride, padding, output_padding, pre_act, act)`` = ``ConvTranspose2d(act?(x))``

Forward and both data-gradient directions are ONE kernel with different gather geometry (see
``csrc/conv.hip``); weight gradients are the transposed implicit GEMM; bias gradients a channel sum.
Weights are re-packed per call into the kernel's ``[CoutPad][taps][Cpad]`` layout (zero padded) with torch
permutes -- a few KB per layer.
"""

from __future__ import annotations

import ctypes as C
import os
import weakref

import torch
from torch import Tensor

from multimodal_mtrssm_amd import _lib
from multimodal_mtrssm_amd.linear import grad_target, grad_target_owner


def _pad_to(n: int, m: int) -> int:
    return (n + m - 1) // m * m


def _pads(cout: int, cin: int) -> tuple[int, int]:
    return _pad_to(cout, 64 if cout > 32 else 32), _pad_to(cin, 16)


_PACK_BUFFERS: dict[tuple, tuple[Tensor, Tensor | None]] = {}

# Paired launches: the audio and the vision stack run the same layer on different planes; inside
# ``paired()`` the gather launches are collected instead of issued and then go out two per launch
# (``mtrssm_conv_gather_gemm_pair``), the k-th launch of the first callable with the k-th of the second.
PAIR_LAUNCH = os.environ.get("MTRSSM_PAIR_LAUNCH", "1") != "0"
_DEFER: list[tuple] | None = None


def _pack_buffers(w: Tensor, grid: tuple[int, int] | None = None) -> tuple[Tensor, Tensor | None]:
    o, i, kh, kw = w.shape
    if grid is not None:
        kh, kw = grid
    opad, ipad = _pads(o, i)
    wp = torch.zeros(opad, kh * kw, ipad, device=w.device, dtype=torch.float32)
    wq = torch.zeros(_MFMA_SPLIT, opad, kh * kw, ipad, device=w.device, dtype=torch.int16) if _MFMA_SPLIT else None
    return wp, wq


def _pack_row(w: Tensor, wp: Tensor, wq: Tensor | None, grid: tuple[int, int] | None) -> list[int]:
    """One descriptor of ``mtrssm_pack_conv_weights`` (include/mtrssm.h: MTRSSM_PACK_DESC_WORDS)."""
    o, i, kh, kw = w.shape
    gh, gw = grid if grid is not None else (kh, kw)
    return [w.data_ptr(), wp.data_ptr(), 0 if wq is None else wq.data_ptr(), o, i, gh, gw, *w.stride(), wp.shape[0], wp.shape[2],
            0 if wq is None else wq.shape[0], kh if grid is not None else 0, kw if grid is not None else 0]


def _pack_now(w: Tensor, wp: Tensor, wq: Tensor | None, grid: tuple[int, int] | None = None) -> None:
    if grid is not None and tuple(grid) != tuple(w.shape[2:]):
        # a view that fills only the leading taps of its tap grid: the table entry describes it (rare: outside a step's plan)
        table = torch.tensor([_pack_row(w, wp, wq, grid)], dtype=torch.int64).to(w.device)
        _lib.check(_lib.TIMERS.call("mtrssm_pack_conv_weights", _lib.load().mtrssm_pack_conv_weights, _lib.raw_ptr(table), 1, 8,
                                    _lib.stream_ptr(w.device)), "mtrssm_pack_conv_weights")
        return
    o, i, kh, kw = w.shape
    so, si, sh, sw = w.stride()
    _lib.check(_lib.TIMERS.call("mtrssm_pack_conv_weight", _lib.load().mtrssm_pack_conv_weight, _lib.raw_ptr(w), o, i, kh, kw, so, si,
                                sh, sw, wp.shape[0], wp.shape[2], 0 if wq is None else wq.shape[0], _lib.ptr(wp), _lib.raw_ptr(wq),
                                _lib.stream_ptr(w.device)), "mtrssm_pack_conv_weight")


class _PackPlan:
    """Every conv weight of a step packed by ONE launch (``mtrssm_pack_conv_weights``) instead of one per use (96 in the
    MoPoE-MRSSM train step).  ``begin_step()`` -- called by ``shared_step`` -- packs the weight views the previous step asked
    for, each into buffers of its own; ``pack_weight`` then finds them there.  An entry is trusted only if it was packed in
    the current epoch and the tensor's version counter has not moved since; ``invalidate()`` (called by ``FlatAdamW.step``,
    which writes parameters through raw pointers) starts a new epoch in which every use packs on the spot, as before, until
    the next ``begin_step``.  Not seen: in-place edits through ``.data`` between a ``begin_step`` and a later use in the same
    epoch (``.data`` has its own version counter) -- call ``invalidate_packs()`` after such an edit."""

    def __init__(self) -> None:
        # key -> [w (strong ref: keeps the address valid), wp, wq, version, epoch packed, epoch used, serial, tap grid]
        self.entries: dict[tuple, list] = {}
        self.epoch = 0
        self.step_epoch = -1  # the epoch begin_step opened; after invalidate() nothing is trusted until the next begin_step
        self.stream = -1  # the stream begin_step packed on: only launches on that stream may read its copies
        self.serial = 0
        self.table: Tensor | None = None
        self.table_serials: tuple = ()
        self.pinned = 0  # live captured graphs reading the packed buffers: entries are never dropped while > 0

    def invalidate(self) -> None:
        self.epoch += 1

    @staticmethod
    def key(w: Tensor, grid: tuple[int, int] | None = None) -> tuple:
        return (w.data_ptr(), tuple(w.shape), w.stride(), _MFMA_SPLIT, w.device, grid)

    def begin_step(self, device: torch.device) -> None:
        self.epoch += 1
        self.step_epoch = self.epoch
        # keep what the last step used (weights of modules that went away leave with their strong reference)
        if not self.pinned:
            self.entries = {k: e for k, e in self.entries.items() if e[5] >= self.epoch - 2}
        self.stream = torch.cuda.current_stream(device).cuda_stream
        live = [(k, e) for k, e in self.entries.items() if k[4] == device and k[3] == _MFMA_SPLIT]
        if not live:
            return
        serials = tuple(e[6] for _, e in live)
        if serials != self.table_serials or self.table is None or self.table.device != device:
            rows = [_pack_row(e[0], e[1], e[2], e[7]) for _, e in live]
            self.table = torch.tensor(rows, dtype=torch.int64).to(device)
            self.table_serials = serials
        _lib.check(_lib.TIMERS.call("mtrssm_pack_conv_weights", _lib.load().mtrssm_pack_conv_weights, _lib.raw_ptr(self.table), len(live),
                                    32, _lib.stream_ptr(device)), "mtrssm_pack_conv_weights")
        for _, e in live:
            e[3], e[4] = e[0]._version, self.epoch

    def get(self, w: Tensor, grid: tuple[int, int] | None = None) -> tuple[Tensor, Tensor | None]:
        k = self.key(w, grid)
        e = self.entries.get(k)
        if e is None:
            self.serial += 1
            e = self.entries[k] = [w, *_pack_buffers(w, grid), -1, -1, self.epoch, self.serial, grid]
        if e[4] != self.epoch or e[3] != w._version or self.epoch != self.step_epoch:
            _pack_now(w, e[1], e[2], grid)
            e[3], e[4] = w._version, self.epoch
        e[5] = self.epoch
        return e[1], e[2]


PACK_PLAN = os.environ.get("MTRSSM_PACK_PLAN", "1") != "0"
_PLAN = _PackPlan()


def begin_step(device: torch.device) -> None:
    """Start of a train / validation step: pack every conv weight the previous step used, in one launch."""
    _GRAD_SINK.discard()
    if PACK_PLAN and device.type == "cuda":
        _PLAN.begin_step(device)


def invalidate_packs() -> None:
    """Parameters were written behind autograd's back (``FlatAdamW.step``): packed copies are stale."""
    _PLAN.invalidate()


def pack_weight(w: Tensor, sub: int = 0, grid: tuple[int, int] | None = None) -> tuple[Tensor, Tensor | None]:
    """``w[O][I][kh][kw]`` (any strided view) -> zero-padded fp32 ``wp[OPad][kh*kw][IPad]`` (channel fastest) and, in a
    bf16 MFMA mode, its bf16 pieces ``wq[pieces][OPad][kh*kw][IPad]`` (``mtrssm_pack_conv_weight``).  ``grid = (KH, KW)``:
    the view holds the leading ``kh x kw`` taps of a ``KH x KW`` tap grid whose other taps are zeros.

    Inside a step announced by ``begin_step`` the copies come from the step's one pack launch (``_PackPlan``).  Otherwise
    the buffers are cached per (shape, device, stream): every use is "pack, then enqueue the kernel that reads it" on one
    stream, so a later pack of another same-shaped layer cannot overtake the earlier kernel.
    """
    o, i, kh, kw = w.shape
    if grid is not None and tuple(grid) == (kh, kw):
        grid = None
    if PACK_PLAN and w.is_cuda and kh * kw > 0 and torch.cuda.current_stream(w.device).cuda_stream == _PLAN.stream:
        return _PLAN.get(w, grid)
    slot = len(_DEFER) if _DEFER is not None else 0  # deferred (paired) launches: one buffer per pending job
    key = (o, i, kh, kw, grid, _MFMA_SPLIT, slot, sub, w.device, torch.cuda.current_stream(w.device).cuda_stream if w.is_cuda else 0)
    bufs = _PACK_BUFFERS.get(key)  # (`sub`: same-shaped weights packed for ONE pending launch, e.g. the four parity sub-kernels)
    if bufs is None:
        bufs = _PACK_BUFFERS[key] = _pack_buffers(w, grid)
    if kh * kw > 0:
        _pack_now(w, *bufs, grid)
    return bufs


# MFMA operand format of the patch-staged kernels (``MtrssmConvGeom.mfma_split``, include/mtrssm.h)
MFMA_MODES = {"f32": 0, "bf16x3": 3, "bf16x2": 2, "bf16": 1}
_MFMA_SPLIT = MFMA_MODES[os.environ.get("MTRSSM_CONV_MFMA", "bf16x2")]


def set_mfma_mode(mode: str) -> None:
    """``"bf16x2"`` (default): every fp32 operand as two bf16 pieces (16 significant bits), the three products
    ``p0 q0 + p0 q1 + p1 q0`` as bf16 MFMAs with fp32 accumulation; against the golden fixtures: losses 7e-7 relative,
    posterior 1.3e-7 absolute, gradients 8e-6 of the tensor's max -- the fp32 MFMA kernels' own figures are 7e-7 / 0 / 3e-6
    (tests/mode_errors.py), two orders inside the parity tolerances.  ``"bf16x3"``: three pieces, six products (~2^-24:
    indistinguishable from fp32).  ``"f32"``: fp32 MFMA (bitwise an fma chain).  ``"bf16"``: plain bf16 operands
    (posterior 7e-5, gradients 7e-3: outside the 1e-5 posterior tolerance)."""
    global _MFMA_SPLIT  # noqa: PLW0603
    if mode not in MFMA_MODES:
        msg = f"unknown MFMA mode {mode!r}; choose from {sorted(MFMA_MODES)}"
        raise ValueError(msg)
    _MFMA_SPLIT = MFMA_MODES[mode]


def mfma_mode() -> str:
    return next(k for k, v in MFMA_MODES.items() if v == _MFMA_SPLIT)


def gemm_pieces() -> int:
    """bf16 pieces per operand for the large Linear layers INSIDE the conv stacks (cnn.Encoder head, cnn.Decoder stem): the
    arithmetic of the convolutions around them (``bf16x2`` -> 2, ``bf16x3`` -> 3, ``f32`` -> 0 = fp32 MFMA, ``bf16`` -> 2)."""
    return {0: 0, 3: 3, 2: 2, 1: 2}[_MFMA_SPLIT]


def _geom(**kw: int) -> C.Structure:
    g = _lib.ConvGeom()
    for k, v in kw.items():
        setattr(g, k, int(v))
    g.mfma_split = _MFMA_SPLIT
    return g


def _gather_gemm(geom: C.Structure, src: Tensor, src2: Tensor | None, wpq: tuple[Tensor, Tensor | None], bias: Tensor | None,
                 actgrad_in: Tensor | None, out: Tensor, add_in: Tensor | None = None) -> None:
    wp, wq = wpq
    lib = _lib.load()
    # algorithmic work of this launch: 2 FLOPs per (output element, tap, real channel); bytes = source read once + output written once
    job = (geom, src, src2, wp, wq, bias, actgrad_in, add_in, out)
    if _DEFER is not None:
        _DEFER.append(job)
        return
    _launch_gather(job)


def _job_work(job: tuple) -> tuple[float, float]:
    geom, actgrad_in, add_in = job[0], job[6], job[7]
    pixels = geom.N * geom.Hq * geom.Wq
    flops = 2.0 * pixels * geom.Cout * geom.KH * geom.KW * (geom.C + geom.C2)
    # source once + output once + each epilogue operand (act' input, skip / skip gradient) once
    nbytes = 4.0 * (geom.N * geom.C * geom.Hs * geom.Ws + pixels * geom.Cout * (1 + (actgrad_in is not None) + (add_in is not None)))
    return flops, nbytes


def _job_args(job: tuple) -> tuple:
    geom, src, src2, wp, wq, bias, actgrad_in, add_in, out = job
    return (C.byref(geom), _lib.ptr(src), _lib.ptr(src2), _lib.ptr(wp), _lib.raw_ptr(wq), _lib.ptr(bias), _lib.ptr(actgrad_in),
            _lib.ptr(add_in), _lib.ptr(out))


def _quad_args(job: tuple) -> tuple:
    _, geoms, y, wqs, bias, out, actgrad = job
    arr = (C.c_void_p * 4)(*[_lib.raw_ptr(wq) for wq in wqs])
    return (geoms, _lib.ptr(y), arr, _lib.ptr(bias), _lib.ptr(actgrad), _lib.ptr(out))


def _quad_work(job: tuple) -> tuple[float, float]:
    g = job[1][0]
    pixels = g.N * g.Ho * g.Wo
    return 2.0 * pixels * g.Cout * 4 * g.C, 4.0 * (g.N * g.C * g.Hs * g.Ws + pixels * g.Cout * (2 if job[6] is not None else 1))


def _launch_quad(ja: tuple, jb: tuple | None) -> None:
    lib = _lib.load()
    fa, ba = _quad_work(ja)
    fb, bb = _quad_work(jb) if jb is not None else (0.0, 0.0)
    args_b = _quad_args(jb) if jb is not None else (None, None, None, None, None, None)
    _lib.check(_lib.TIMERS.call("mtrssm_convt_quad", lib.mtrssm_convt_quad, *_quad_args(ja), *args_b, _lib.stream_ptr(ja[2].device),
                                flops=fa + fb, nbytes=ba + bb), "mtrssm_convt_quad")


def _launch_gather(job: tuple) -> None:
    if job[0] == "quad":
        _launch_quad(job, None)
        return
    lib = _lib.load()
    flops, nbytes = _job_work(job)
    _lib.check(_lib.TIMERS.call("mtrssm_conv_gather_gemm", lib.mtrssm_conv_gather_gemm, *_job_args(job),
                                _lib.stream_ptr(job[1].device), flops=flops, nbytes=nbytes), "mtrssm_conv_gather_gemm")


def paired(fn_a, fn_b):  # noqa: ANN001, ANN201
    """``(fn_a(), fn_b())`` with their gather launches merged two by two (same count and order in both callables)."""
    global _DEFER  # noqa: PLW0603
    if not PAIR_LAUNCH or _DEFER is not None:
        return fn_a(), fn_b()
    _DEFER = []
    try:
        ra = fn_a()
        na = len(_DEFER)
        rb = fn_b()
        jobs = _DEFER
    finally:
        _DEFER = None
    if na * 2 != len(jobs):
        for job in jobs:
            _launch_gather(job)
        return ra, rb
    lib = _lib.load()
    for ja, jb in zip(jobs[:na], jobs[na:], strict=True):
        qa, qb = ja[0] == "quad", jb[0] == "quad"
        if qa or qb:
            if qa and qb and lib.mtrssm_convt_quad_supported(ja[1]) == lib.mtrssm_convt_quad_supported(jb[1]):
                _launch_quad(ja, jb)
            else:
                _launch_gather(ja)
                _launch_gather(jb)
            continue
        if not lib.mtrssm_conv_gather_pair_merges(C.byref(ja[0]), C.byref(jb[0]), int(ja[4] is not None and jb[4] is not None)):
            _launch_gather(ja)  # thin / fp32-kernel layers: nothing to merge
            _launch_gather(jb)
            continue
        fa, ba = _job_work(ja)
        fb, bb = _job_work(jb)
        _lib.check(_lib.TIMERS.call("mtrssm_conv_gather_gemm_pair", lib.mtrssm_conv_gather_gemm_pair, *_job_args(ja), *_job_args(jb),
                                    _lib.stream_ptr(ja[1].device), flops=fa + fb, nbytes=ba + bb), "mtrssm_conv_gather_gemm_pair")
    return ra, rb


def _conv_forward_gather(x: Tensor, coords: Tensor | None, w: Tensor, bias: Tensor | None, stride: int, pad: int,  # noqa: PLR0913
                         pre_act: bool, act: int, actgrad_in: Tensor | None = None, add_in: Tensor | None = None) -> Tensor:  # noqa: FBT001
    """``out = (bias + Conv2d_{w}(pre(x ++ coords))) * act'(actgrad_in)``; ``w[O][I][k][k]``."""
    n, c, hs, ws = x.shape
    o, i, kh, kw = w.shape
    c2 = 0 if coords is None else coords.shape[0]
    if i != c + c2:
        msg = f"weight expects {i} input channels, got {c}+{c2}"
        raise ValueError(msg)
    ho = (hs + 2 * pad - kh) // stride + 1
    wo = (ws + 2 * pad - kw) // stride + 1
    wpq = pack_weight(w)
    wp = wpq[0]
    out = torch.empty(n, o, ho, wo, device=x.device, dtype=torch.float32)
    geom = _geom(N=n, C=c, Hs=hs, Ws=ws, C2=c2, Cpad=wp.shape[2], KH=kh, KW=kw, SS=stride, TS=1, OFFY=-pad, OFFX=-pad,
                 Hq=ho, Wq=wo, OS=1, QY=0, QX=0, Ho=ho, Wo=wo, Cout=o, CoutPad=wp.shape[0], pre_act=int(pre_act), act=act)
    _gather_gemm(geom, x, coords, wpq, bias, actgrad_in, out, add_in)
    return out


def _conv_transposed_gather(y: Tensor, w: Tensor, bias: Tensor | None, stride: int, pad: int, out_hw: tuple[int, int],  # noqa: PLR0913
                            pre_act: bool, act: int, actgrad_in: Tensor | None = None, add_in: Tensor | None = None) -> Tensor:  # noqa: FBT001
    """``out[n,c,iy,ix] = (bias[c] + sum_{o,ky,kx} w[o][c][ky][kx] pre(y)[n,o,(iy+p-ky)/s,(ix+p-kx)/s]) * act'(actgrad_in)``.

    = Conv2d backward-data (``y`` = dOut, ``w`` the conv weight) = ConvTranspose2d forward (``y`` = input, ``w`` its weight).
    One launch per output parity class; each visits only the taps that reach it.
    """
    n, o, hs, ws = y.shape
    o2, c, kh, kw = w.shape
    if o2 != o:
        msg = f"weight expects {o2} gathered channels, got {o}"
        raise ValueError(msg)
    ho, wo = out_hw
    out = torch.empty(n, c, ho, wo, device=y.device, dtype=torch.float32)
    if (stride == 2 and kh == 4 and kw == 4 and pad == 1 and c <= 2 and (ho, wo) == (2 * hs, 2 * ws) and actgrad_in is None
            and add_in is None and o * ((8 + 2) * (32 + 2) + 1) * 4 <= 64 * 1024):
        # the decoders' last layer: all four parity classes in one pass (csrc/conv.hip: convt_k4s2_thin_kernel)
        lib = _lib.load()
        wc = w.contiguous()
        if _MFMA_SPLIT == 2 and CONVT_BAND and lib.mtrssm_convt_k4s2_band_supported(n, o, hs, ws, c):  # noqa: PLR2004
            # the reference's shapes in the default operand format: on the MFMA, a frame staged once (csrc/conv_s2_band.h)
            _lib.check(_lib.TIMERS.call(
                "mtrssm_convt_k4s2_band", lib.mtrssm_convt_k4s2_band, n, o, hs, ws, c, _lib.ptr(y), _lib.ptr(wc), _lib.ptr(bias),
                int(pre_act), act, _lib.ptr(out), _lib.stream_ptr(y.device), flops=2.0 * n * ho * wo * c * 4 * o,
                nbytes=4.0 * (y.numel() + out.numel())), "mtrssm_convt_k4s2_band")
            return out
        _lib.check(_lib.TIMERS.call(
            "mtrssm_convt_k4s2_thin", lib.mtrssm_convt_k4s2_thin, n, o, hs, ws, c, _lib.ptr(y), _lib.ptr(wc), _lib.ptr(bias),
            int(pre_act), act, _lib.ptr(out), _lib.stream_ptr(y.device), flops=2.0 * n * ho * wo * c * 4 * o,
            nbytes=4.0 * (y.numel() + out.numel())), "mtrssm_convt_k4s2_thin")
        return out
    quad_fwd = kh == 4 and kw == 4 and actgrad_in is None and (o, c, hs * ws) in ((64, 32, 64), (32, 16, 256))  # noqa: PLR2004
    # The same kernel takes the backward-data of a Conv2d(k = 3, s = 2, p = 1) as the transposed convolution of its kernel
    # zero-padded to 4 x 4 (the encoders' third conv: 4 paired launches of ~54 us -> one of ~90): the parity-class sub-kernels
    # are strided views of the PARAMETER with 2 x 2, 2 x 1, 1 x 2 and 1 x 1 taps, packed into a 2 x 2 tap grid (`grid`), so the
    # step's pack plan learns them like every other weight view.
    quad_bwd = (CONVT_QUAD_BWD and kh == 3 and kw == 3 and actgrad_in is not None and bias is None  # noqa: PLR2004
                and (o, c, hs * ws) in ((32, 16, 64), (16, 8, 256)))
    if (stride == 2 and pad == 1 and (ho, wo) == (2 * hs, 2 * ws) and add_in is None and _MFMA_SPLIT == 2 and CONVT_QUAD  # noqa: PLR2004
            and (quad_fwd or quad_bwd)):
        # the decoders' ConvTranspose layers (and the backward-data of the encoders' third conv = the same transposed conv with
        # its 3 x 3 kernel zero-padded to 4 x 4): all four parity classes in one pass over the source (convt_quad_resident_kernel)
        geoms = (_lib.ConvGeom * 4)()
        wqs = []
        for q in range(4):
            qy, qx = q >> 1, q & 1
            ky0, kx0 = (qy + pad) % stride, (qx + pad) % stride
            wsub = w[:, :, ky0::stride, kx0::stride].permute(1, 0, 2, 3)  # [c][o][2 or 1][2 or 1]
            wp, wq = pack_weight(wsub, sub=q, grid=(2, 2))
            geoms[q] = _geom(N=n, C=o, Hs=hs, Ws=ws, C2=0, Cpad=wp.shape[2], KH=2, KW=2, SS=1, TS=-1,
                             OFFY=(qy + pad - ky0) // stride, OFFX=(qx + pad - kx0) // stride, Hq=hs, Wq=ws, OS=stride, QY=qy, QX=qx,
                             Ho=ho, Wo=wo, Cout=c, CoutPad=wp.shape[0], pre_act=int(pre_act), act=act)
            wqs.append(wq)
        if _lib.load().mtrssm_convt_quad_supported(geoms):
            job = ("quad", geoms, y, wqs, bias, out, actgrad_in)
            if _DEFER is not None:
                _DEFER.append(job)
            else:
                _launch_gather(job)
            return out
    if stride == 2 and c <= 8 and kh * kw * o <= 256 and ho % 2 == 0 and wo % 2 == 0 and TGATHER_THIN:  # noqa: PLR2004
        # few output channels (backward-data of the encoders' second conv): all parity classes in one VALU pass
        lib = _lib.load()
        wc = w.contiguous()
        taps_used = kh * kw / (stride * stride)
        _lib.check(_lib.TIMERS.call(
            "mtrssm_conv_tgather_thin", lib.mtrssm_conv_tgather_thin, n, o, hs, ws, c, kh, kw, stride, pad, ho, wo, _lib.ptr(y),
            _lib.ptr(wc), _lib.ptr(bias), int(pre_act), act, _lib.ptr(actgrad_in), _lib.ptr(add_in), _lib.ptr(out),
            _lib.stream_ptr(y.device), flops=2.0 * n * ho * wo * c * taps_used * o,
            nbytes=4.0 * (y.numel() + out.numel() * (1 + (actgrad_in is not None) + (add_in is not None)))), "mtrssm_conv_tgather_thin")
        return out
    for qy in range(min(stride, ho)):
        ky0 = (qy + pad) % stride
        for qx in range(min(stride, wo)):
            kx0 = (qx + pad) % stride
            wsub = w[:, :, ky0::stride, kx0::stride].permute(1, 0, 2, 3)  # [c][o][nty][ntx]
            wpq = pack_weight(wsub)
            wp = wpq[0]
            geom = _geom(N=n, C=o, Hs=hs, Ws=ws, C2=0, Cpad=wp.shape[2], KH=wsub.shape[2], KW=wsub.shape[3], SS=1, TS=-1,
                         OFFY=(qy + pad - ky0) // stride, OFFX=(qx + pad - kx0) // stride,
                         Hq=(ho - qy + stride - 1) // stride, Wq=(wo - qx + stride - 1) // stride, OS=stride, QY=qy, QX=qx,
                         Ho=ho, Wo=wo, Cout=c, CoutPad=wp.shape[0], pre_act=int(pre_act), act=act)
            _gather_gemm(geom, y, None, wpq, bias, actgrad_in, out, add_in)
    return out


# Accumulation targets of the weight-gradient kernels (fp32 atomics) must start at zero.  They are carved out of a zeroed
# chunk -- one fill per chunk instead of one per tensor (the train step has ~90 such targets of a few KB each); a chunk is
# never handed out twice and is freed by the allocator when the last view of it dies.
TGATHER_THIN = os.environ.get("MTRSSM_TGATHER_THIN", "1") != "0"  # A/B switch of conv_tgather_thin_kernel
CONVT_QUAD = os.environ.get("MTRSSM_CONVT_QUAD", "1") != "0"  # A/B switch of convt_quad_resident_kernel
CONVT_QUAD_BWD = os.environ.get("MTRSSM_CONVT_QUAD_BWD", "1") != "0"
CONVT_BAND = os.environ.get("MTRSSM_CONVT_BAND", "1") != "0"  # A/B switch of convt4s2_band_kernel (the decoders' last layer on the MFMA)  # its use for a k=3 s=2 conv's backward-data (see there)
_ZERO_CHUNK_FLOATS = 2 << 20
_ZERO_CHUNKS: dict[tuple, list] = {}


def _zeros(n: int, device: torch.device) -> Tensor:
    need = (n + 63) // 64 * 64  # 256-byte aligned slices
    key = (device, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    st = _ZERO_CHUNKS.get(key)
    if st is None or st[1] + need > st[0].numel():
        st = _ZERO_CHUNKS[key] = [torch.zeros(max(_ZERO_CHUNK_FLOATS, need), device=device, dtype=torch.float32), 0]
    out = st[0][st[1] : st[1] + n]
    st[1] += need
    return out


class _ConvGradSink:
    """Conv weight gradients of parameters that live in an ``optim.FlatParameters`` buffer: each weight owns a persistent
    packed accumulation buffer ``[OPad][taps][IPad]`` (what the weight-gradient kernels' coalesced atomics want), and ONE
    ``mtrssm_unpack_conv_grads`` launch at the end of the backward pass (autograd's ``queue_callback``) adds them all into the
    flat gradient buffer in the parameter layout and clears them.  Replaces, per conv weight, a zero fill + a strided
    AccumulateGrad add (~80 of each per train step); the autograd nodes return None for these weights."""

    def __init__(self) -> None:
        self.entries: dict[tuple, tuple] = {}
        self.table: Tensor | None = None
        self.retired: list[Tensor] = []  # superseded tables stay allocated: a captured graph may still read one (64 B per entry)
        self.pending = False

    def target(self, weight: Tensor | None, o: int, i: int, taps: int, opad: int, ipad: int) -> Tensor | None:
        """The packed accumulation buffer for ``weight`` ([o][i][kh][kw]-shaped parameter), or None when it has no flat home."""
        if weight is None or not weight.is_contiguous():
            return None
        found = grad_target_owner(weight)
        if found is None:
            return None
        dst, owner = found
        # keyed by the DESTINATION (a live flat gradient buffer), never by the weight's address: a later model whose flat
        # parameter buffer reuses a freed one's address must not inherit that model's entries (its gradients would land in
        # the dead model's buffer).  Entries of collected owners are dropped.
        key = (dst.data_ptr(), o, i, taps, opad, ipad)
        e = self.entries.get(key)
        if e is not None and e[2]() is not owner:
            e = None
        if e is None:
            self.entries = {k: v for k, v in self.entries.items() if v[2]() is not None and k != key}
            packed = torch.zeros(opad, taps, ipad, device=weight.device, dtype=torch.float32)
            e = self.entries[key] = (packed, dst, weakref.ref(owner))
            self.table = None
        if not self.pending:
            self.pending = True
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)  # noqa: SLF001
        return e[0]

    def discard(self) -> None:
        """A backward pass raised before autograd ran ``flush`` (its final callbacks are dropped then): the packed buffers hold
        half-finished sums and ``pending`` would stay set for good -- every later backward would skip ``queue_callback`` and the
        conv weights would silently stop receiving gradients.  Called at the next ``begin_step`` / ``zero_grad``: clear the
        buffers and re-arm."""
        if self.pending:
            self.pending = False
            _reduce_deferred()  # (drops the recorded sums of the abandoned backward; their targets are cleared below)
            for packed, _dst, _owner in self.entries.values():
                packed.zero_()

    def flush(self) -> None:
        self.pending = False
        _reduce_deferred()  # the partial-set sums of the step's weight-gradient kernels, one launch, before they are unpacked
        if any(v[2]() is None for v in self.entries.values()):
            self.entries = {k: v for k, v in self.entries.items() if v[2]() is not None}
            self.table = None
        if not self.entries:
            return
        dev = next(iter(self.entries.values()))[0].device
        if self.table is None:
            rows = [[packed.data_ptr(), dst.data_ptr(), k[1], k[2], k[3], k[5], 0, 0] for k, (packed, dst, _) in self.entries.items()]
            self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
            self.retired.append(self.table)
        _lib.check(_lib.TIMERS.call("mtrssm_unpack_conv_grads", _lib.load().mtrssm_unpack_conv_grads, _lib.raw_ptr(self.table),
                                    len(self.entries), 32, _lib.stream_ptr(dev)), "mtrssm_unpack_conv_grads")


_GRAD_SINK = _ConvGradSink()


def discard_pending_grads() -> None:
    """Start of a step (``begin_step``, ``FlatParameters.zero_grad``): drop what a backward that raised left in the sink."""
    _GRAD_SINK.discard()


def flush_pending_grads() -> None:
    """``FlatAdamW.step`` / ``FlatDataParallel.reduce``: gradients still parked in the sink (autograd never ran the flush
    callback because the backward raised, and the caller went on) are added to the flat buffer before it is consumed."""
    if _GRAD_SINK.pending:
        _GRAD_SINK.flush()


def reset_scratch(*, pin: bool = False) -> None:
    """Forget the partly used zeroed chunks (the next weight gradient opens a fresh one).  ``graph.CapturedTrainStep`` calls
    this right before and right after a capture: a chunk zeroed BEFORE the capture would come back dirty on the second
    replay, and one allocated inside the capture belongs to the graph's private pool.  ``pin`` keeps every packed-weight
    buffer of the step plan alive for good (the captured kernels read those addresses)."""
    _ZERO_CHUNKS.clear()
    if pin:
        _PLAN.pinned += 1


def unpin_scratch() -> None:
    """The captured graph that asked for ``reset_scratch(pin=True)`` is gone: the step plan may drop entries again."""
    _PLAN.pinned = max(0, _PLAN.pinned - 1)


# Workspace of the staged weight-gradient kernels (their partial tile sets, include/mtrssm.h: mtrssm_conv_weight_grad): ONE
# buffer per (device, stream), shared by every weight-gradient launch of that stream (they run in order), sized by the
# library's own query, grown on demand.  Outgrown buffers are kept: a captured hipGraph may still hold their address.
_WGRAD_WS: dict[tuple, Tensor] = {}
_WGRAD_WS_RETIRED: list[Tensor] = []
_WGRAD_NEED: dict[tuple, int] = {}


# With the partial-set sums deferred to the end of the backward pass (``DEFER_WGRAD_REDUCE``; include/mtrssm.h:
# mtrssm_conv_weight_grad_deferred) every weight-gradient launch of the step needs bytes of its own until then: slot k of the
# (device, stream) serves the k-th deferred launch since the last reduce.
DEFER_WGRAD_REDUCE = os.environ.get("MTRSSM_DEFER_WGRAD_REDUCE", "1") != "0"
_WGRAD_SLOTS: dict[tuple, list[Tensor]] = {}
_WGRAD_SLOT_NEXT: dict[tuple, int] = {}


def _reduce_deferred() -> None:
    for (device, stream), used in list(_WGRAD_SLOT_NEXT.items()):
        if used:
            _WGRAD_SLOT_NEXT[(device, stream)] = 0
            _lib.check(_lib.TIMERS.call("mtrssm_conv_weight_grad_reduce", _lib.load().mtrssm_conv_weight_grad_reduce, C.c_void_p(stream)),
                       "mtrssm_conv_weight_grad_reduce")


def _wgrad_workspace(geom: C.Structure, pre_act_a: int, device: torch.device, *, own: bool = False) -> Tensor | None:
    gkey = (tuple(getattr(geom, n) for n, _ in geom._fields_), pre_act_a)
    need = _WGRAD_NEED.get(gkey)
    if need is None:
        need = _WGRAD_NEED[gkey] = max(0, int(_lib.load().mtrssm_conv_weight_grad_workspace_bytes(C.byref(geom), pre_act_a)))
    if need == 0:
        return None
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    if own:
        slots = _WGRAD_SLOTS.setdefault(key, [])
        k = _WGRAD_SLOT_NEXT.get(key, 0)
        _WGRAD_SLOT_NEXT[key] = k + 1
        if k == len(slots):
            slots.append(torch.empty((need + 3) // 4 + 64, device=device, dtype=torch.float32))
        elif slots[k].numel() * 4 < need:
            _WGRAD_WS_RETIRED.append(slots[k])
            slots[k] = torch.empty((need + 3) // 4 + 64, device=device, dtype=torch.float32)
        return slots[k]
    ws = _WGRAD_WS.get(key)
    if ws is None or ws.numel() * 4 < need:
        if ws is not None:
            _WGRAD_WS_RETIRED.append(ws)
        ws = _WGRAD_WS[key] = torch.empty((need + 3) // 4 + 64, device=device, dtype=torch.float32)
    return ws


def _weight_grad(a: Tensor, src: Tensor, coords: Tensor | None, kh: int, kw: int, stride: int, pad: int, pre_act_a: bool,  # noqa: PLR0913
                 pre_act_src: bool, act: int, *, want_bias: bool = False, weight: Tensor | None = None,
                 bias: Tensor | None = None) -> tuple[Tensor | None, Tensor | None]:  # noqa: FBT001
    """``dw[o][i][ky][kx] = sum preA(a)[n,o,y,x] * pre(src ++ coords)[n,i,y*s-p+ky,x*s-p+kx]`` -> ``[O][I][kh][kw]``.

    With ``want_bias`` (only when ``a`` is the raw output gradient) the same pass also returns ``sum_{n,y,x} a``.
    ``weight`` / ``bias``: the parameters themselves; when they live in a flat gradient buffer the gradients are accumulated
    there (``_ConvGradSink``; the bias directly by the kernel's atomics) and None is returned in their place."""
    lib = _lib.load()
    n, o, hq, wq = a.shape
    _, c, hs, ws = src.shape
    c2 = 0 if coords is None else coords.shape[0]
    opad, ipad = _pads(o, c + c2)
    sunk = _GRAD_SINK.target(weight, o, c + c2, kh * kw, opad, ipad)
    dwp = sunk if sunk is not None else _zeros(opad * kh * kw * ipad, a.device).view(opad, kh * kw, ipad)
    dbias = bias_sunk = None
    if want_bias:
        bias_sunk = grad_target(bias) if bias is not None and bias.is_contiguous() else None
        dbias = bias_sunk if bias_sunk is not None else _zeros(o, a.device)
    geom = _geom(N=n, C=c, Hs=hs, Ws=ws, C2=c2, Cpad=ipad, KH=kh, KW=kw, SS=stride, TS=1, OFFY=-pad, OFFX=-pad, Hq=hq, Wq=wq,
                 OS=1, QY=0, QX=0, Ho=hq, Wo=wq, Cout=o, CoutPad=opad, pre_act=int(pre_act_src), act=act)
    flops = 2.0 * n * hq * wq * o * kh * kw * (c + c2)
    nbytes = 4.0 * (a.numel() + src.numel())
    # the sums of the partial sets wait for the end of the backward pass when both results go to the flat buffer (the sink's
    # flush runs them in one launch before it unpacks); a gradient handed back to autograd as a tensor is summed on the spot
    defer = DEFER_WGRAD_REDUCE and sunk is not None and (not want_bias or bias_sunk is not None)
    ws = _wgrad_workspace(geom, int(pre_act_a), a.device, own=defer)
    entry = lib.mtrssm_conv_weight_grad_deferred if defer and ws is not None else lib.mtrssm_conv_weight_grad
    _lib.check(_lib.TIMERS.call(
        "mtrssm_conv_weight_grad", entry, C.byref(geom), _lib.ptr(a), _lib.ptr(src), _lib.ptr(coords),
        int(pre_act_a), _lib.ptr(dwp), _lib.ptr(dbias), _lib.raw_ptr(ws), 0 if ws is None else ws.numel() * 4, _lib.stream_ptr(a.device),
        flops=flops, nbytes=nbytes), "mtrssm_conv_weight_grad")
    g_w = None if sunk is not None else dwp[:o, :, : c + c2].reshape(o, kh, kw, c + c2).permute(0, 3, 1, 2)
    return g_w, (None if bias_sunk is not None else dbias)


def _channel_sum(x: Tensor, bias: Tensor | None = None) -> Tensor | None:
    """``sum_{n,y,x} x[n, c, y, x]``; straight into ``bias``'s flat gradient view when it has one (None is returned then)."""
    lib = _lib.load()
    n, c, h, w = x.shape
    sunk = grad_target(bias) if bias is not None and bias.is_contiguous() else None
    out = sunk if sunk is not None else _zeros(c, x.device)
    _lib.check(_lib.TIMERS.call("mtrssm_channel_sum", lib.mtrssm_channel_sum, _lib.ptr(x), n, c, h * w, _lib.ptr(out),
                                _lib.stream_ptr(x.device)), "mtrssm_channel_sum")
    return None if sunk is not None else out


class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, coords, stride, pad, pre_act, act):  # noqa: ANN001, PLR0913
        x, weight = x.contiguous(), weight.contiguous()
        bias = None if bias is None else bias.contiguous()
        coords = None if coords is None else coords.contiguous()
        out = _conv_forward_gather(x, coords, weight, bias, stride, pad, pre_act, act)
        ctx.save_for_backward(x, weight, coords if coords is not None else x.new_zeros(0))
        ctx.cfg = (stride, pad, pre_act, act, coords is not None, bias is not None)
        ctx.bias = bias  # the parameter (for its gradient's destination), not needed as a value
        return out

    @staticmethod
    def backward(ctx, g_out):  # noqa: ANN001, ANN205
        x, weight, coords = ctx.saved_tensors
        stride, pad, pre_act, act, has_coords, has_bias = ctx.cfg
        coords = coords if has_coords else None
        g_out = g_out.contiguous()
        kh, kw = weight.shape[2:]
        c = x.shape[1]
        g_x = None
        if ctx.needs_input_grad[0]:
            g_x = _conv_transposed_gather(g_out, weight[:, :c], None, stride, pad, (x.shape[2], x.shape[3]), False, act,
                                          actgrad_in=x if pre_act else None)
        g_w = g_b = None
        want_b = has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            g_w, g_b = _weight_grad(g_out, x, coords, kh, kw, stride, pad, False, pre_act, act, want_bias=want_b, weight=weight,
                                    bias=ctx.bias)
        elif want_b:
            g_b = _channel_sum(g_out, ctx.bias)
        return g_x, g_w, g_b, None, None, None, None, None


class _ConvTranspose2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, out_pad, pre_act, act):  # noqa: ANN001, PLR0913
        x, weight = x.contiguous(), weight.contiguous()
        bias = None if bias is None else bias.contiguous()
        kh, kw = weight.shape[2:]
        ho = (x.shape[2] - 1) * stride - 2 * pad + kh + out_pad
        wo = (x.shape[3] - 1) * stride - 2 * pad + kw + out_pad
        out = _conv_transposed_gather(x, weight, bias, stride, pad, (ho, wo), pre_act, act)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, pad, pre_act, act, bias is not None)
        ctx.bias = bias
        return out

    @staticmethod
    def backward(ctx, g_out):  # noqa: ANN001, ANN205
        x, weight = ctx.saved_tensors
        stride, pad, pre_act, act, has_bias = ctx.cfg
        g_out = g_out.contiguous()
        kh, kw = weight.shape[2:]
        g_x = None
        if ctx.needs_input_grad[0]:
            # dX[n,ci,y,x] = sum_{co,ky,kx} w[ci][co][ky][kx] dOut[n,co,y*s-p+ky,x*s-p+kx]: a plain conv gather of dOut
            g_x = _conv_forward_gather(g_out, None, weight, None, stride, pad, False, act, actgrad_in=x if pre_act else None)
        g_w = None
        if ctx.needs_input_grad[1]:
            # dW[ci][co][ky][kx] = sum pre(x)[n,ci,y,x] dOut[n,co,y*s-p+ky,x*s-p+kx]
            g_w, _ = _weight_grad(x, g_out, None, kh, kw, stride, pad, pre_act, False, act, weight=weight)
        g_b = _channel_sum(g_out, ctx.bias) if has_bias and ctx.needs_input_grad[2] else None
        return g_x, g_w, g_b, None, None, None, None, None


RESBLOCK_FUSE = True  # the blocks' forward as ONE launch where the library has a fused kernel (mtrssm_residual_block_fwd)


def _residual_job(x: Tensor, w3: Tensor, b3: Tensor, w1: Tensor, b1: Tensor, act: int, sub: int = 0) -> tuple | None:  # noqa: PLR0913
    """Arguments of ``mtrssm_residual_block_fwd`` for one block (with fresh ``h`` and ``y``), or None when it has no fused kernel.
    ``sub``: the pack-buffer slot (``pack_weight``) -- the second block of a paired launch must not re-use the first's."""
    if not RESBLOCK_FUSE or w3.shape[2:] != (3, 3) or w1.shape[2:] != (1, 1) or w1.shape[:2] != (x.shape[1], w3.shape[0]):
        return None
    n, c, hs, ws = x.shape
    o = w3.shape[0]
    wp, wq = pack_weight(w3, sub)
    if wq is None:
        return None
    geom = _geom(N=n, C=c, Hs=hs, Ws=ws, C2=0, Cpad=wp.shape[2], KH=3, KW=3, SS=1, TS=1, OFFY=-1, OFFX=-1, Hq=hs, Wq=ws, OS=1, QY=0,
                 QX=0, Ho=hs, Wo=ws, Cout=o, CoutPad=wp.shape[0], pre_act=1, act=act)
    if not _lib.load().mtrssm_residual_block_fwd_supported(C.byref(geom)):
        return None
    h = torch.empty(n, o, hs, ws, device=x.device, dtype=torch.float32)
    return (geom, x, wq, b3, w1, b1, h, torch.empty_like(x))


def _residual_launch(ja: tuple, jb: tuple | None) -> None:
    def work(job: tuple) -> tuple[float, float]:
        g = job[0]
        pixels = g.N * g.Hq * g.Wq
        return 2.0 * pixels * g.Cout * (9 * g.C + g.C), 4.0 * pixels * (2 * g.C + g.Cout)

    def args(job: tuple | None) -> tuple:
        if job is None:
            return (None,) * 8
        geom, x, wq, b3, w1, b1, h, y = job
        return (C.byref(geom), _lib.ptr(x), _lib.raw_ptr(wq), _lib.ptr(b3), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(h), _lib.ptr(y))

    fa, ba = work(ja)
    fb, bb = work(jb) if jb is not None else (0.0, 0.0)
    lib = _lib.load()
    _lib.check(_lib.TIMERS.call("mtrssm_residual_block_fwd", lib.mtrssm_residual_block_fwd, *args(ja), *args(jb),
                                _lib.stream_ptr(ja[1].device), flops=fa + fb, nbytes=ba + bb), "mtrssm_residual_block_fwd")


class _ResidualBlock(torch.autograd.Function):
    """``y = x + Conv1x1(act(Conv3x3(act(x))))`` as ONE node: the skip add rides in the second conv's epilogue and the
    skip's gradient in the epilogue of the first conv's backward-data (``oracle/ref_cnn.py:ResidualBlock``)."""

    @staticmethod
    def forward(ctx, x, w3, b3, w1, b1, act):  # noqa: ANN001, PLR0913
        x, w3, b3, w1, b1 = (t.contiguous() for t in (x, w3, b3, w1, b1))
        p3, p1 = w3.shape[2] // 2, w1.shape[2] // 2
        job = _residual_job(x, w3, b3, w1, b1, act)
        if job is not None:
            _residual_launch(job, None)
            h, y = job[6], job[7]
        else:
            h = _conv_forward_gather(x, None, w3, b3, 1, p3, True, act)
            y = _conv_forward_gather(h, None, w1, b1, 1, p1, True, act, add_in=x)
        ctx.save_for_backward(x, h, w3, w1)
        ctx.act, ctx.biases = act, (b3, b1)
        return y

    @staticmethod
    def backward(ctx, g_y):  # noqa: ANN001, ANN205
        x, h, w3, w1 = ctx.saved_tensors
        act = ctx.act
        g_y = g_y.contiguous()
        p3, p1 = w3.shape[2] // 2, w1.shape[2] // 2
        g_h = _conv_transposed_gather(g_y, w1, None, 1, p1, (h.shape[2], h.shape[3]), False, act, actgrad_in=h)
        b3, b1 = ctx.biases
        g_w1, g_b1 = _weight_grad(g_y, h, None, w1.shape[2], w1.shape[3], 1, p1, False, True, act, want_bias=True, weight=w1, bias=b1)
        g_x = None
        if ctx.needs_input_grad[0]:
            g_x = _conv_transposed_gather(g_h, w3, None, 1, p3, (x.shape[2], x.shape[3]), False, act, actgrad_in=x, add_in=g_y)
        g_w3, g_b3 = _weight_grad(g_h, x, None, w3.shape[2], w3.shape[3], 1, p3, False, True, act, want_bias=True, weight=w3, bias=b3)
        return g_x, g_w3, g_b3, g_w1, g_b1, None


class _PairResidualBlock(torch.autograd.Function):
    """``_ResidualBlock`` of the audio and of the vision stack as ONE node: every gather of the pair (two forward, two
    backward-data) is one launch for both branches; the weight gradients stay one launch each (persistent kernels)."""

    @staticmethod
    def forward(ctx, xa, w3a, b3a, w1a, b1a, xv, w3v, b3v, w1v, b1v, act):  # noqa: ANN001, PLR0913
        xa, w3a, b3a, w1a, b1a, xv, w3v, b3v, w1v, b1v = (t.contiguous() for t in (xa, w3a, b3a, w1a, b1a, xv, w3v, b3v, w1v, b1v))
        p3, p1 = w3a.shape[2] // 2, w1a.shape[2] // 2
        ja = _residual_job(xa, w3a, b3a, w1a, b1a, act)
        jv = _residual_job(xv, w3v, b3v, w1v, b1v, act, sub=1) if ja is not None else None
        if ja is not None and jv is not None:
            _residual_launch(ja, jv)
            ha, ya, hv, yv = ja[6], ja[7], jv[6], jv[7]
        else:
            ha, hv = paired(lambda: _conv_forward_gather(xa, None, w3a, b3a, 1, p3, True, act),
                            lambda: _conv_forward_gather(xv, None, w3v, b3v, 1, p3, True, act))
            ya, yv = paired(lambda: _conv_forward_gather(ha, None, w1a, b1a, 1, p1, True, act, add_in=xa),
                            lambda: _conv_forward_gather(hv, None, w1v, b1v, 1, p1, True, act, add_in=xv))
        ctx.save_for_backward(xa, ha, w3a, w1a, xv, hv, w3v, w1v)
        ctx.act, ctx.biases = act, (b3a, b1a, b3v, b1v)
        return ya, yv

    @staticmethod
    def backward(ctx, g_ya, g_yv):  # noqa: ANN001, ANN205
        xa, ha, w3a, w1a, xv, hv, w3v, w1v = ctx.saved_tensors
        act = ctx.act
        g_ya, g_yv = g_ya.contiguous(), g_yv.contiguous()
        p3, p1 = w3a.shape[2] // 2, w1a.shape[2] // 2
        g_ha, g_hv = paired(lambda: _conv_transposed_gather(g_ya, w1a, None, 1, p1, (ha.shape[2], ha.shape[3]), False, act, actgrad_in=ha),
                            lambda: _conv_transposed_gather(g_yv, w1v, None, 1, p1, (hv.shape[2], hv.shape[3]), False, act, actgrad_in=hv))
        b3a, b1a, b3v, b1v = ctx.biases
        g_w1a, g_b1a = _weight_grad(g_ya, ha, None, w1a.shape[2], w1a.shape[3], 1, p1, False, True, act, want_bias=True, weight=w1a, bias=b1a)
        g_w1v, g_b1v = _weight_grad(g_yv, hv, None, w1v.shape[2], w1v.shape[3], 1, p1, False, True, act, want_bias=True, weight=w1v, bias=b1v)
        g_xa, g_xv = paired(
            lambda: _conv_transposed_gather(g_ha, w3a, None, 1, p3, (xa.shape[2], xa.shape[3]), False, act, actgrad_in=xa, add_in=g_ya),
            lambda: _conv_transposed_gather(g_hv, w3v, None, 1, p3, (xv.shape[2], xv.shape[3]), False, act, actgrad_in=xv, add_in=g_yv))
        g_w3a, g_b3a = _weight_grad(g_ha, xa, None, w3a.shape[2], w3a.shape[3], 1, p3, False, True, act, want_bias=True, weight=w3a, bias=b3a)
        g_w3v, g_b3v = _weight_grad(g_hv, xv, None, w3v.shape[2], w3v.shape[3], 1, p3, False, True, act, want_bias=True, weight=w3v, bias=b3v)
        return g_xa, g_w3a, g_b3a, g_w1a, g_b1a, g_xv, g_w3v, g_b3v, g_w1v, g_b1v, None


def residual_block_pair(xa: Tensor, pa: tuple[Tensor, Tensor, Tensor, Tensor], xv: Tensor, pv: tuple[Tensor, Tensor, Tensor, Tensor],
                        *, act: int) -> tuple[Tensor, Tensor]:
    """Two residual blocks with the same channel counts (``p* = (w3, b3, w1, b1)``) on different planes."""
    return _PairResidualBlock.apply(xa, *pa, xv, *pv, int(act))


def residual_block(x: Tensor, w3: Tensor, b3: Tensor, w1: Tensor, b1: Tensor, *, act: int) -> Tensor:
    return _ResidualBlock.apply(x, w3, b3, w1, b1, int(act))


class _BranchCtx:
    """Stand-in for the autograd ctx of one branch inside a paired node."""

    def __init__(self, needs: tuple = ()) -> None:
        self.needs_input_grad = needs
        self.saved_tensors: tuple = ()
        self.cfg: tuple = ()

    def save_for_backward(self, *tensors: Tensor) -> None:
        self.saved_tensors = tensors


def _pair_function(base: type) -> type:
    """``base`` (``_Conv2d`` / ``_ConvTranspose2d``) for two branches as ONE node whose gathers go out as paired launches.
    Valid because the only launches ``base`` issues besides gathers (weight gradient, bias sum) never read a gather's
    output of the same call -- a deferred gather may therefore run after them."""

    class _Pair(torch.autograd.Function):
        @staticmethod
        def forward(ctx, n, *args):  # noqa: ANN001, ANN002, ANN205
            ca, cb = _BranchCtx(), _BranchCtx()
            ya, yb = paired(lambda: base.forward(ca, *args[:n]), lambda: base.forward(cb, *args[n:]))
            ctx.n, ctx.na, ctx.cfgs = n, len(ca.saved_tensors), (ca.cfg, cb.cfg)
            ctx.biases = (getattr(ca, "bias", None), getattr(cb, "bias", None))
            ctx.save_for_backward(*ca.saved_tensors, *cb.saved_tensors)
            return ya, yb

        @staticmethod
        def backward(ctx, ga, gb):  # noqa: ANN001, ANN205
            saved, needs = ctx.saved_tensors, ctx.needs_input_grad[1:]
            ca, cb = _BranchCtx(needs[: ctx.n]), _BranchCtx(needs[ctx.n :])
            ca.saved_tensors, cb.saved_tensors = saved[: ctx.na], saved[ctx.na :]
            ca.cfg, cb.cfg = ctx.cfgs
            ca.bias, cb.bias = ctx.biases
            ra, rb = paired(lambda: base.backward(ca, ga), lambda: base.backward(cb, gb))
            return (None, *ra, *rb)

    _Pair.__name__ = f"_Pair{base.__name__}"
    return _Pair


_PairConv2d = _pair_function(_Conv2d)
_PairConvTranspose2d = _pair_function(_ConvTranspose2d)


def conv2d_pair(a: tuple, b: tuple) -> tuple[Tensor, Tensor]:
    """Two ``conv2d`` calls (``(x, weight, bias, stride, padding, pre_act, act, coords)`` each) with shared launches."""
    fa = (a[0], a[1], a[2], a[7], int(a[3]), int(a[4]), bool(a[5]), int(a[6]))
    fb = (b[0], b[1], b[2], b[7], int(b[3]), int(b[4]), bool(b[5]), int(b[6]))
    return _PairConv2d.apply(len(fa), *fa, *fb)


def conv_transpose2d_pair(a: tuple, b: tuple) -> tuple[Tensor, Tensor]:
    """Two ``conv_transpose2d`` calls (``(x, weight, bias, stride, padding, output_padding, pre_act, act)`` each)."""
    fa = (a[0], a[1], a[2], int(a[3]), int(a[4]), int(a[5]), bool(a[6]), int(a[7]))
    fb = (b[0], b[1], b[2], int(b[3]), int(b[4]), int(b[5]), bool(b[6]), int(b[7]))
    return _PairConvTranspose2d.apply(len(fa), *fa, *fb)


def conv2d(x: Tensor, weight: Tensor, bias: Tensor | None, *, stride: int, padding: int, pre_act: bool, act: int,  # noqa: PLR0913
           coords: Tensor | None = None) -> Tensor:
    return _Conv2d.apply(x, weight, bias, coords, int(stride), int(padding), bool(pre_act), int(act))


def conv_transpose2d(x: Tensor, weight: Tensor, bias: Tensor | None, *, stride: int, padding: int, output_padding: int,  # noqa: PLR0913
                     pre_act: bool, act: int) -> Tensor:
    return _ConvTranspose2d.apply(x, weight, bias, int(stride), int(padding), int(output_padding), bool(pre_act), int(act))
