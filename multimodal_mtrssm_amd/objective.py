"""Gaussian negative log-likelihood with unit scale, as one fused HIP reduction.

Replaces ``models/objective.py:7-23`` (``-Independent(Normal(pred, scale), k).log_prob(target).mean()``).
For ``scale == 1`` the value is ``mean_frames sum_event [0.5 (t - p)^2 + 0.5 log 2pi]``; the kernel reads
prediction and target once (HBM-bound, 16 B / lane) and the backward writes ``(p - t) / frames``.
"""

from __future__ import annotations

import math

import torch
from torch import Tensor

from multimodal_mtrssm_amd import _lib


class _GaussianNLL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prediction: Tensor, target: Tensor, event_ndims: int, act: int = 0) -> Tensor:  # noqa: ANN001
        lib = _lib.load()
        pred, tgt = prediction.contiguous(), target.contiguous()
        event = math.prod(pred.shape[-event_ndims:])
        frames = pred.numel() // event
        out = torch.empty((), device=pred.device, dtype=torch.float32)
        _lib.check(_lib.TIMERS.call("mtrssm_gaussian_nll_fwd", lib.mtrssm_gaussian_nll_fwd, _lib.ptr(pred), _lib.ptr(tgt), frames, event,
                                    int(act), _lib.ptr(out), _lib.stream_ptr(pred.device), nbytes=8.0 * pred.numel()),
                   "mtrssm_gaussian_nll_fwd")
        ctx.save_for_backward(pred, tgt)
        ctx.frames, ctx.event, ctx.act = frames, event, int(act)
        return out

    @staticmethod
    def backward(ctx, g_out: Tensor):  # noqa: ANN001, ANN205
        lib = _lib.load()
        pred, tgt = ctx.saved_tensors
        g_pred = torch.empty_like(pred)
        g = g_out.contiguous()
        _lib.check(_lib.TIMERS.call("mtrssm_gaussian_nll_bwd", lib.mtrssm_gaussian_nll_bwd, _lib.ptr(pred), _lib.ptr(tgt), _lib.ptr(g), ctx.frames,
                                    ctx.event, ctx.act, _lib.ptr(g_pred), _lib.stream_ptr(pred.device), nbytes=12.0 * pred.numel()),
                   "mtrssm_gaussian_nll_bwd")
        return g_pred, None, None, None


def likelihood(prediction: Tensor, target: Tensor, event_ndims: int, scale: float = 1.0, *, out_act: int = 0) -> Tensor:
    """Negative mean log-likelihood of ``target`` under ``Normal(act(prediction), scale)`` (``objective.py:7``).  ``out_act``
    (0 = Identity as in the reference's signature, 3 = Tanh) lets the decoder hand in its raw last-layer output: the
    out_activation is applied while the kernel reads it and its derivative in the backward."""
    if prediction.shape != target.shape:
        msg = f"prediction {tuple(prediction.shape)} and target {tuple(target.shape)} must have the same shape"
        raise ValueError(msg)
    if scale != 1.0:
        # Normal(pred, s): 0.5 ((t-p)/s)^2 + log s + 0.5 log 2pi, by rescaling the unit-scale kernel
        event = math.prod(prediction.shape[-event_ndims:])
        if out_act:
            prediction = torch.tanh(prediction) if out_act == 3 else prediction  # noqa: PLR2004
        unit = _GaussianNLL.apply(prediction / scale, target / scale, event_ndims, 0)
        return unit + event * math.log(scale)
    return _GaussianNLL.apply(prediction, target, event_ndims, int(out_act))
