"""Flat parameter / gradient buffers and the fused HIP AdamW step.

``FlatParameters`` re-homes every (unique) parameter of a module as a view into ONE contiguous fp32
buffer and every ``.grad`` as a view into ONE gradient buffer (with a few spare slots at the tail for
loss scalars), so that

* data-parallel training needs exactly one RCCL all-reduce per step (``parallel.FlatDataParallel``);
* gradient clipping is one reduction and AdamW one elementwise launch over the flat buffer
  (``FlatAdamW`` -> ``mtrssm_sumsq`` + ``mtrssm_adamw_step``), instead of one launch group per tensor.

Semantics follow the reference's trainer config: ``torch.optim.AdamW(lr=1e-3)`` (betas 0.9/0.999, eps
1e-8, weight_decay 1e-2) and ``gradient_clip_val: 10`` by global norm
(``mrssm/mopoe_mrssm/configs/default.yaml:103-107,119``).  One difference: parameters that never receive
a gradient (MMTRSSM's dead ``l_posterior`` / dummy ``transition``, SURVEY.md section 2 "Hazard") see a zero
gradient here rather than being skipped, i.e. they still decay.
"""

from __future__ import annotations

import torch
from torch import Tensor, nn

from multimodal_mtrssm_amd import _lib, conv


class FlatParameters:
    """One flat fp32 parameter buffer + one flat gradient buffer (+ ``extra`` tail slots)."""

    def __init__(self, module: nn.Module, extra: int = 8) -> None:
        params: list[nn.Parameter] = []
        seen: set[int] = set()
        for p in module.parameters():
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                params.append(p)
        if not params:
            msg = "module has no trainable parameters"
            raise ValueError(msg)
        device = params[0].device
        # 4-float (16-byte) alignment of every view keeps the vectorised kernels on aligned addresses
        offsets, total = [], 0
        for p in params:
            offsets.append(total)
            total += (p.numel() + 3) // 4 * 4
        self.numel = total
        self.extra = extra
        self.param = torch.zeros(total, device=device, dtype=torch.float32)
        self.grad_full = torch.zeros(total + extra, device=device, dtype=torch.float32)
        self.grad = self.grad_full[:total]
        self.tail = self.grad_full[total:]
        self.params = params
        with torch.no_grad():
            for p, off in zip(params, offsets, strict=True):
                view = self.param[off : off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[off : off + p.numel()].view_as(p)

    def zero_grad(self) -> None:
        self.grad_full.zero_()

    def check_views(self) -> None:
        """Raise if something replaced a ``.grad`` (e.g. ``zero_grad(set_to_none=True)``)."""
        base = self.grad_full.untyped_storage().data_ptr()
        for p in self.params:
            if p.grad is None or p.grad.untyped_storage().data_ptr() != base:
                msg = "a parameter's .grad no longer aliases the flat gradient buffer; use FlatParameters.zero_grad()"
                raise RuntimeError(msg)


class FlatAdamW:
    """AdamW + global-norm clipping over a ``FlatParameters`` buffer: two HIP launches per step."""

    def __init__(self, flat: FlatParameters, lr: float = 1e-3, betas: tuple[float, float] = (0.9, 0.999),  # noqa: PLR0913
                 eps: float = 1e-8, weight_decay: float = 1e-2, clip_norm: float = 10.0) -> None:
        self.flat = flat
        self.lr, self.betas, self.eps, self.weight_decay, self.clip_norm = lr, betas, eps, weight_decay, clip_norm
        self.exp_avg = torch.zeros_like(flat.param)
        self.exp_avg_sq = torch.zeros_like(flat.param)
        self.sumsq = torch.zeros(1, device=flat.param.device, dtype=torch.float32)
        self.steps = 0
        self.param_groups = [{"lr": lr}]  # lr schedulers (ReduceLROnPlateau, yaml 109-114) mutate this

    def zero_grad(self, set_to_none: bool = False) -> None:  # noqa: FBT001, FBT002
        del set_to_none  # the views must survive
        self.flat.zero_grad()

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0) -> Tensor:
        """Clip by global norm (after ``grad_scale``), then AdamW.  Returns the device scalar sum(g^2)."""
        lib = _lib.load()
        f = self.flat
        stream = _lib.stream_ptr(f.param.device)
        self.steps += 1
        lr = float(self.param_groups[0]["lr"])
        _lib.check(lib.mtrssm_sumsq(_lib.ptr(f.grad), f.numel, _lib.ptr(self.sumsq), stream), "mtrssm_sumsq")
        _lib.check(lib.mtrssm_adamw_step(
            _lib.ptr(f.param), _lib.ptr(f.grad), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq), f.numel,
            _lib.ptr(self.sumsq), float(self.clip_norm), float(grad_scale), lr, self.betas[0], self.betas[1], self.eps,
            self.weight_decay, self.steps, stream), "mtrssm_adamw_step")
        conv.invalidate_packs()  # the parameters changed behind autograd's version counters
        return self.sumsq

    def state_dict(self) -> dict[str, object]:
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "steps": self.steps, "lr": self.param_groups[0]["lr"]}

    def load_state_dict(self, state: dict[str, object]) -> None:
        self.exp_avg.copy_(state["exp_avg"])  # type: ignore[arg-type]
        self.exp_avg_sq.copy_(state["exp_avg_sq"])  # type: ignore[arg-type]
        self.steps = int(state["steps"])  # type: ignore[arg-type]
        self.param_groups[0]["lr"] = float(state["lr"])  # type: ignore[arg-type]


class ReduceLROnPlateau:
    """``torch.optim.lr_scheduler.ReduceLROnPlateau`` semantics for ``FlatAdamW`` (the reference wraps it as
    ``lightning.pytorch.cli.ReduceLROnPlateau``: monitor ``val/loss``, mode ``min``, factor 0.5, patience 50,
    ``default.yaml:109-114``): relative threshold 1e-4, no cooldown.  ``step(metric)`` once per validation epoch."""

    def __init__(self, optimizer: FlatAdamW, mode: str = "min", factor: float = 0.1, patience: int = 10,  # noqa: PLR0913
                 threshold: float = 1e-4, min_lr: float = 0.0, eps: float = 1e-8) -> None:
        if mode not in {"min", "max"}:
            msg = f"mode {mode} is unknown!"
            raise ValueError(msg)
        if factor >= 1.0:
            msg = "Factor should be < 1.0."
            raise ValueError(msg)
        self.optimizer, self.mode, self.factor, self.patience = optimizer, mode, factor, patience
        self.threshold, self.min_lr, self.eps = threshold, min_lr, eps
        self.best = float("inf") if mode == "min" else -float("inf")
        self.num_bad_epochs = 0

    def _better(self, a: float) -> bool:
        if self.mode == "min":
            return a < self.best * (1.0 - self.threshold)
        return a > self.best * (1.0 + self.threshold)

    def step(self, metric: float | Tensor) -> None:
        current = float(metric)
        if self._better(current):
            self.best = current
            self.num_bad_epochs = 0
        else:
            self.num_bad_epochs += 1
        if self.num_bad_epochs > self.patience:
            for group in self.optimizer.param_groups:
                old = float(group["lr"])
                new = max(old * self.factor, self.min_lr)
                if old - new > self.eps:
                    group["lr"] = new
            self.num_bad_epochs = 0


def load_reference_checkpoint(module: nn.Module, path: str, *, strict: bool = True) -> dict[str, object]:
    """Loads a checkpoint written by the reference's Lightning run (``{"state_dict": {...}, ...}``, SURVEY section 8b:
    same parameter names, ``representation.*`` aliasing ``audio_representation.*``) or a bare ``state_dict`` file into
    ``module``.  The file is read with ``weights_only=True`` (nothing in it is executed).  Returns the rest of the
    checkpoint (epoch, optimizer states ...) for the caller."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    state = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
    with torch.no_grad():  # parameters may be views of a FlatParameters buffer: copy in place
        missing, unexpected = module.load_state_dict(state, strict=strict)
    rest = {k: v for k, v in ckpt.items() if k != "state_dict"} if isinstance(ckpt, dict) and "state_dict" in ckpt else {}
    rest["missing_keys"], rest["unexpected_keys"] = list(missing), list(unexpected)
    return rest
