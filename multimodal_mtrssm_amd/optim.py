"""Flat parameter / gradient buffers and the fused HIP AdamW step.

``FlatParameters`` re-homes every (unique) parameter of a module as a view into ONE contiguous fp32
buffer and every ``.grad`` as a view into ONE gradient buffer (with a few spare slots at the tail for
loss scalars), so that

* data-parallel training needs exactly one RCCL all-reduce per step (``parallel.FlatDataParallel``);
* gradient clipping is one reduction and AdamW one elementwise launch over the flat buffer
  (``FlatAdamW`` -> ``mtrssm_sumsq`` + ``mtrssm_adamw_step``), instead of one launch group per tensor.

Semantics follow the reference's trainer config: ``torch.optim.AdamW(lr=1e-3)`` (betas 0.9/0.999, eps
1e-8, weight_decay 1e-2) and ``gradient_clip_val: 10`` by global norm
(``mrssm/mopoe_mrssm/configs/default.yaml:103-107,119``), including torch's rule that a parameter whose
``.grad`` is None is skipped entirely (no decay, no moments): parameters that never receive a gradient (MMTRSSM's
dead ``l_posterior`` / dummy ``transition``, SURVEY.md section 2 "Hazard") are masked out of the fused step, so a
checkpoint trained here equals torch's on those tensors too.  The learning rate and the step count live in
device memory, so the whole step can sit inside a captured hipGraph (``graph.CapturedTrainStep``).
"""

from __future__ import annotations

import torch
from torch import Tensor, nn

from multimodal_mtrssm_amd import _lib, conv, linear, scan


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
        self.offsets = offsets
        # which parameters have EVER been handed a gradient by autograd (torch.optim skips `.grad is None` parameters; with
        # pre-set gradient views "None" cannot be observed, so a one-shot hook per parameter records the first accumulation)
        self.touched = [False] * len(params)
        self.touched_version = 0
        self._hooks: list = []
        with torch.no_grad():
            for i, (p, off) in enumerate(zip(params, offsets, strict=True)):
                view = self.param[off : off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[off : off + p.numel()].view_as(p)
                self._hooks.append(p.register_post_accumulate_grad_hook(self._touch_hook(i)))
        self._grad_ptrs = [p.grad.data_ptr() for p in params]
        linear.register_sink(self)  # kernels that own a weight gradient accumulate it straight into self.grad

    def _touch_hook(self, i: int):  # noqa: ANN202
        def hook(_p: Tensor) -> None:
            self.mark_touched(i)
        return hook

    def mark_touched(self, i: int) -> None:
        """Parameter ``i`` received a gradient (autograd's accumulation hook, or a kernel that wrote its flat view)."""
        if not self.touched[i]:
            self.touched[i] = True
            self.touched_version += 1

    def prune_hooks(self) -> None:
        """Drop the one-shot hooks of parameters already seen (called between steps, never from inside backward)."""
        for i, h in enumerate(self._hooks):
            if h is not None and self.touched[i]:
                h.remove()
                self._hooks[i] = None

    def index_of(self, p: Tensor) -> int | None:
        """Index of the parameter whose flat view starts at ``p``'s address (None if ``p`` is not one of them)."""
        if not hasattr(self, "_by_ptr"):
            self._by_ptr = {q.data_ptr(): i for i, q in enumerate(self.params)}
        return self._by_ptr.get(p.data_ptr())

    def active_mask(self) -> Tensor | None:
        """One byte per flat element: 1 where the parameter has ever received a gradient; None when all have."""
        if all(self.touched):
            return None
        mask = torch.zeros(self.numel, dtype=torch.uint8)
        for p, off, t in zip(self.params, self.offsets, self.touched, strict=True):
            if t:
                mask[off : off + p.numel()] = 1
        return mask.to(self.param.device)

    def zero_grad(self) -> None:
        """One launch over the flat gradient buffer (+ tail).  On the GPU a plain kernel (``mtrssm_clear``), not a memset:
        inside a captured train step a 16 MB memset node came back with foreign bytes at the buffer's head on replay."""
        g = self.grad_full
        conv.discard_pending_grads()  # packed conv weight gradients of a backward that raised belong to the sums being cleared
        linear.discard_deferred()
        if g.is_cuda:
            _lib.check(_lib.load().mtrssm_clear(_lib.ptr(g), g.numel() * 4, _lib.stream_ptr(g.device)), "mtrssm_clear")
        else:
            g.zero_()

    def check_views(self) -> None:
        """Raise if something replaced a ``.grad`` (e.g. a stock ``model.zero_grad()``, whose set_to_none=True makes autograd
        allocate fresh gradients that the flat optimizer step would never see)."""
        for p, want in zip(self.params, self._grad_ptrs, strict=True):
            if p.grad is None or p.grad.data_ptr() != want:
                msg = ("a parameter's .grad no longer aliases the flat gradient buffer (model.zero_grad() / set_to_none?); "
                       "use FlatParameters.zero_grad() or FlatAdamW.zero_grad()")
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
        # device-resident scalars read by the kernels: [lr, steps taken, 1 - b1^step, sqrt(1 - b2^step)]
        self.state = torch.tensor([lr, 0.0, 0.0, 0.0], dtype=torch.float32).to(flat.param.device)
        self._lr_on_device = float(lr)
        self._mask: Tensor | None = None
        self._mask_version = -1

    def zero_grad(self, set_to_none: bool = False) -> None:  # noqa: FBT001, FBT002
        del set_to_none  # the views must survive
        self.flat.zero_grad()

    def sync_lr(self) -> None:
        """Upload ``param_groups[0]["lr"]`` if a scheduler changed it (an eager fill; call it OUTSIDE a graph capture)."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_on_device:
            self.state[0:1].fill_(lr)
            self._lr_on_device = lr

    def active_mask(self) -> Tensor | None:
        if self._mask_version != self.flat.touched_version:
            self.flat.prune_hooks()
            self._mask = self.flat.active_mask()
            self._mask_version = self.flat.touched_version
        return self._mask

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0, *, check: bool = True) -> Tensor:
        """Clip by global norm (after ``grad_scale``), then AdamW.  Returns the device scalar sum(g^2).

        Two launches (+ one 4-byte memset), no host-side scalar in either: safe inside a hipGraph capture (call
        ``sync_lr()`` before replays when a scheduler may have changed the rate; ``check`` then has to be False).

        The step's cooperative scan kernels (``scan.py``) leave a sticky status word when an exchange between their workgroups
        gave up: both launches are handed that word and skip the update on the device when it is set, and (``check``) the
        host raises ``MtrssmLibraryError`` here as soon as the asynchronous copy posted by an earlier step shows it --
        without synchronising."""
        f = self.flat
        if check:
            scan.STATUS.poll()  # raises if a scan launch of an EARLIER step failed (its update was skipped on the device)
            conv.flush_pending_grads()  # a backward that raised half-way left conv weight gradients in their packed buffers
            linear.flush_deferred()  # ... and weight-gradient GEMMs waiting for the end of the pass
            f.check_views()
            self.sync_lr()
        lib = _lib.load()
        stream = _lib.stream_ptr(f.param.device)
        self.steps += 1
        mask = self.active_mask()
        status = scan.status_word(f.param.device)
        _lib.check(_lib.TIMERS.call("mtrssm_adamw_prepare", lib.mtrssm_adamw_prepare, _lib.ptr(f.grad), f.numel, _lib.ptr(self.sumsq),
                                    _lib.ptr(self.state), _lib.raw_ptr(status), self.betas[0], self.betas[1], stream, nbytes=4.0 * f.numel),
                   "mtrssm_adamw_prepare")
        _lib.check(_lib.TIMERS.call(
            "mtrssm_adamw_apply", lib.mtrssm_adamw_apply, _lib.ptr(f.param), _lib.ptr(f.grad), _lib.ptr(self.exp_avg),
            _lib.ptr(self.exp_avg_sq), _lib.raw_ptr(mask), f.numel, _lib.ptr(self.sumsq), _lib.ptr(self.state), _lib.raw_ptr(status),
            float(self.clip_norm), float(grad_scale), self.betas[0], self.betas[1], self.eps, self.weight_decay, stream,
            nbytes=28.0 * f.numel), "mtrssm_adamw_apply")
        conv.invalidate_packs()  # the parameters changed behind autograd's version counters
        if check:
            scan.STATUS.post()  # asynchronous copy of the status words; read by the next step's poll
        return self.sumsq

    def state_dict(self) -> dict[str, object]:
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "steps": self.steps, "lr": self.param_groups[0]["lr"]}

    def load_state_dict(self, state: dict[str, object]) -> None:
        self.exp_avg.copy_(state["exp_avg"])  # type: ignore[arg-type]
        self.exp_avg_sq.copy_(state["exp_avg_sq"])  # type: ignore[arg-type]
        self.steps = int(state["steps"])  # type: ignore[arg-type]
        self.param_groups[0]["lr"] = float(state["lr"])  # type: ignore[arg-type]
        self.state.copy_(torch.tensor([float(state["lr"]), float(self.steps), 0.0, 0.0]))  # type: ignore[arg-type]
        self._lr_on_device = float(state["lr"])  # type: ignore[arg-type]


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
    conv.invalidate_packs()  # packed conv-weight copies of the old values must not outlive the load
    rest = {k: v for k, v in ckpt.items() if k != "state_dict"} if isinstance(ckpt, dict) and "state_dict" in ckpt else {}
    rest["missing_keys"], rest["unexpected_keys"] = list(missing), list(unexpected)
    return rest
