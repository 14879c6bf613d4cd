"""Data-parallel training of the rollout path: shard B, ONE all-reduce per step.

The reference never names a collective; Lightning's DDP would all-reduce gradients bucket by bucket
during backward (SURVEY.md section 2 "Collective (implicit) #1") and the logged losses once per epoch (#2).
Batch rows are independent through the whole T loop, so here every rank runs the full model on its B/N
rows and a single ``all_reduce(SUM)`` over the flat gradient buffer -- with the loss scalars riding in
its tail slots -- is the only exchange (RCCL over xGMI on MI355X: backend "nccl"; gloo in the CPU
tests).  Messages are small (2.5 MB at config-2 core dims) so one un-bucketed call is right; the
1/world scale is folded into the optimizer's ``grad_scale``, not a separate pass.  Parameters that got
no gradient contribute zeros (MMTRSSM's dead parameters).
"""

from __future__ import annotations

import torch
import torch.distributed as dist
from torch import Tensor

from multimodal_mtrssm_amd import conv, linear
from multimodal_mtrssm_amd.optim import FlatParameters


class GlobalRowNoise:
    """Sampling uniforms keyed by (seed, draw number, GLOBAL batch row): SURVEY.md section 8e.

    Every rank holds a generator in the same state and draws the uniforms of the whole global batch
    (``[B_global, ...]`` per key, keys in sorted order -- a few hundred KB), then keeps its own rows.  Row g
    of the global batch therefore sees the same numbers whether the job runs on 1, 2 or 8 ranks, and B=64 on one
    rank equals 2 x 32 on two ranks bit for bit.  ``out`` lets a caller keep the result in fixed buffers (the captured
    train step reads them; the draw itself stays outside the capture)."""

    def __init__(self, seed: int, world: int, rank: int, device: torch.device | str) -> None:
        self.world, self.rank = int(world), int(rank)
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(seed))

    def draw(self, shapes: dict[str, tuple[int, ...]], out: dict[str, Tensor] | None = None) -> dict[str, Tensor]:
        """``shapes``: per key the shape for THIS rank's rows (first extent = local batch); equal on every rank."""
        res: dict[str, Tensor] = {}
        for key in sorted(shapes):
            local = tuple(shapes[key])
            full = torch.rand((local[0] * self.world, *local[1:]), generator=self.gen, device=self.device, dtype=torch.float32)
            mine = full[self.rank * local[0] : (self.rank + 1) * local[0]]
            if out is not None:
                out[key].copy_(mine)
                mine = out[key]
            res[key] = mine
        return res


class FlatDataParallel:
    def __init__(self, flat: FlatParameters, process_group: dist.ProcessGroup | None = None) -> None:
        self.flat = flat
        self.group = process_group
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if self.active else 1
        self.rank = dist.get_rank(process_group) if self.active else 0

    def broadcast_parameters(self, src: int = 0) -> None:
        """Make every rank start from rank ``src``'s weights (one broadcast of the flat buffer)."""
        if self.active:
            dist.broadcast(self.flat.param, src=src, group=self.group)
            conv.invalidate_packs()

    def shard(self, batch: tuple[Tensor, ...]) -> tuple[Tensor, ...]:
        """This rank's contiguous slice of the global batch (rows are independent: no halo)."""
        b = batch[0].shape[0]
        if b % self.world:
            msg = f"global batch {b} is not divisible by world size {self.world}"
            raise ValueError(msg)
        per = b // self.world
        return tuple(x[self.rank * per : (self.rank + 1) * per] for x in batch)

    def noise_source(self, seed: int, device: torch.device | str | None = None) -> GlobalRowNoise:
        """A ``GlobalRowNoise`` for this rank: pass ``source.draw(model.noise_shapes(B_local, T))`` as ``shared_step``'s ``noise``."""
        return GlobalRowNoise(seed, self.world, self.rank, self.flat.param.device if device is None else device)

    @property
    def grad_scale(self) -> float:
        """What the optimizer multiplies the summed gradient by (mean over ranks)."""
        return 1.0 / self.world

    @torch.no_grad()
    def stage_scalars(self, scalars: dict[str, Tensor] | None = None) -> list[str]:
        """Copy the loss scalars into the gradient buffer's tail slots (they ride in the same all-reduce)."""
        keys = list(scalars or {})
        if len(keys) > self.flat.extra:
            msg = f"at most {self.flat.extra} scalars fit in the gradient buffer's tail"
            raise ValueError(msg)
        if keys:  # one stack + one copy instead of a copy per scalar
            self.flat.tail[: len(keys)].copy_(torch.stack([scalars[k].detach().reshape(()) for k in keys]))  # type: ignore[index]
        return keys

    @torch.no_grad()
    def reduce(self) -> None:
        """THE exchange of the step: one all-reduce (SUM, left un-normalised) over gradients + tail."""
        conv.flush_pending_grads()  # (only after a backward that raised: the sink's callback never ran)
        linear.flush_deferred()
        if self.active:  # also with one rank: same code path, the collective is then a no-op copy
            dist.all_reduce(self.flat.grad_full, op=dist.ReduceOp.SUM, group=self.group)

    @torch.no_grad()
    def sync(self, scalars: dict[str, Tensor] | None = None) -> dict[str, Tensor]:
        """All-reduce gradients (SUM, left un-normalised) and average ``scalars`` in the same call."""
        keys = self.stage_scalars(scalars)
        self.reduce()
        if not keys:
            return {}
        mean = self.flat.tail[: len(keys)] / self.world  # one launch for all scalars
        return {k: mean[i] for i, k in enumerate(keys)}
