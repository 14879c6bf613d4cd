"""The whole train step as ONE captured hipGraph.

One MoPoE-MRSSM train step is a chain of a few hundred small dependent launches (encoders, the scan,
decoders, their backward, the optimizer); enqueued one by one the host needs about as long as the GPU
(DESIGN.md section 4).  ``CapturedTrainStep`` records

    zero_grad -> shared_step -> backward [-> clip + AdamW when there is one rank]

once (``torch.cuda.graph`` = ``hipStreamBeginCapture`` on a side stream; the library's launches go to torch's
current stream, so they are captured like torch's own) and replays it with one ``hipGraphLaunch`` per step.
What makes that legal here:

* every scalar the kernels need lives in device memory (``FlatAdamW.state``: learning rate, step count, bias
  corrections), so nothing is frozen into the graph that changes between steps;
* the sampling uniforms are drawn OUTSIDE the graph into fixed buffers (``GlobalRowNoise.draw(out=...)``) and the
  batch is copied into fixed buffers, which the captured kernels read;
* the conv layer's zeroed accumulation chunks and packed-weight buffers are (re)created inside the capture
  (``conv.reset_scratch``), so each replay starts from the state the capture started from.

With more than one rank the gradient all-reduce (RCCL) and the optimizer run eagerly after the replay: the
exchange stays a plain ``torch.distributed`` call.
"""

from __future__ import annotations

import torch
from torch import Tensor

from multimodal_mtrssm_amd import conv, scan
from multimodal_mtrssm_amd.optim import FlatAdamW, FlatParameters
from multimodal_mtrssm_amd.parallel import FlatDataParallel, GlobalRowNoise


class CapturedTrainStep:
    """``step(batch)`` = one train step of ``model`` on a batch of the captured shape; returns the loss scalars
    (device tensors, averaged over ranks) exactly as the eager sequence would.

    Construction runs ``warmup`` real steps on the capture stream (every lazy allocation, plan entry and workspace must
    exist before the capture) and then puts parameters, Adam moments, the device-side step count and the noise generator
    BACK where they were: a run with the graph is step for step the eager run.  ``close()`` (or garbage collection) releases
    the pin on the conv layer's packed-weight plan."""

    def __init__(self, model: torch.nn.Module, flat: FlatParameters, opt: FlatAdamW, dp: FlatDataParallel,  # noqa: PLR0913
                 batch: tuple[Tensor, ...], noise: GlobalRowNoise, *, warmup: int = 3) -> None:
        self.model, self.flat, self.opt, self.dp, self.noise = model, flat, opt, dp, noise
        self.batch = tuple(x.clone() for x in batch)
        b, t = batch[0].shape[:2]
        self.shapes = model.noise_shapes(b, t)
        dev = batch[0].device
        self.uniforms = {k: torch.empty(s, device=dev, dtype=torch.float32) for k, s in self.shapes.items()}
        self.fused_optimizer = dp.world == 1
        self.keys: list[str] = []
        self.graph: torch.cuda.CUDAGraph | None = None
        self._capture(warmup)

    # the captured region -------------------------------------------------------------------------
    def _body(self) -> list[str]:
        self.opt.zero_grad()
        out = self.model.shared_step(self.batch, self.uniforms)
        out["loss"].backward()
        keys = list(out)
        if self.fused_optimizer:
            self.dp.sync({k: out[k] for k in keys})  # one rank: only the scalars' copies into the gradient buffer's tail
            self.opt.step(grad_scale=self.dp.grad_scale, check=False)
        else:
            self.dp.stage_scalars({k: out[k] for k in keys})
        return keys

    def _tail(self) -> None:
        if not self.fused_optimizer:
            self.dp.reduce()
            self.opt.step(grad_scale=self.dp.grad_scale, check=False)

    def _capture(self, warmup: int) -> None:
        dev = self.batch[0].device
        # the warm-up steps are real steps: snapshot what they change and restore it afterwards
        snap = (self.flat.param.clone(), self.opt.exp_avg.clone(), self.opt.exp_avg_sq.clone(), self.opt.state.clone(), self.opt.steps,
                self.noise.gen.get_state())
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):  # eager steps on the capture stream: every lazy allocation / plan entry exists
                self.noise.draw(self.shapes, out=self.uniforms)
                self.opt.sync_lr()
                self._body()
                self._tail()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        with torch.no_grad():
            self.flat.param.copy_(snap[0])
            self.opt.exp_avg.copy_(snap[1])
            self.opt.exp_avg_sq.copy_(snap[2])
            self.opt.state.copy_(snap[3])
        self.opt.steps = snap[4]
        self.noise.gen.set_state(snap[5])
        conv.invalidate_packs()
        scan.STATUS.check()  # a warm-up step whose cooperative scan gave up must not be captured
        self.flat.check_views()
        self.opt.active_mask()  # built from what the warm-up steps touched; a fixed buffer from here on
        self.opt.sync_lr()  # (no draw here: the recorded step is not executed, and a draw would shift the stream of uniforms)
        conv.reset_scratch(pin=True)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
            self.keys = self._body()
        conv.reset_scratch()
        self._pinned = True
        self.graph = graph  # capture records, it does not execute: the first step() runs it
        if self.fused_optimizer:
            self.opt.steps -= 1  # FlatAdamW.step counted the recorded (not executed) step on the host

    def close(self) -> None:
        """Drop the graph and the pin it holds on the conv layer's packed-weight plan."""
        self.graph = None
        if getattr(self, "_pinned", False):
            self._pinned = False
            conv.unpin_scratch()

    def __del__(self) -> None:
        self.close()

    # one step ---------------------------------------------------------------------------------------
    def step(self, batch: tuple[Tensor, ...] | None = None) -> dict[str, Tensor]:
        scan.STATUS.poll()  # a cooperative scan launch of an earlier replay gave up (the device skipped that update): raise
        if batch is not None and batch[0] is not self.batch[0]:
            for dst, src in zip(self.batch, batch, strict=True):
                dst.copy_(src)
        self.noise.draw(self.shapes, out=self.uniforms)
        self.opt.sync_lr()
        assert self.graph is not None
        self.graph.replay()
        self._tail()
        self.opt.steps += 1 if self.fused_optimizer else 0  # host mirror of the device-side step count
        scan.STATUS.post()
        if self.dp.world == 1:
            return {k: self.flat.tail[i] for i, k in enumerate(self.keys)}
        return {k: self.flat.tail[i] / self.dp.world for i, k in enumerate(self.keys)}
