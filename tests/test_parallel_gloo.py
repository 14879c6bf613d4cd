"""CPU, world_size 2, gloo: the one-all-reduce data-parallel step.

The HIP model cannot run here, so the exchange is exercised with a CPU module: sharding the batch over two
ranks + one flat all-reduce must reproduce the single-process full-batch gradient and the averaged loss
scalars, parameters that never get a gradient must contribute zeros, and both ranks must end identical.
"""

from __future__ import annotations

import os
import socket
import sys
from pathlib import Path

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


class _Net(torch.nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self.used = torch.nn.Linear(6, 4)
        self.head = torch.nn.Linear(4, 1)
        self.dead = torch.nn.Linear(3, 3)  # never called: like MMTRSSM's l_posterior / dummy transition

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.head(torch.tanh(self.used(x))).squeeze(-1)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    sys.path.insert(0, str(ROOT))
    from multimodal_mtrssm_amd.optim import FlatParameters
    from multimodal_mtrssm_amd.parallel import FlatDataParallel

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)  # ranks start DIFFERENT on purpose: broadcast must fix it
        net = _Net()
        flat = FlatParameters(net, extra=4)
        dp = FlatDataParallel(flat)
        dp.broadcast_parameters(0)
        g = torch.Generator().manual_seed(5)
        x, y = torch.randn(8, 6, generator=g), torch.randn(8, generator=g)
        xs, ys = dp.shard((x, y))
        flat.zero_grad()
        loss = (net(xs) - ys).square().mean()
        loss.backward()
        flat.check_views()
        scalars = dp.sync({"loss": loss, "aux": loss * 2})
        torch.save({"grad": flat.grad.clone() * dp.grad_scale, "param": flat.param.clone(), "loss": scalars["loss"].clone(),
                    "aux": scalars["aux"].clone()}, f"{out_dir}/rank{rank}.pt")
    finally:
        dist.destroy_process_group()


def test_two_rank_flat_allreduce_matches_single_process(tmp_path: Path) -> None:
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    assert torch.equal(r0["param"], r1["param"])  # broadcast made the ranks identical
    assert torch.equal(r0["grad"], r1["grad"])  # and the reduced gradient is identical on both
    # single-process reference on the full batch with rank 0's initial weights
    sys.path.insert(0, str(ROOT))
    from multimodal_mtrssm_amd.optim import FlatParameters

    torch.manual_seed(100)
    net = _Net()
    flat = FlatParameters(net, extra=4)
    assert torch.equal(flat.param, r0["param"])
    g = torch.Generator().manual_seed(5)
    x, y = torch.randn(8, 6, generator=g), torch.randn(8, generator=g)
    loss = (net(x) - y).square().mean()
    loss.backward()
    torch.testing.assert_close(r0["grad"], flat.grad, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(r0["loss"], loss.detach(), rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(r0["aux"], 2 * loss.detach(), rtol=1e-6, atol=1e-7)
    dead = torch.cat([p.grad.flatten() for p in net.dead.parameters()])
    assert float(dead.abs().max()) == 0.0  # never-touched parameters contribute zeros


def test_shard_rejects_uneven_batches() -> None:
    sys.path.insert(0, str(ROOT))
    from multimodal_mtrssm_amd.optim import FlatParameters
    from multimodal_mtrssm_amd.parallel import FlatDataParallel

    dp = FlatDataParallel(FlatParameters(_Net()))
    assert dp.world == 1 and dp.grad_scale == 1.0
    (x,) = dp.shard((torch.zeros(5, 6),))
    assert x.shape[0] == 5
    assert dp.sync({"loss": torch.tensor(3.0)})["loss"].item() == 3.0
