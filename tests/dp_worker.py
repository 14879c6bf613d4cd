"""CPU worker for tests/test_parallel_gloo.py: two train steps of a small module through ``FlatDataParallel`` with
``GlobalRowNoise``; started by ``bench.launch_workers`` (gloo, one process per rank) or called in-process with world 1.

    python tests/dp_worker.py OUT_DIR GLOBAL_BATCH
"""

from __future__ import annotations

import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


class Net(torch.nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self.used = torch.nn.Linear(6, 4)
        self.head = torch.nn.Linear(4, 1)
        self.dead = torch.nn.Linear(3, 3)  # never called: like MMTRSSM's l_posterior / dummy transition

    def forward(self, x: torch.Tensor, u: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
        # the uniforms enter the loss row by row, as the sampled latents do in the rollout
        return self.head(torch.tanh(self.used(x) + u.mean(dim=(1, 2), keepdim=False).unsqueeze(-1))).squeeze(-1) * v[:, 0]


def run(out_dir: str, global_batch: int) -> None:
    from multimodal_mtrssm_amd.optim import FlatParameters
    from multimodal_mtrssm_amd.parallel import FlatDataParallel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)  # ranks start DIFFERENT on purpose: broadcast must fix it
        net = Net()
        flat = FlatParameters(net, extra=4)
        dp = FlatDataParallel(flat)
        assert dp.world == world and dp.rank == rank
        dp.broadcast_parameters(0)
        g = torch.Generator().manual_seed(5)
        x, y = torch.randn(global_batch, 6, generator=g), torch.randn(global_batch, generator=g)
        xs, ys = dp.shard((x, y))
        source = dp.noise_source(seed=3, device="cpu")
        per = global_batch // world
        keep = []
        for _ in range(2):
            noise = source.draw({"u": (per, 5, 3), "v": (per, 2)})
            keep.append({k: t.clone() for k, t in noise.items()})
            flat.zero_grad()
            loss = (net(xs, noise["u"], noise["v"]) - ys).square().mean()
            loss.backward()
            flat.check_views()
            scalars = dp.sync({"loss": loss})
            with torch.no_grad():
                flat.param.sub_(0.1 * dp.grad_scale * flat.grad)
        torch.save({"param": flat.param.clone(), "grad": flat.grad.clone() * dp.grad_scale, "loss": scalars["loss"].clone(),
                    "noise": keep, "touched": list(flat.touched)}, f"{out_dir}/rank{rank}of{world}.pt")
    finally:
        if world > 1:
            dist.destroy_process_group()


if __name__ == "__main__":
    run(sys.argv[1], int(sys.argv[2]))
