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


def test_bench_launcher_runs_two_ranks_and_noise_is_rank_count_invariant(tmp_path: Path) -> None:
    """`bench.py --gpus N` starts its own workers.  The launcher code (`bench.launch_workers`: a child
    `torch.distributed.run` job on 127.0.0.1) drives a CPU worker (tests/dp_worker.py, gloo) through two sharded train steps;
    the same worker run in-process on ONE rank with the full batch must see, row for row, the same uniforms
    (`GlobalRowNoise`: keyed by global batch row) and end with the same parameters."""
    sys.path.insert(0, str(ROOT))
    import bench
    from tests import dp_worker

    rc = bench.launch_workers(2, script=ROOT / "tests" / "dp_worker.py", argv=[str(tmp_path), "8"])
    assert rc == 0
    r0 = torch.load(tmp_path / "rank0of2.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1of2.pt", weights_only=True)
    env = {k: os.environ.pop(k, None) for k in ("WORLD_SIZE", "RANK")}
    try:
        dp_worker.run(str(tmp_path), 8)
    finally:
        os.environ.update({k: v for k, v in env.items() if v is not None})
    one = torch.load(tmp_path / "rank0of1.pt", weights_only=True)
    for step in range(2):
        for key in ("u", "v"):
            both = torch.cat([r0["noise"][step][key], r1["noise"][step][key]])
            assert torch.equal(both, one["noise"][step][key]), (step, key)  # bit for bit
    assert torch.equal(r0["param"], r1["param"])
    torch.testing.assert_close(r0["param"], one["param"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(r0["loss"], one["loss"], rtol=1e-6, atol=1e-7)
    assert r0["touched"] == one["touched"] == [True, True, True, True, False, False]
    # a failing rank fails the launcher
    assert bench.launch_workers(2, script=ROOT / "tests" / "dp_worker.py", argv=[str(tmp_path / "missing"), "8"]) != 0


def test_shard_rejects_uneven_batches() -> None:
    sys.path.insert(0, str(ROOT))
    from multimodal_mtrssm_amd.optim import FlatParameters
    from multimodal_mtrssm_amd.parallel import FlatDataParallel

    dp = FlatDataParallel(FlatParameters(_Net()))
    assert dp.world == 1 and dp.grad_scale == 1.0
    (x,) = dp.shard((torch.zeros(5, 6),))
    assert x.shape[0] == 5
    assert dp.sync({"loss": torch.tensor(3.0)})["loss"].item() == 3.0


def test_bench_main_runs_two_ranks_end_to_end_and_fails_when_a_rank_dies() -> None:
    """the REAL ``bench.py`` entry -- argument parsing, self-launch of ``torch.distributed.run`` on 127.0.0.1,
    process-group warm-up all-reduce, sharded steps through FlatParameters / FlatDataParallel / GlobalRowNoise, barriers, the
    MAX-over-ranks timing and the one JSON line -- with ``--device cpu`` (a stub model and gloo in place of the HIP model and
    RCCL: a rehearsal, its line says so).  Exactly one JSON line on stdout, n_gpus 2, both ranks in the warm-up all-reduce; a
    rank that dies makes the whole command exit non-zero."""
    import json
    import subprocess

    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--device", "cpu", "--steps", "3", "--warmup", "1"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    run = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300, check=False)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, run.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["backend"] == "gloo" and line["config"]["collective_ranks"] == 2 and line["config"]["parallelism"] == "dp2"
    assert line["config"]["global_batch"] == 2 * 4 and line["value"] > 0 and line["higher_is_better"] is True
    assert "rehearsal" in line["data"] and line["vs_baseline"] is None
    assert abs(line["value"] - line["config"]["global_batch"] * line["config"]["seq_len"] / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    dead = subprocess.run(cmd, capture_output=True, text=True, env=dict(env, MTRSSM_BENCH_REHEARSAL_DIE_RANK="1"), timeout=300, check=False)
    assert dead.returncode != 0
    assert not [ln for ln in dead.stdout.splitlines() if ln.startswith("{")]
