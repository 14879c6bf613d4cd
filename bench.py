"""bench.py -- seq-steps/s of the MoPoE-MRSSM train step on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: starts its own N worker processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                   (or under a launcher: RANK / WORLD_SIZE from the env)

One "step" = one full train step (encoders -> initial state -> T-step scan -> decoders -> Gaussian NLL +
KL -> backward -> [one RCCL all-reduce] -> global-norm clip + AdamW) on one synthetic batch that is already
resident in HBM.  Workload = BASELINE.json configs[1]: MoPoE-MRSSM B=64 per GPU, T=50, deter=200,
stoch=30 (6 categoricals x 5 classes), action=4, vision 1x64x64 + audio 1x128x32 (128 mel bins x 32
frames); the dims BASELINE leaves open are build-chosen and printed in ``config``.  N > 1 shards the
sequence batch data-parallel (weak scaling: 64 sequences per GPU), one process per GPU.

Rank 0 prints ONE JSON line.  ``roofline`` is for the dominant hand-written kernel of the step (HIP events
on the launch stream, 3 steps right after the timed region); ``cpu_baseline`` times the oracle
(``oracle/ref_model.py``, the op-for-op eager restatement of the reference loop) on this box's host cores at the
SAME batch (B = 64); ``elbo_rel_delta`` is the loss of the trained weights against that oracle on a 2-row sub-batch.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16 = 16x the fp32 MFMA rate)
# MFMA instructions issued per algorithmic multiply-add block, by conv MFMA mode (csrc/conv_split.h)
MFMA_PRODUCTS = {"f32": 1, "bf16x3": 6, "bf16x2": 3, "bf16": 1}

WORKLOADS = {
    # BASELINE configs[1] / configs[2]: the metric's config
    "base": dict(batch_per_gpu=64, steps=50, deter=200, hidden=200, classes=5, cats=6, action=4, embed=256,
                 vision=(1, 64, 64), audio=(1, 128, 32)),
    # BASELINE configs[4] "Large" (deter = 1024, stoch = 128 = 16 x 8, B = 256, T = 100 on 8 GPUs): the per-GPU shard, B = 32
    "large": dict(batch_per_gpu=32, steps=100, deter=1024, hidden=1024, classes=8, cats=16, action=4, embed=1024,
                  vision=(1, 64, 64), audio=(1, 128, 32)),
}
WORKLOAD = WORKLOADS["base"]  # main() switches it for --model large


def build_model(device: str, kind: str = "mrssm"):  # noqa: ANN201
    import multimodal_mtrssm_amd as mt
    from multimodal_mtrssm_amd.factory import decoder_config, encoder_config

    w = WORKLOAD
    torch.manual_seed(42)  # yaml: seed_everything: 42
    kind = "mrssm" if kind == "large" else kind
    if kind == "mmtrssm":  # BASELINE configs[2]: two-timescale MTState variant, same dims (ld = hd = 200, ls = hs = 30)
        feat = 2 * (w["deter"] + w["classes"] * w["cats"])
        model = mt.make_mmtrssm(
            hd=w["deter"], hs=(w["classes"], w["cats"]), ld=w["deter"], ls=(w["classes"], w["cats"]), hidden=w["hidden"],
            action=w["action"], embed=w["embed"], enc_audio=encoder_config(w["audio"], w["embed"]),
            enc_vision=encoder_config(w["vision"], w["embed"]), dec_audio=decoder_config(feat, w["audio"]),
            dec_vision=decoder_config(feat, w["vision"]))
        return model.to(device)
    feat = w["deter"] + w["classes"] * w["cats"]
    model = mt.make_mrssm(
        deter=w["deter"], hidden=w["hidden"], classes=w["classes"], cats=w["cats"], action=w["action"], embed=w["embed"],
        enc_audio=encoder_config(w["audio"], w["embed"]), enc_vision=encoder_config(w["vision"], w["embed"]),
        dec_audio=decoder_config(feat, w["audio"]), dec_vision=decoder_config(feat, w["vision"]),
    )
    return model.to(device)


def synthetic_batch(batch: int, device: str, seed: int) -> tuple[torch.Tensor, ...]:
    """targets U(-1,1), inputs = targets + N(0, 0.1^2), actions N(0,1)."""
    w = WORKLOAD
    g = torch.Generator().manual_seed(seed)
    t = w["steps"]
    act_t = torch.randn(batch, t, w["action"], generator=g)
    aud_t = torch.rand(batch, t, *w["audio"], generator=g) * 2 - 1
    vis_t = torch.rand(batch, t, *w["vision"], generator=g) * 2 - 1
    act_i = act_t + 0.1 * torch.randn(act_t.shape, generator=g)
    aud_i = aud_t + 0.1 * torch.randn(aud_t.shape, generator=g)
    vis_i = vis_t + 0.1 * torch.randn(vis_t.shape, generator=g)
    return tuple(x.to(device) for x in (act_i, aud_i, vis_i, act_t, aud_t, vis_t))


def measured_traffic(kernel: str) -> float | None:
    """HBM bytes per launch of ``kernel`` from the committed PMC passes (profiles/*_pmc_summary.json: separate
    ``rocprofv3 --pmc FETCH_SIZE`` / ``WRITE_SIZE`` runs of this script), with the gfx950 correction of
    MI355X_MICROARCH.md (FETCH_SIZE counts half the bytes of a coalesced stream): 2 * FETCH + WRITE.  None when no
    profile has been collected for the kernel.  PMC collection cannot run inside the timed region."""
    best = None
    for path in sorted((ROOT / "profiles").glob("*_pmc_summary.json")):
        try:
            row = json.loads(path.read_text())["kernels"].get(kernel)
        except (OSError, ValueError, KeyError):
            continue
        if row:
            best = (2.0 * row["FETCH_SIZE_KB_avg"] + row["WRITE_SIZE_KB_avg"]) * 1024.0
    return best


def host_cores() -> int:
    """CPU share of this process: the affinity mask, capped at the 16-core share a one-GPU box gets."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def oracle_model(kind: str):  # noqa: ANN201
    """The oracle (oracle/ref_model.py) at the bench's exact dims -- the CHECKER: used by cpu_baseline() and elbo_delta() only."""
    from oracle.cases import CASES, build_model

    case = CASES[{"mrssm": "mrssm_bench", "mmtrssm": "mmtrssm_bench", "large": "mrssm_large_bench"}[kind]]
    return case, build_model(case)


def cpu_baseline(kind: str, seconds_budget: float = 30.0) -> dict[str, object]:
    """The oracle's train step (fwd + bwd + clip + AdamW) on the host cores at the bench's OWN batch (B = 64, T = 50, the
    same synthetic tensors rank 0 trains on): like for like with `value`.  One warm-up step, then as
    many timed steps as fit the budget (at least one, at most three); the median is reported."""
    case, model = oracle_model(kind)
    w = WORKLOAD
    cores = host_cores()
    torch.set_num_threads(cores)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    b = w["batch_per_gpu"]
    batch = synthetic_batch(b, "cpu", 1000)
    from oracle.cases import build_noise
    noise = build_noise(case, 7, batch=b, steps=w["steps"])

    def step() -> None:
        opt.zero_grad()
        out = model.shared_step(batch, noise, wasteful=True) if kind != "mmtrssm" else model.shared_step(batch, noise)
        out["loss"].backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        opt.step()

    t0 = time.perf_counter()
    step()  # warm-up
    warm = time.perf_counter() - t0
    times: list[float] = []
    while len(times) < 3 and (not times or warm + sum(times) + max(times) < seconds_budget):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": b * w["steps"] / med, "unit": "seq-steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle/ref_model.py train step (fwd+bwd+clip+AdamW; the reference's discarded draws included for MRSSM), "
                      f"B={b} T={w['steps']}: the bench's own batch, dims and frame sizes; median of {len(times)} after 1 warm-up "
                      f"({warm:.1f}s), {cores} torch threads"}


def elbo_delta(model, batch: tuple[torch.Tensor, ...], kind: str, rows: int = 2) -> dict[str, object]:  # noqa: ANN001
    """ELBO delta vs the reference path (BASELINE metric's second half; ``core.py:187-221`` / mmtrssm ``core.py:563-606``):
    the model's CURRENT weights are copied by state-dict name into the oracle, both run ``shared_step`` on the first ``rows``
    sequences of the training batch with the same injected uniforms (seed screened so that no draw sits within 1e-4 of a CDF
    edge), and every loss term's relative difference is reported.  Outside the timed region; rank 0 only."""
    from oracle.cases import screened_noise

    case, oracle = oracle_model(kind)
    oracle.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()}, strict=True)
    sub = tuple(x[:rows] for x in batch)
    cpu_sub = tuple(x.cpu() for x in sub)
    torch.set_num_threads(host_cores())
    noise, margin, seed = screened_noise(case, oracle, cpu_sub)
    with torch.no_grad():
        ref = oracle.shared_step(cpu_sub, noise)
        out = model.shared_step(sub, {k: v.to(sub[0].device) for k, v in noise.items()})
    rel = {k: abs(float(out[k]) - float(ref[k])) / max(abs(float(ref[k])), 1e-12) for k in out}
    return {"loss": rel["loss"], "terms": rel, "reference_loss": float(ref["loss"]), "rows": rows, "noise_seed": seed,
            "sampling_margin": margin, "tolerance": 1e-4, "ok": bool(max(rel.values()) <= 1e-4)}


def feed_benchmark(device: str, batches: int = 20) -> dict[str, object]:
    """SURVEY section 8f-4, measured beside the train step (never part of `value`): one [64, 50, ...] 6-tuple batch per
    iteration from an HBM-resident store of 256 synthetic episodes x 60 steps (TakeFirstN(50) + GaussianNoise(0.1) fused in
    mtrssm_episode_gather; shuffled epoch order; normals drawn by torch.randn on the device)."""
    from multimodal_mtrssm_amd.dataset import DeviceEpisodeLoader, _Stream
    from multimodal_mtrssm_amd.transform import Compose, GaussianNoise, TakeFirstN

    w = WORKLOAD
    n, t_full, t, b = 256, 60, w["steps"], w["batch_per_gpu"]
    g = torch.Generator(device=device).manual_seed(3)
    shapes = [(w["action"],), w["audio"], w["vision"]]
    noisy, clean = Compose([TakeFirstN(t), GaussianNoise(0.1)]), Compose([TakeFirstN(t)])
    streams = tuple(_Stream(torch.rand(n, t_full, *sh, device=device, generator=g) * 2 - 1, noisy, clean) for sh in shapes)
    loader = DeviceEpisodeLoader(streams, b, shuffle=True)
    order = torch.randperm(n, device=device)
    for i in range(3):
        loader.batch(order[i * b % n: i * b % n + b].contiguous())
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for i in range(batches):
        lo = (i * b) % (n - b + 1)
        loader.batch(order[lo: lo + b].contiguous())
    stop.record()
    torch.cuda.synchronize()
    ms = start.elapsed_time(stop) / batches
    elems = b * t * sum(int(torch.tensor(sh).prod()) for sh in shapes)
    # algorithmic bytes per batch: store read once, normals written by the generator and read once, input + target written
    nbytes = 4.0 * elems * 5
    return {"value": b * t / (ms * 1e-3), "unit": "seq-steps/s", "ms_per_batch": ms, "achieved_GBps": nbytes / (ms * 1e-3) / 1e9,
            "hbm_frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "what": f"6-tuple batch B={b} T={t} from {n} HBM-resident episodes x {t_full} steps; gather + TakeFirstN + GaussianNoise fused (mtrssm_episode_gather) + torch.randn"}


class _StubModel(torch.nn.Module):
    """``--device cpu`` rehearsal only (tests/test_parallel_gloo.py): a CPU module with the model's bench-facing surface
    (``shared_step(batch, noise)`` -> loss dict, ``noise_shapes``), so that THIS file's argument parsing, self-launch, sharded
    step, barriers, MAX-over-ranks timing and JSON line run end to end on a box without a GPU.  Never a measurement."""

    def __init__(self) -> None:
        super().__init__()
        self.body = torch.nn.Linear(4, 8)
        self.head = torch.nn.Linear(8, 1)
        self.dead = torch.nn.Linear(3, 3)  # never called: like MMTRSSM's l_posterior / dummy transition

    @staticmethod
    def noise_shapes(batch: int, steps: int) -> dict[str, tuple[int, ...]]:
        return {"u_init": (batch, 2), "u_post": (batch, steps, 2)}

    def shared_step(self, batch: tuple[torch.Tensor, ...], noise: dict[str, torch.Tensor]) -> dict[str, torch.Tensor]:
        h = torch.tanh(self.body(batch[0]) + noise["u_post"].mean(-1, keepdim=True))
        recon = (self.head(h).squeeze(-1) - batch[3][..., 0]).square().mean()
        kl = noise["u_init"].mean() * 0.0 + h.square().mean()
        return {"recon": recon, "kl": kl, "loss": recon + kl}


class _StubOptimizer:
    """SGD over the flat buffer with FlatAdamW's call surface (the fused AdamW is a HIP kernel: there is none on the CPU)."""

    def __init__(self, flat) -> None:  # noqa: ANN001
        self.flat = flat

    def zero_grad(self) -> None:
        self.flat.zero_grad()

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0) -> None:
        self.flat.param.sub_(1e-3 * grad_scale * self.flat.grad)


def _free_port() -> int:
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def launch_workers(n: int, script: Path | None = None, argv: list[str] | None = None) -> int:
    """``python bench.py --gpus N`` from a plain shell: start N worker processes (one per GPU) as a CHILD
    ``torch.distributed.run`` job -- before this process has touched the GPU (importing torch does not) -- and pass rank 0's
    JSON line through (the children inherit stdout).  Returns the job's exit code (non-zero if any rank failed).
    ``script`` / ``argv`` default to this file and this process's own arguments (tests/test_parallel_gloo.py drives a CPU
    worker through the same code)."""
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, n))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(Path(script or __file__).resolve()), *(sys.argv[1:] if argv is None else argv)]
    return subprocess.run(cmd, env=env, check=False).returncode


def parse_args() -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-elbo-check", action="store_true")
    ap.add_argument("--model", choices=("mrssm", "mmtrssm", "large"), default="mrssm",
                    help="mrssm = BASELINE configs[1] (the metric's config); mmtrssm = configs[2] (MTState variant); large = the "
                         "per-GPU shard of configs[4] (MoPoE-MRSSM deter = hidden = embed = 1024, stoch 16 x 8, B = 32, T = 100)")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group even for one rank (exercises the N>1 code path)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="nccl = RCCL over xGMI (the product path); gloo only to rehearse the launcher and the sharded step on a box with fewer GPUs")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (needs --backend gloo: RCCL refuses two ranks on one GPU)")
    ap.add_argument("--device", choices=("cuda", "cpu"), default="cuda",
                    help="cpu = REHEARSAL of this file's launch / sharding / timing / JSON path with a stub model over gloo on a box "
                         "without a GPU (tests/test_parallel_gloo.py); its line says data = 'stub' and is never a measurement")
    ap.add_argument("--graph", choices=("on", "off"), default="off",
                    help="on: the train step (zero_grad, forward, backward, and with one rank the optimizer) is captured once in a "
                         "hipGraph and replayed (graph.CapturedTrainStep); off: every launch enqueued by the host each step")
    ap.add_argument("--single-stream", action="store_true", help="(default and only mode; kept for the commands quoted in profiles/round1_*)")
    ap.add_argument("--conv-mfma", choices=("bf16x2", "bf16x3", "f32", "bf16"), default="bf16x2",
                    help="conv MFMA operand format: bf16x2 = two bf16 pieces per fp32 operand, three products, fp32 accumulate (the mode "
                         "the parity tests run in; errors vs golden 7e-7 / 1.3e-7 / 8e-6, tests/mode_errors.py), bf16x3 = three pieces, "
                         "six products (~2^-24), f32 = fp32 MFMA, bf16 = plain bf16 operands (outside the posterior tolerance: dtype bf16)")
    return ap.parse_args()


DTYPE_LABEL = {"bf16x2": "bf16x2-split, fp32 accumulate", "bf16x3": "bf16x3-split, fp32 accumulate", "f32": "f32", "bf16": "bf16"}


def main() -> None:  # noqa: PLR0914, PLR0915
    args = parse_args()
    global WORKLOAD  # noqa: PLW0603
    WORKLOAD = WORKLOADS["large" if args.model == "large" else "base"]
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_workers(args.gpus))  # nothing above touched the GPU

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        msg = f"--gpus {args.gpus} but WORLD_SIZE={world}"
        raise SystemExit(msg)
    if args.share_device and args.backend != "gloo":
        raise SystemExit("--share-device needs --backend gloo")
    if args.share_device:  # two processes on one GPU: the cluster scan's workgroups could not all be resident
        os.environ["MTRSSM_SCAN_CLUSTER"] = "0"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cpu = args.device == "cpu"
    if cpu:
        args.backend, args.no_cpu_baseline, args.no_elbo_check = "gloo", True, True
        device = "cpu"
    else:
        dev_index = 0 if args.share_device else local_rank
        torch.cuda.set_device(dev_index)
        device = f"cuda:{dev_index}"

    def sync() -> None:
        if not cpu:
            torch.cuda.synchronize()

    use_dist = world > 1 or args.force_dist
    rccl_ranks = collective_ranks = 0
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL prints a version banner on stdout when its communicator comes up; stdout carries the ONE JSON line, so the
        # banner goes to stderr (fd-level: the library writes through C stdio)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            warm = torch.ones(1, device=device)
            dist.all_reduce(warm)
            sync()
            collective_ranks = int(warm.item())  # ranks that took part in a real all-reduce of the chosen backend
            rccl_ranks = collective_ranks if args.backend == "nccl" else 0
            assert collective_ranks == dist.get_world_size()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    import multimodal_mtrssm_amd as mt
    from multimodal_mtrssm_amd import conv, scan
    from multimodal_mtrssm_amd.optim import FlatParameters

    conv.set_mfma_mode(args.conv_mfma)

    w = WORKLOAD
    if cpu:
        w = dict(w, batch_per_gpu=4, steps=6, vision=(1, 2, 2), audio=(1, 2, 2))
        torch.manual_seed(42)
        model = _StubModel()
        die = os.environ.get("MTRSSM_BENCH_REHEARSAL_DIE_RANK")  # the launcher must turn a dead rank into a non-zero exit
        if die is not None and int(die) == rank:
            os._exit(3)  # noqa: SLF001
    else:
        model = build_model(device, args.model)
    flat = FlatParameters(model, extra=8)
    dp = mt.FlatDataParallel(flat)
    dp.broadcast_parameters(0)
    opt = _StubOptimizer(flat) if cpu else mt.FlatAdamW(flat, lr=1e-3, clip_norm=10.0)
    b, t = w["batch_per_gpu"], w["steps"]
    if cpu:
        g0 = torch.Generator().manual_seed(1000 + rank)
        batch = tuple(torch.randn(b, t, n, generator=g0) for n in (4, 4, 4, 4, 4, 4))
    else:
        batch = synthetic_batch(b, device, seed=1000 + rank)  # each rank owns its own 64 sequences (weak scaling)
    # sampling uniforms keyed by GLOBAL batch row: the same generator state on every rank, the whole
    # global batch drawn, this rank's rows kept -- a row's trajectory does not depend on the number of ranks
    noise_source = dp.noise_source(seed=7)
    shapes = model.noise_shapes(b, t)

    def eager_step() -> dict[str, torch.Tensor]:
        noise = noise_source.draw(shapes)
        opt.zero_grad()
        out = model.shared_step(batch, noise)
        out["loss"].backward()
        scalars = dp.sync({k: out[k] for k in out})
        opt.step(grad_scale=dp.grad_scale)
        return scalars

    train_step = eager_step
    if args.graph == "on":
        from multimodal_mtrssm_amd.graph import CapturedTrainStep

        captured = CapturedTrainStep(model, flat, opt, dp, batch, noise_source)
        train_step = captured.step

    def barrier() -> None:
        if use_dist:
            dist.barrier()
        sync()

    class _HostMark:  # --device cpu: wall-clock marks with torch.cuda.Event's two methods
        def record(self) -> None:
            self.t = time.perf_counter()

        def elapsed_time(self, other: "_HostMark") -> float:
            return (other.t - self.t) * 1e3

    for _ in range(args.warmup):
        train_step()
    marks = [_HostMark() if cpu else torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]  # one per STEP (none per launch)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        marks[i].record()
        scalars = train_step()
    marks[args.steps].record()
    barrier()
    elapsed = time.perf_counter() - t0
    scan.check_cluster_status()  # a cluster-scan exchange that timed out would have left invalid results
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2]
    # Kernel durations, right AFTER the timed region (HIP events around every launch cost the step ~3 %): 3 more steps
    # with an event pair around each library launch, on the launch stream (agrees with rocprofv3 --kernel-trace --stats
    # of this command, profiles/).
    kernel_ms: dict[str, dict[str, float]] = {}
    if rank == 0 and not cpu:
        scan.KERNEL_TIMERS.enable()
    for _ in range(3):  # EVERY rank takes these steps (each holds an all-reduce); only rank 0 times its launches
        eager_step()  # never the captured graph: events cannot be recorded into a replay
    sync()
    if rank == 0 and not cpu:
        kernel_ms = scan.KERNEL_TIMERS.summary()
        scan.KERNEL_TIMERS.disable()
    tt = torch.tensor([elapsed, median_ms], device=device, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed, median_ms = float(tt[0].item()), float(tt[1].item())

    if rank == 0:
        seq_steps = b * world * t
        ms = elapsed / args.steps * 1e3
        # dominant hand-written kernel of the step = the device kernel with the largest total time (HIP events on the
        # launch stream).  Its algorithmic work is stated by the caller of each launch (conv.py / scan.py): conv kernels
        # are MFMA-bound, the scan is latency-bound and is priced against HBM.
        timed = {k: v for k, v in kernel_ms.items() if v["flops"] or v["bytes"]}
        name, row = max(timed.items(), key=lambda kv: kv[1]["total_ms"]) if timed else ("none", None)
        n_steps = 3
        roof: dict[str, object] = {"kernel": name}
        if row:
            secs = row["total_ms"] * 1e-3
            if "conv" in name and "thin" not in name:
                # algorithmic FLOPs (each multiply-add counted once) against the rate at which the MFMA pipe can deliver
                # them in this operand format: fp32 MFMA 157.3 TFLOP/s; split kernels issue `products` bf16 MFMAs per
                # algorithmic block, so their ceiling is the dense bf16 peak / products (bf16x2: 2500 / 3 = 833).  The roof
                # that binds is the roofline model's: a kernel whose arithmetic intensity (algorithmic FLOPs per
                # algorithmic byte) lies below the ridge peak / HBM rate is priced against HBM, above it against the MFMA.
                split = any(tag in name for tag in ("split_kernel", "resident_kernel", "stream_kernel", "staged_kernel"))
                products = MFMA_PRODUCTS[args.conv_mfma] if split else 1
                peak = MFMA_BF16_PEAK_TFLOPS / products if split else MFMA_F32_PEAK_TFLOPS
                achieved = row["flops"] / secs / 1e12
                gbs = row["bytes"] / secs / 1e9
                intensity = row["flops"] / row["bytes"] if row["bytes"] else float("inf")
                ridge = peak * 1e12 / (HBM_PEAK_GBS * 1e9)
                if intensity >= ridge:
                    roof.update(bound="mfma", achieved=achieved, peak=peak, unit="TFLOP/s", frac=achieved / peak)
                else:
                    roof.update(bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS)
                roof.update(mfma_products_per_block=products, mfma_frac=achieved / peak, hbm_frac=gbs / HBM_PEAK_GBS,
                            flops_per_byte=intensity, ridge_flops_per_byte=ridge,
                            frac_of_raw_bf16_peak=achieved / MFMA_BF16_PEAK_TFLOPS)
            else:
                achieved = row["bytes"] / secs / 1e9
                roof.update(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS)
            roof.update(traffic=measured_traffic(name), launches_per_step=row["launches"] / n_steps, avg_us=row["avg_ms"] * 1e3,
                        algorithmic_flops_per_launch=row["flops"] / row["launches"],
                        algorithmic_bytes_per_launch=row["bytes"] / row["launches"],
                        share_of_step=row["total_ms"] / n_steps / ms)
        # ---- the step against the machine (never a substitute for the dominant-kernel entry above, which may be latency-bound):
        # every library launch states its algorithmic FLOPs and bytes (conv.py / scan.py / linear.py); their sums per step,
        # the end-to-end algorithmic I/O of SURVEY section 8d (103 KB fp32 per seq-step: inputs, targets, states, once) and the
        # MFMA ceiling of the operand format (dense bf16 peak / products per multiply-add block)
        products = MFMA_PRODUCTS[args.conv_mfma]
        ceiling = MFMA_BF16_PEAK_TFLOPS / products if args.conv_mfma != "f32" else MFMA_F32_PEAK_TFLOPS
        step_s = ms * 1e-3
        flops_step = sum(v["flops"] for v in kernel_ms.values()) / n_steps
        bytes_step = sum(v["bytes"] for v in kernel_ms.values()) / n_steps
        io_step = 103.0e3 * b * t
        roof["step"] = {
            "algorithmic_flops": flops_step, "achieved_TFLOPs": flops_step / step_s / 1e12, "mfma_ceiling_TFLOPs": ceiling,
            "frac_of_mfma_ceiling": flops_step / step_s / 1e12 / ceiling, "frac_of_raw_bf16_peak": flops_step / step_s / 1e12 / MFMA_BF16_PEAK_TFLOPS,
            "sum_of_kernel_algorithmic_bytes": bytes_step, "its_GBps": bytes_step / step_s / 1e9,
            "its_frac_of_hbm": bytes_step / step_s / 1e9 / HBM_PEAK_GBS,
            "end_to_end_algorithmic_bytes": io_step, "end_to_end_GBps": io_step / step_s / 1e9,
            "inter_kernel_traffic_ratio": bytes_step / io_step,
            "library_launches_per_step": sum(v["launches"] for v in kernel_ms.values()) / n_steps,
            "library_kernel_ms_per_step": sum(v["total_ms"] for v in kernel_ms.values()) / n_steps,
        }
        # the largest kernel that a pipe bounds (the dominant one above may be the latency-bound recurrence)
        best = None
        for k, v in kernel_ms.items():
            if "conv" not in k or "thin" in k or not v["flops"] or not v["total_ms"]:
                continue
            split = any(tag in k for tag in ("split_kernel", "resident_kernel", "stream_kernel", "staged_kernel"))
            peak = MFMA_BF16_PEAK_TFLOPS / (products if split else 1) if split else MFMA_F32_PEAK_TFLOPS
            secs = v["total_ms"] * 1e-3
            tf, gbs = v["flops"] / secs / 1e12, v["bytes"] / secs / 1e9
            ridge = peak * 1e12 / (HBM_PEAK_GBS * 1e9)
            bound_mfma = v["flops"] / max(v["bytes"], 1.0) >= ridge
            frac = tf / peak if bound_mfma else gbs / HBM_PEAK_GBS
            if best is None or v["total_ms"] > best[1]["total_ms"]:
                best = (k, v, "mfma" if bound_mfma else "hbm", tf, gbs, peak, frac)
        if best is not None:
            k, v, bound, tf, gbs, peak, frac = best
            roof["largest_pipe_bound"] = {"kernel": k, "bound": bound, "ms_per_step": v["total_ms"] / n_steps, "avg_us": v["avg_ms"] * 1e3,
                                          "achieved_TFLOPs": tf, "achieved_GBps": gbs, "peak": peak if bound == "mfma" else HBM_PEAK_GBS,
                                          "unit": "TFLOP/s" if bound == "mfma" else "GB/s", "frac": frac, "traffic": measured_traffic(k)}
        roof["kernels"] = {k: {"launches_per_step": v["launches"] / 3, "avg_us": round(v["avg_ms"] * 1e3, 1),
                               "ms_per_step": round(v["total_ms"] / 3, 3),
                               "tflops": round(v["flops"] / (v["total_ms"] * 1e-3) / 1e12, 2) if v["total_ms"] else None,
                               "gbs": round(v["bytes"] / (v["total_ms"] * 1e-3) / 1e9, 1) if v["total_ms"] else None}
                           for k, v in sorted(kernel_ms.items(), key=lambda kv: -kv[1]["total_ms"])}
        line = {
            "metric": {"mrssm": "seq-steps/s (BxT) MoPoE-MRSSM train step", "mmtrssm": "seq-steps/s (BxT) MoPoE-MMTRSSM train step",
                       "large": "seq-steps/s (BxT) MoPoE-MRSSM Large train step"}[args.model],
            "value": seq_steps / (elapsed / args.steps),
            "unit": "seq-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "ms_per_step_median": median_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": DTYPE_LABEL[args.conv_mfma],
            "data": "stub model on the CPU: a rehearsal of the launch / sharding / JSON path, NOT a measurement" if cpu else "synthetic",
            "config": {
                "workload": {"mrssm": "BASELINE configs[1]: MoPoE-MRSSM train step, B=64/GPU T=50 deter=200 stoch=30, vision 1x64x64 + audio 1x128x32 + action 4",
                             "mmtrssm": "BASELINE configs[2]: MoPoE-MMTRSSM (MTState, tau 2/4) train step, B=64/GPU T=50 ld=hd=200 ls=hs=30, same frames",
                             "large": "BASELINE configs[4] per-GPU shard: MoPoE-MRSSM Large train step, B=32/GPU (256 on 8) T=100 deter=hidden=embed=1024 "
                                      "stoch=128 (16x8), vision 1x64x64 + audio 1x128x32 + action 4"}[args.model],
                "global_batch": b * world, "seq_len": t, "parallelism": f"dp{world}",
                "hidden": w["hidden"], "embed": w["embed"], "categoricals_x_classes": f"{w['cats']}x{w['classes']}",
                "enc_channels": [8, 16, 32], "dec_channels": [32, 16, 1], "residual_blocks": 3, "activation": "ELU",
                "optimizer": "AdamW lr 1e-3 + clip 10 (fused HIP)", "params": flat.numel, "streams": 1,
                "launch": "one hipGraph replay per step" if args.graph == "on" else "eager (host enqueues every launch)",
                "backend": (args.backend if use_dist else "none"), "rccl_ranks": rccl_ranks, "collective_ranks": collective_ranks,
                "noise": "uniforms keyed by global batch row (parallel.GlobalRowNoise)",
                "conv_mfma": {"bf16x3": "fp32 tensors; conv MFMA operands as 3 bf16 pieces, 6 bf16-MFMA products, fp32 accumulate (~2^-24 per product)",
                              "bf16x2": "fp32 tensors; conv MFMA operands as 2 bf16 pieces (16 significant bits), 3 bf16-MFMA products, fp32 "
                                        "accumulate; the mode of the parity tests (vs golden: losses 7e-7 rel, posterior 1.3e-7, gradients 8e-6 "
                                        "of max); scan, losses and optimizer in fp32",
                              "f32": "fp32 MFMA", "bf16": "bf16 operands, fp32 accumulate; tensors, scan, losses, optimizer fp32"}[args.conv_mfma],
            },
            "loss": float(scalars["loss"]),
            "roofline": roof,
        }
        if not args.no_elbo_check:
            # "ELBO delta vs ref" (the metric's second half): the trained weights of THIS run against the oracle on a 2-row
            # sub-batch, outside the timed region
            delta = elbo_delta(model, batch, args.model)
            line["elbo_rel_delta"] = delta["loss"]
            line["elbo_check"] = delta
        if world == 1 and not cpu:
            line["data_feed"] = feed_benchmark(device)
        if not args.no_cpu_baseline and world == 1 and args.model != "large":  # (Large: one CPU step takes minutes)
            base = cpu_baseline(args.model)
            line["cpu_baseline"] = base
            line["speedup_vs_cpu"] = line["value"] / base["value"]
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
