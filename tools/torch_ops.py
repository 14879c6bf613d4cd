"""Profiling helper (not part of the product): which torch operators still launch device kernels inside the train step,
how many per step, and from which source line (torch.profiler with stacks)."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench
import multimodal_mtrssm_amd as mt
from multimodal_mtrssm_amd.optim import FlatParameters

dev = "cuda:0"
model = bench.build_model(dev)
flat = FlatParameters(model, extra=8)
dp = mt.FlatDataParallel(flat)
opt = mt.FlatAdamW(flat, lr=1e-3, clip_norm=10.0)
batch = bench.synthetic_batch(64, dev, 1)


def step():
    opt.zero_grad()
    out = model.shared_step(batch, None)
    out["loss"].backward()
    dp.sync({k: out[k] for k in out})
    opt.step(grad_scale=dp.grad_scale)


for _ in range(4):
    step()
torch.cuda.synchronize()
STEPS = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(STEPS):
        step()
    torch.cuda.synchronize()
rows = collections.defaultdict(lambda: [0, 0.0])
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or not ev.kernels:
        continue
    where = "?"
    for fr in ev.stack:
        if ("multimodal_mtrssm_amd/" in fr or "bench.py" in fr) and "torch_ops.py" not in fr:
            where = fr[fr.index("multimodal_mtrssm_amd/") if "multimodal_mtrssm_amd/" in fr else fr.index("bench.py"):]
            break
    if where == "?" and ev.stack:
        where = "(autograd) " + ev.stack[0][-60:]
    key = (ev.name, where)
    rows[key][0] += len(ev.kernels)
    rows[key][1] += sum(k.duration for k in ev.kernels)
tot_n = tot_t = 0
for (name, where), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{n / STEPS:6.1f} launches/step {t / STEPS:8.1f} us/step  {name:28s} {where}")
    tot_n += n
    tot_t += t
print(f"total {tot_n / STEPS:.1f} launches/step, {tot_t / STEPS:.1f} us/step")
