"""Host-side profile of the train step (profiling helper): cProfile over a few steps, top functions by own time."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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

for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
