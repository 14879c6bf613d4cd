import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
import multimodal_mtrssm_amd as mt
from multimodal_mtrssm_amd import scan
from multimodal_mtrssm_amd.optim import FlatParameters
from multimodal_mtrssm_amd.graph import CapturedTrainStep
dev="cuda:0"
model=bench.build_model(dev)
flat=FlatParameters(model, extra=8); dp=mt.FlatDataParallel(flat); opt=mt.FlatAdamW(flat, lr=1e-3, clip_norm=10.0)
batch=bench.synthetic_batch(64, dev, 1000)
src=dp.noise_source(seed=7)
cap=CapturedTrainStep(model, flat, opt, dp, batch, src)
def dump(tag):
    torch.cuda.synchronize()
    for k,ws in scan._CLUSTER_WS.items():
        print(tag, k[0] if isinstance(k[0], str) else "fwd", ws[:2].view(torch.int32).tolist())
dump("after capture")
marks=[torch.cuda.Event(enable_timing=True) for _ in range(13)]
for i in range(int(os.environ.get("DBG_N","12"))):
    if os.environ.get("DBG_EVENTS"): marks[i].record()
    out=cap.step()
    if os.environ.get("DBG_SYNC"): torch.cuda.synchronize()
dump("end")
print(float(out["loss"]))
