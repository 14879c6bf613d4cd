"""Phase timing of the cluster scan (profiling helper): s_memtime stamps of workgroup 0 at the phase boundaries of timesteps
8..11 (csrc/mrssm_cluster.hip: MTRSSM_CLU_STAMP).  Usage on the GPU box: python tools/cluster_probe.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from multimodal_mtrssm_amd import _lib

dev = "cuda:0"
model = bench.build_model(dev)
batch = bench.synthetic_batch(64, dev, 1)
lib = _lib.load()
buf = torch.zeros(64, dtype=torch.int64, device=dev)
fn = lib.mtrssm_debug_set_cluster_profile
fn.argtypes, fn.restype = [ctypes.c_void_p], ctypes.c_int
for it in range(3):
    assert fn(buf.data_ptr()) == 0
    with torch.no_grad():
        model.shared_step(batch, None)
    torch.cuda.synchronize()
    assert fn(None) == 0
    st = buf.cpu().view(4, 16)[:, :8]
    names = ["h1", "gi/gh", "gates+X1", "heads0", "logit part+X2", "sum+cat block"]
    idx = [0, 1, 2, 3, 4, 5, 7]
    for row in st.tolist():
        d = [row[idx[i + 1]] - row[idx[i]] for i in range(6)]
        print("iter", it, "step cycles", row[7] - row[0], " ".join(f"{n}={v}" for n, v in zip(names, d)))
print("s_memtime runs at 100 MHz (10 ns per tick) on gfx950")
