"""Probe (not part of the product): the GEMM problems of one train step of a bench model, in launch order.
usage: python tools/gemm_trace.py [base|mmtrssm|large]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from multimodal_mtrssm_amd import linear

which = sys.argv[1] if len(sys.argv) > 1 else "base"
dev = "cuda:0"
if which == "large":
    bench.WORKLOAD = bench.WORKLOADS["large"]
model = bench.build_model(dev, {"base": "mrssm", "mmtrssm": "mmtrssm", "large": "large"}[which])
batch = bench.synthetic_batch(32 if which == "large" else 64, dev, 1)
log = []
orig = linear._problem
def traced(a, b, c, **kw):
    m, n = c.shape[0], c.shape[1]
    r = a.shape[0] if kw.get("a_rmajor") else a.shape[1]
    log.append((m, n, r, int(bool(kw.get("a_rmajor"))), int(bool(kw.get("b_rmajor"))), kw.get("mfma_split"), bool(kw.get("accumulate"))))
    return orig(a, b, c, **kw)
linear._problem = traced
for it in range(2):
    log.clear()
    out = model.shared_step(batch, None)
    nf = len(log)
    out["loss"].backward()
    linear.flush_deferred()
    torch.cuda.synchronize()
print("forward", nf, "backward", len(log) - nf)
for i, e in enumerate(log):
    print(("F " if i < nf else "B ") + "M=%5d N=%5d R=%5d ar=%d br=%d pieces=%s acc=%s  %.2f GF" % (*e, 2e-9 * e[0] * e[1] * e[2]))
