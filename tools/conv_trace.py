"""Probe (not part of the product): every library launch of one train step of a bench model in launch order, with the conv
geometry of its first problem, the algorithmic bytes its caller states and its HIP-event time.
usage: python tools/conv_trace.py [base|mmtrssm|large] [filter]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from multimodal_mtrssm_amd import _lib

which = sys.argv[1] if len(sys.argv) > 1 else "base"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
dev = "cuda:0"
if which == "large":
    bench.WORKLOAD = bench.WORKLOADS["large"]
model = bench.build_model(dev, {"base": "mrssm", "mmtrssm": "mmtrssm", "large": "large"}[which])
batch = bench.synthetic_batch(32 if which == "large" else 64, dev, 1)
log = []


class Trace(_lib.KernelTimers):
    def call(self, name, fn, *args, flops=0.0, nbytes=0.0):  # noqa: ANN001, ANN002, ANN201
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = fn(*args)
        e.record()
        k = _lib.load().mtrssm_last_kernel()
        geoms = []
        for a in args:
            g = getattr(a, "_obj", None)
            if g is not None and hasattr(g, "Cout"):
                geoms.append("N%d C%d+%d %dx%d k%dx%d ss%d ts%d -> Cout%d %dx%d (q %dx%d os%d) pre%d act%d" % (
                    g.N, g.C, g.C2, g.Hs, g.Ws, g.KH, g.KW, g.SS, g.TS, g.Cout, g.Ho, g.Wo, g.Hq, g.Wq, g.OS, g.pre_act, g.act))
        log.append((name, k.decode() if k else name, geoms, flops, nbytes, s, e))
        return rc


_lib.TIMERS.__class__ = Trace
_lib.TIMERS.on = True
for it in range(3):
    log.clear()
    out = model.shared_step(batch, None)
    nf = len(log)
    out["loss"].backward()
    torch.cuda.synchronize()
for i, (name, k, geoms, flops, nbytes, s, e) in enumerate(log):
    us = s.elapsed_time(e) * 1e3
    if flt and flt not in k and flt not in name:
        continue
    print(("F " if i < nf else "B ") + f"{us:7.1f} us {nbytes / 1e6:7.1f} MB {nbytes / us / 1e6 if us else 0:5.2f} TB/s {flops / us / 1e6 if us else 0:6.1f} TF  {k[8:70] if k.startswith('mtrssm::') else k[:62]}")
    for g in geoms:
        print("          " + g)
