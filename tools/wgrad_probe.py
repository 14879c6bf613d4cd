"""Profiling probe (not part of the product): weight gradient of one 3x3 residual-stack layer, repeated, with the
phase stamps of conv3x3_wgrad_resident_kernel (workgroup 0 and the middle workgroup; s_memtime ticks of 10 ns).

usage: python tools/wgrad_probe.py CIN COUT H W [FRAMES] [REPS]
"""
import ctypes
import sys

import torch

from multimodal_mtrssm_amd import _lib, conv

cin, cout, h, w = map(int, sys.argv[1:5])
frames = int(sys.argv[5]) if len(sys.argv) > 5 else 3200
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(frames, cin, h, w, generator=g).to(dev)
wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).to(dev).requires_grad_()
b = torch.randn(cout, generator=g).to(dev).requires_grad_()
lib = _lib.load()
fn = lib.mtrssm_debug_set_resident_profile
fn.argtypes, fn.restype = [ctypes.c_void_p], ctypes.c_int
prof = torch.zeros(64, dtype=torch.int64, device=dev)
_lib.TIMERS.enable()
for _ in range(reps):
    y = conv.conv2d(x, wt, b, stride=1, padding=1, pre_act=True, act=2)
    y.backward(torch.ones_like(y))
torch.cuda.synchronize()
assert fn(prof.data_ptr()) == 0
y = conv.conv2d(x, wt, b, stride=1, padding=1, pre_act=True, act=2)
y.backward(torch.ones_like(y))
torch.cuda.synchronize()
assert fn(None) == 0
st = prof.cpu().tolist()
for name, o in (("workgroup 0", 32), ("middle workgroup", 40)):
    t = st[o:o + 4]
    print(name, "prologue %.2f us, loop %.2f us, epilogue %.2f us" % ((t[1] - t[0]) / 100, (t[2] - t[1]) / 100, (t[3] - t[2]) / 100))
print("start skew middle - first: %.2f us" % ((st[40] - st[32]) / 100))
for name, rec in _lib.TIMERS.summary().items():
    if "grad" in name:
        print(name, "avg %.1f us" % (rec["avg_ms"] * 1e3))
