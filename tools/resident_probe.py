"""Probe (not part of the product): a 3x3 / s1 / p1 layer on 8x8 planes, weight-resident gather kernel vs torch on the host.

usage: python tools/resident_probe.py CIN COUT [FRAMES] [REPS]
"""
import sys

import torch
import torch.nn.functional as F  # noqa: N812

from multimodal_mtrssm_amd import _lib, conv

cin, cout = int(sys.argv[1]), int(sys.argv[2])
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 6
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
import os
PRE = os.environ.get("PRE", "1") == "1"
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x = torch.randn(frames, cin, 8, 8, generator=g).requires_grad_()
wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).requires_grad_()
b = torch.randn(cout, generator=g).requires_grad_()
gout = torch.randn(frames, cout, 8, 8, generator=g)
gd = gout.to(dev)
xg, wg, bg = (t.detach().to(dev).requires_grad_() for t in (x, wt, b))
_lib.TIMERS.enable()
import ctypes
lib = _lib.load()
fn = lib.mtrssm_debug_set_resident_profile
fn.argtypes, fn.restype = [ctypes.c_void_p], ctypes.c_int
prof = torch.zeros(64, dtype=torch.int64, device=dev)
for _ in range(reps):
    xg.grad = None
    y = conv.conv2d(xg, wg, bg, stride=1, padding=1, pre_act=PRE, act=2)
    if not os.environ.get('FWD_ONLY'):
        y.backward(gd)
torch.cuda.synchronize()
if os.environ.get("STAMPS"):
    assert fn(prof.data_ptr()) == 0
    y = conv.conv2d(xg, wg, bg, stride=1, padding=1, pre_act=PRE, act=2)
    torch.cuda.synchronize()
    assert fn(None) == 0
    st = prof.cpu().tolist()
    print("prologue cycles:", st[1] - st[0], "weights", st[56] - st[0], "zero fill + bias", st[57] - st[56], "first image", st[1] - st[57])
    for k in range(6):
        r = st[2 + 8 * k: 2 + 8 * k + 6]
        if r[0]:
            print("tile", k, "unit 0 loop", r[1] - r[0], "copy out", r[2] - r[1], "unit 1", r[4] - r[2], "barrier", r[5] - r[4], "total", r[5] - r[0])
if frames <= 64:
    want = F.conv2d(F.elu(x) if PRE else x, wt, b, 1, 1)
    want.backward(gout)
    ef = (y.detach().cpu() - want.detach()).abs()
    eb = (xg.grad.cpu() - x.grad).abs()
    print("fwd max err", float(ef.max()), "bwd max err", float(eb.max()))
    if float(eb.max()) > 1e-3:
        bad = eb > 1e-3
        print("bad count", int(bad.sum()), "of", bad.numel())
        print("bad by frame", bad.sum((1, 2, 3)).tolist())
        print("bad by channel", bad.sum((0, 2, 3)).tolist())
        print("bad by y", bad.sum((0, 1, 3)).tolist(), "by x", bad.sum((0, 1, 2)).tolist())
        print("got", xg.grad.cpu()[0, 0], "\nwant", x.grad[0, 0])
for name, rec in _lib.TIMERS.summary().items():
    print(name, rec)
