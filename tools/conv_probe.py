"""Profiling probe (not part of the product): one conv layer of the bench workload, forward / backward, repeated.

usage: [STRIDE=2] python tools/conv_probe.py MODE CIN COUT H W K [FRAMES] [REPS] [bwd]
"""
import os
import sys

import torch

from multimodal_mtrssm_amd import _lib, conv

mode, cin, cout, h, w, k = sys.argv[1], *map(int, sys.argv[2:7])
frames = int(sys.argv[7]) if len(sys.argv) > 7 else 3200
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 10
bwd = len(sys.argv) > 9
conv.set_mfma_mode(mode)
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(frames, cin, h, w, generator=g).to(dev).requires_grad_(bwd)
wt = (torch.randn(cout, cin, k, k, generator=g) * 0.1).to(dev).requires_grad_(bwd)
b = torch.randn(cout, generator=g).to(dev).requires_grad_(bwd)
_lib.TIMERS.enable()
for _ in range(reps):
    y = conv.conv2d(x, wt, b, stride=int(os.environ.get("STRIDE", "1")), padding=k // 2, pre_act=True, act=2)
    if bwd:
        y.backward(torch.ones_like(y))
torch.cuda.synchronize()
for name, rec in _lib.TIMERS.summary().items():
    print(name, rec)
