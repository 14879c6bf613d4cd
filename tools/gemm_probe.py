"""Probe (not part of the product): time mtrssm_gemm on one shape.  usage: python tools/gemm_probe.py M N R a_rm b_rm [pieces] [split_r]"""
import sys

import torch

from multimodal_mtrssm_amd import _lib
from multimodal_mtrssm_amd.linear import gemm

m, n, r, a_rm, b_rm = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == "1", sys.argv[5] == "1"
pieces = int(sys.argv[6]) if len(sys.argv) > 6 else 2
split_r = int(sys.argv[7]) if len(sys.argv) > 7 else 0
dev = torch.device("cuda:0")
a = torch.randn((r, m) if a_rm else (m, r), device=dev)
b = torch.randn((r, n) if b_rm else (n, r), device=dev)
c = torch.zeros(m, n, device=dev)
_lib.TIMERS.enable()
for _ in range(6):
    gemm(a, b, c, a_rmajor=a_rm, b_rmajor=b_rm, mfma_split=pieces, split_r=split_r, accumulate=split_r > 1)
torch.cuda.synchronize()
for name, rec in _lib.TIMERS.summary().items():
    if "gemm" in name:
        print(m, n, r, a_rm, b_rm, "pieces", pieces, "split_r", split_r, name, round(rec["avg_ms"] * 1e3, 1), "us", round(rec["flops"] / rec["total_ms"] / 1e9, 1), "TFLOP/s")
