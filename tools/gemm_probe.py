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
import ctypes
import os

if os.environ.get("STAMPS"):  # needs a library built with the stamped debug kernel (mtrssm_debug_set_gemm_profile)
    lib = _lib.load()
    fn = lib.mtrssm_debug_set_gemm_profile
    fn.argtypes, fn.restype = [ctypes.c_void_p], ctypes.c_int
    prof = torch.zeros(64, dtype=torch.int64, device=dev)
    assert fn(prof.data_ptr()) == 0
    gemm(a, b, c, a_rmajor=a_rm, b_rmajor=b_rm, mfma_split=pieces, split_r=split_r, accumulate=split_r > 1)
    torch.cuda.synchronize()
    assert fn(None) == 0
    st = prof.cpu().view(8, 8).tolist()
    for k in range(4):
        r = st[k]
        print("k-step", 8 + k, "to_lds", r[1] - r[0], "issue loads", r[2] - r[1], "lds reads", r[3] - r[2], "mfma chain issue", r[4] - r[3],
              "barrier", r[5] - r[4], "total", r[5] - r[0], "to next", (st[k + 1][0] - r[0]) if k < 3 else "")
for name, rec in _lib.TIMERS.summary().items():
    if "gemm" in name:
        print(m, n, r, a_rm, b_rm, "pieces", pieces, "split_r", split_r, name, round(rec["avg_ms"] * 1e3, 1), "us", round(rec["flops"] / rec["total_ms"] / 1e9, 1), "TFLOP/s")
