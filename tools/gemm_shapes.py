"""Probe (not part of the product): mtrssm_gemm on the GEMM shapes of the bench models, fp32 MFMA vs two bf16 pieces.
usage: python tools/gemm_shapes.py [base|large]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_mtrssm_amd import _lib
from multimodal_mtrssm_amd.linear import gemm

which = sys.argv[1] if len(sys.argv) > 1 else "base"
BT = 3200
# (name, M, N, R, a_rmajor, b_rmajor, accumulate)
base = [
    ("enc head fwd   Y=XW^T", BT, 256, 4096, 0, 0, 0), ("enc head dX    dYW", BT, 4096, 256, 0, 1, 0), ("enc head dW    dY^TX", 256, 4096, BT, 1, 1, 1),
    ("dec stem fwd", BT, 4096, 64, 0, 0, 0), ("dec stem dX", BT, 64, 4096, 0, 1, 0), ("dec stem dW", 4096, 64, BT, 1, 1, 1),
    ("dec stem0 fwd 230->64", BT, 64, 230, 0, 0, 0), ("scan proj pa 256->200", BT, 200, 256, 0, 0, 0), ("scan dW_hh 600x200", 600, 200, BT, 1, 1, 1),
    ("scan dW wh1 200x200", 200, 200, BT, 1, 1, 1), ("scan dW heads 30x200", 30, 200, BT, 1, 1, 1), ("d_h2 = d_gi W_ih", BT, 200, 600, 0, 1, 0),
    ("fused wf_t 200x600", 200, 600, 200, 1, 0, 0),
]
large = [
    ("enc head fwd", BT, 1024, 4096, 0, 0, 0), ("enc head dX", BT, 4096, 1024, 0, 1, 0), ("enc head dW", 1024, 4096, BT, 1, 1, 1),
    ("scan proj pa 1024->1024", BT, 1024, 1024, 0, 0, 0), ("scan dW_hh 3072x1024", 3072, 1024, BT, 1, 1, 1), ("scan dW wh1 1024x1024", 1024, 1024, BT, 1, 1, 1),
    ("scan dW heads 128x1024", 128, 1024, BT, 1, 1, 1), ("d_h2 = d_gi W_ih", BT, 1024, 3072, 0, 1, 0), ("fused wf_t 1024x3072", 1024, 3072, 1024, 1, 0, 0),
    ("dec stem0 fwd 1152->64", BT, 64, 1152, 0, 0, 0),
]
dev = torch.device("cuda:0")
for name, m, n, r, a_rm, b_rm, acc in (base if which == "base" else large):
    a = torch.randn((r, m) if a_rm else (m, r), device=dev)
    b = torch.randn((r, n) if b_rm else (n, r), device=dev)
    c = torch.zeros(m, n, device=dev)
    row = []
    for pieces in (0, 2, 3):
        for _ in range(3):
            gemm(a, b, c, a_rmajor=bool(a_rm), b_rmajor=bool(b_rm), mfma_split=pieces, accumulate=bool(acc))
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            gemm(a, b, c, a_rmajor=bool(a_rm), b_rmajor=bool(b_rm), mfma_split=pieces, accumulate=bool(acc))
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 100.0
        row.append(f"pieces={pieces}: {us:7.1f} us {2.0 * m * n * r / us / 1e6:6.1f} TF")
    mb = 4.0 * (m * r + n * r + m * n) / 1e6
    print(f"{name:28s} M={m:5d} N={n:5d} R={r:5d} ar={a_rm} br={b_rm} {mb:6.1f} MB ({mb / 5.5e-3 / 1e3:5.1f} us at 5.5 TB/s) | " + " | ".join(row))
