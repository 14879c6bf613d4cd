"""Probe (not part of the product): the decoders' last ConvTranspose layer alone (audio plane), 12 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_mtrssm_amd import conv
dev = "cuda:0"
hs, ws = (64, 16) if (len(sys.argv) < 2 or sys.argv[1] == "audio") else (32, 32)
x = torch.randn(3200, 16, hs, ws, device=dev)
w = torch.randn(16, 1, 4, 4, device=dev) * 0.1
b = torch.randn(1, device=dev)
with torch.no_grad():
    for _ in range(12):
        y = conv.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=0, pre_act=True, act=2)
torch.cuda.synchronize()
print(float(y.abs().mean()))
