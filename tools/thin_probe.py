import sys, torch
from multimodal_mtrssm_amd import _lib, conv
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
_lib.TIMERS.enable()
# first vision conv: 1 (+2 coords) -> 8, 3x3 s2 on 64x64; last deconv 16 -> 1 k4 s2 on 32x32
x = torch.randn(3200, 1, 64, 64, generator=g).to(dev)
cc = torch.randn(2, 64, 64, generator=g).to(dev)
w = (torch.randn(8, 3, 3, 3, generator=g) * 0.1).to(dev).requires_grad_()
b = torch.randn(8, generator=g).to(dev).requires_grad_()
xd = torch.randn(3200, 16, 32, 32, generator=g).to(dev).requires_grad_()
wd = (torch.randn(16, 1, 4, 4, generator=g) * 0.1).to(dev).requires_grad_()
bd = torch.randn(1, generator=g).to(dev).requires_grad_()
for _ in range(10):
    y = conv.conv2d(x, w, b, stride=2, padding=1, pre_act=False, act=2, coords=cc)
    y.backward(torch.ones_like(y))
torch.cuda.synchronize()
for name, rec in _lib.TIMERS.summary().items():
    if "weight_grad" in name: print("conv", name, round(rec["avg_ms"] * 1e3, 1))
_lib.TIMERS.reset() if hasattr(_lib.TIMERS, "reset") else None
for _ in range(10):
    y = conv.conv_transpose2d(xd, wd, bd, stride=2, padding=1, output_padding=0, pre_act=True, act=2)
    y.backward(torch.ones_like(y))
torch.cuda.synchronize()
for name, rec in _lib.TIMERS.summary().items():
    if "weight_grad" in name: print("deconv(+conv avg)", name, round(rec["avg_ms"] * 1e3, 1))
