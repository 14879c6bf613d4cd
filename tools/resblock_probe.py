"""Fused residual-block forward against float64, per frame (GPU box): `python tools/resblock_probe.py`."""
import torch, torch.nn.functional as F, sys
sys.path.insert(0, ".")
from multimodal_mtrssm_amd import conv, _lib
dev = "cuda:0"
gen = torch.Generator().manual_seed(3)
def rnd(*s, scale=1.0): return (torch.randn(*s, generator=gen) * scale).to(dev)
def params(): return (rnd(128, 64, 3, 3, scale=0.05), rnd(128, scale=0.1), rnd(64, 128, 1, 1, scale=0.1), rnd(64, scale=0.1))
def ref(x, p):
    x, p = x.double().cpu(), [t.double().cpu() for t in p]
    return x + F.conv2d(F.elu(F.conv2d(F.elu(x), p[0], p[1], 1, 1)), p[2], p[3])
for na, nv in ((1, 0), (2, 0), (37, 0), (300, 0), (2, 3), (37, 50)):
    xa, pa = rnd(na, 64, 8, 8), params()
    xv, pv = (rnd(nv, 64, 8, 8), params()) if nv else (None, None)
    for fused in (False, True):
        conv.RESBLOCK_FUSE = fused
        conv.invalidate_packs()
        with torch.no_grad():
            if nv:
                ya, yv = conv.residual_block_pair(xa, pa, xv, pv, act=2)
            else:
                ya, yv = conv.residual_block(xa, *pa, act=2), None
        torch.cuda.synchronize()
        ea = (ya.double().cpu() - ref(xa, pa)).abs().amax(dim=(1, 2, 3))
        msg = f"na={na} nv={nv} fused={fused} err_a max={float(ea.max()):.2e} bad frames={[i for i, e in enumerate(ea.tolist()) if e > 1e-3][:10]}"
        if nv:
            ev = (yv.double().cpu() - ref(xv, pv)).abs().amax(dim=(1, 2, 3))
            msg += f" err_v max={float(ev.max()):.2e} bad={[i for i, e in enumerate(ev.tolist()) if e > 1e-3][:10]}"
        print(msg, flush=True)
