"""Phase timing of the wide scans (profiling helper): s_memrealtime stamps (100 MHz) of EVERY workgroup at the phase boundaries
of timesteps 8..11 (csrc/mrssm_wide.hip: MTRSSM_WIDE_STAMP, csrc/mmtrssm_wide.hip: MTRSSM_MMT_STAMP).
Usage on the GPU box: python tools/wide_probe.py [fwd|bwd|mmt-fwd|mmt-bwd]

Per phase it prints, over the workgroups that had work in it, the median / maximum span of the work itself and, over all
workgroups, the span of the barrier that follows (arrival of the LAST workgroup -> everybody released)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from multimodal_mtrssm_amd import _lib

which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
mmt = which.startswith("mmt-")
direction = which.split("-")[-1]
dev = "cuda:0"
if mmt:
    model = bench.build_model(dev, "mmtrssm")
    batch = bench.synthetic_batch(64, dev, 1)
    nrows = 64
else:
    bench.WORKLOAD = bench.WORKLOADS["large"]
    model = bench.build_model(dev, "large")
    batch = bench.synthetic_batch(32, dev, 1)
    nrows = 32
lib = _lib.load()
nblk = torch.cuda.get_device_properties(0).multi_processor_count
buf = torch.zeros(nblk * 4 * 16, dtype=torch.int64, device=dev)
fn = lib.mtrssm_debug_set_mmt_profile if mmt else lib.mtrssm_debug_set_wide_profile
fn.argtypes, fn.restype = [ctypes.c_void_p], ctypes.c_int
names = {"fwd": ["A row: cat + h1", "B gru", "C heads0", "D logits"], "bwd": ["R0 row: cat bwd", "R1 dzh", "R2 gates", "R3 carry/dz1", "R4 carry_s"],
         "mmt-fwd": ["F0 row: cat", "F1 cells", "F2 heads0", "F3 logits"], "mmt-bwd": ["R0 row: cat bwd", "R1 dz", "R2 du", "R3 carries"]}[which]
for it in range(3):
    if direction == "fwd":
        assert fn(buf.data_ptr()) == 0
        with torch.no_grad():
            model.shared_step(batch, None)
        torch.cuda.synchronize()
        assert fn(None) == 0
    else:
        out = model.shared_step(batch, None)
        torch.cuda.synchronize()
        buf.zero_()
        assert fn(buf.data_ptr()) == 0
        out["loss"].backward()
        torch.cuda.synchronize()
        assert fn(None) == 0
    st = buf.cpu().view(nblk, 4, 16).double() * 0.01  # microseconds
    n = len(names)
    for step in (1, 2):
        s = st[:, step]
        t0 = s[:, 0].min()
        print(f"iter {it} step {8 + step}: whole step {float(s[:, 2 * n].max() - t0):7.2f} us")
        for p, name in enumerate(names):
            work = s[:, 2 * p + 1] - s[:, 2 * p]          # phase start -> own arrival at the barrier
            busy = work[work > 0.3]
            last_arrival = s[:, 2 * p + 1].max()
            release = s[:, 2 * p + 2]
            print(f"   {name:18s} work: {busy.numel():3d} wgs median {float(busy.median()) if busy.numel() else 0:6.2f} max {float(work.max()):6.2f} us"
                  f" | phase start spread {float(s[:, 2 * p].max() - s[:, 2 * p].min()):5.2f}"
                  f" | barrier: last arrival -> release median {float((release - last_arrival).median()):5.2f} max {float((release - last_arrival).max()):5.2f} us"
                  f" | phase wall {float(release.max() - s[:, 2 * p].min()):6.2f}")
        rows = s[-nrows:]  # the row workgroups are the last ones of the grid
        d = lambda a, b: float((rows[:, b] - rows[:, a]).median())
        if which == "fwd":
            print(f"      row workgroups (median): logits -> LDS {d(0, 10):5.2f}, mix {d(10, 11):5.2f}, cat block + outputs {d(11, 12):5.2f}, gather + h1 {d(12, 1):5.2f} us")
        elif which == "bwd":
            print(f"      row workgroups (median): staging {d(0, 11):5.2f}, cat bwd {d(11, 12):5.2f}, mix bwd {d(12, 13):5.2f}, outputs + exchange {d(13, 1):5.2f} us")
        else:
            print(f"      row workgroups (median): staging {d(0, 10):5.2f}, categorical work (wave 0) {d(10, 11):5.2f}, outputs + exchange {d(11, 1):5.2f} us")
