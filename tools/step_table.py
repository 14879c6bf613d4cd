"""The "Where the step goes" table of DESIGN.md from profiles/<round>_*bench_kernel_stats.csv: `python tools/step_table.py round3`."""
import csv
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "round3"
SCAN = "scan (forward + backward, cluster kernels)"
GEMM = "GEMMs (all Linear layers and their gradients)"
TORCH = "torch elementwise / copies"


def group(k: str) -> str:
    if "cluster" in k or "wide_fwd" in k or "wide_bwd" in k or "wide_pack" in k:
        return SCAN
    if "gemm" in k:
        return GEMM
    if "conv3x3_resident" in k:
        return "weight-resident 3x3 gathers (forward: the whole residual block, fused)"
    if "conv1x1_stream" in k:
        return "1x1 gathers (streaming kernel; backward-data only)"
    if "wgrad_reduce" in k:
        return "partial-set sums of the staged weight gradients (one batched launch)"
    if "wgrad" in k:
        return "conv weight gradients (staged kernels)"
    if "quad" in k or "rows" in k:
        return "all-parity-class transposed kernels (ConvTranspose forward, stride-2 conv backward-data)"
    if "thin" in k or "gather_split" in k or "band" in k:
        return "stride-2 / thin gathers (band, split, VALU kernels)"
    if "mtrssm::" in k:
        return "other library kernels (NLL, AdamW, sums, pack, unpack, clear, feed, categorical head, scalar epilogue)"
    return TORCH


def table(path: str):
    rows = list(csv.DictReader(open(path)))
    steps = int([r for r in rows if "adamw_masked" in r["Name"]][0]["Calls"])
    groups: dict[str, list[float]] = {}
    tot = nl = 0.0
    for r in rows:
        g, ms, n = group(r["Name"]), float(r["TotalDurationNs"]) / steps / 1e6, int(r["Calls"]) / steps
        groups.setdefault(g, [0.0, 0.0])
        groups[g][0] += ms
        groups[g][1] += n
        tot += ms
        nl += n
    lines = [f"| {g} | {n:.0f} | {ms:.2f} | {100 * ms / tot:.1f} % |" for g, (ms, n) in sorted(groups.items(), key=lambda kv: -kv[1][0])]
    return lines, groups, tot, nl


lines, _, tot, nl = table(f"profiles/{R}_bench_kernel_stats.csv")
_, gm, totm, nlm = table(f"profiles/{R}_mmtrssm_bench_kernel_stats.csv")
_, gl, totl, _ = table(f"profiles/{R}_large_bench_kernel_stats.csv")
print("| group | launches / step | ms / step | share |\n|---|---|---|---|")
print("\n".join(lines))
print(f"\nbase: {nl:.0f} launches, {tot:.2f} ms; mmtrssm: {totm:.2f} ms in {nlm:.0f} launches, scan {gm[SCAN][0]:.2f}, torch {gm[TORCH][0]:.2f}; "
      f"large: {totl:.2f} ms, scan {gl[SCAN][0]:.1f}, GEMMs {gl[GEMM][0]:.1f}")
