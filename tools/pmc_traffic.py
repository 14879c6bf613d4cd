"""Builds profiles/<tag>_pmc_summary.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py.

usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> "<command string>"
Kernel names are normalised to what mtrssm_last_kernel() reports (no "void ", no argument list)."""
import csv, glob, json, sys
from collections import defaultdict

def collect(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            if "mtrssm::" not in name:
                continue
            name = name[name.index("mtrssm::"):].split("(")[0]
            acc[name].append(float(row["Counter_Value"]))
    return acc

fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"command": sys.argv[4],
       "note": "raw counter values per launch (KB), separate passes. MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads "
               "by 2x on gfx950; bench.py prices traffic = 2*FETCH+WRITE (an upper estimate for 4 B/lane streams)",
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, []), write.get(k, [])
    out["kernels"][k] = {"launches": max(len(f), len(w)), "FETCH_SIZE_KB_avg": sum(f) / max(1, len(f)), "WRITE_SIZE_KB_avg": sum(w) / max(1, len(w))}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in list(out["kernels"].items())[:4]}, indent=1))
