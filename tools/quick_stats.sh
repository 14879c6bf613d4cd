#!/bin/bash
# usage (on the GPU box): tools/quick_stats.sh [pattern]  -> kernel stats of the default bench command, rows matching pattern
set -o pipefail
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $ROOT && export PYTHONPATH=$ROOT
OUT=gpurun_out/quick
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench --output-format csv -- python3 bench.py --steps 25 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.log || exit 1
find $OUT -name "*kernel_trace.csv" -delete
python3 - "$OUT/stats/bench_kernel_stats.csv" "${1:-.}" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int([r for r in rows if "adamw_masked" in r["Name"]][0]["Calls"])
tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
print(f"steps {steps}  kernel ms/step {tot:.3f}  launches/step {sum(int(r['Calls']) for r in rows) / steps:.1f}")
for r in rows:
    if re.search(sys.argv[2], r["Name"]):
        print(f"{r['Name'][:120]:120s} {int(r['Calls']) / steps:5.1f}/step  avg {float(r['AverageNs']) / 1e3:7.1f} us  {float(r['TotalDurationNs']) / steps / 1e3:7.1f} us/step")
PY
