#!/bin/bash
# usage (GPU box): tools/step_sequence.sh [bench args]  -> gpurun_out/seq/sequence.txt: the kernels of the LAST profiled step in launch order
set -o pipefail
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $ROOT && export PYTHONPATH=$ROOT
OUT=gpurun_out/seq
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT/t -o k --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-elbo-check "$@" > $OUT/bench.json 2> $OUT/log.txt || exit 1
python3 - <<'PY'
import csv, glob, re
f = glob.glob("gpurun_out/seq/t/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(.*", "", r["Kernel_Name"]) for r in rows]
# the last occurrence of the optimizer kernel ends a step; the one before it ends the previous step
idx = [i for i, n in enumerate(names) if "adamw_masked" in n]
a, b = idx[-3] + 1, idx[-2] + 1
with open("gpurun_out/seq/sequence.txt", "w") as o:
    for r, n in zip(rows[a:b], names[a:b]):
        o.write(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  {n[:150]}\n")
print(b - a, "kernels in the step")
PY
find $OUT/t -name "*.csv" -size +1M -delete
