#!/bin/bash
# usage: tools/kernel_pmc.sh <kernel name substring> <python script> [args]   (on the GPU box; SQ / LDS / traffic counters of one kernel)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && export PYTHONPATH=$GRAFT_REPO_ROOT
K="$1"; shift
OUT=gpurun_out/kernel_pmc
rm -rf $OUT && mkdir -p $OUT
python3 "$@" > /dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/a -o a --output-format csv -- python3 "$@" > $OUT/a.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM -d $OUT/b -o b --output-format csv -- python3 "$@" > $OUT/b.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/c -o c --output-format csv -- python3 "$@" > $OUT/c.log 2>&1
KSUB="$K" python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/kernel_pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if os.environ["KSUB"] in row["Kernel_Name"]:
            acc[row["Kernel_Name"].split("(")[0][:80]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} n={len(v):3d} avg={sum(v)/len(v):16.1f}")
PY
