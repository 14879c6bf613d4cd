#!/bin/bash
# usage: tools/gemm_pmc.sh M N R a_rm b_rm pieces   (on the GPU box; SQ / LDS / traffic counters of one GEMM shape)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && export PYTHONPATH=$GRAFT_REPO_ROOT
OUT=gpurun_out/gemm_pmc
rm -rf $OUT && mkdir -p $OUT
python3 tools/gemm_probe.py "$@"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/a -o a --output-format csv -- python3 tools/gemm_probe.py "$@" > $OUT/a.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM -d $OUT/b -o b --output-format csv -- python3 tools/gemm_probe.py "$@" > $OUT/b.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE -d $OUT/c -o c --output-format csv -- python3 tools/gemm_probe.py "$@" > $OUT/c.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $OUT/d -o d --output-format csv -- python3 tools/gemm_probe.py "$@" > $OUT/d.log 2>&1
python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/gemm_pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "gemm" in row["Kernel_Name"]:
            acc[row["Kernel_Name"].split("(")[0][:80]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} n={len(v):3d} avg={sum(v)/len(v):16.1f}")
PY
