#!/bin/bash
# usage: tools/pmc_probe.sh <probe args...>   (on the GPU box; writes gpurun_out/pmcA, pmcB and prints the summary)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && export PYTHONPATH=$GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcA gpurun_out/pmcB
python3 tools/conv_probe.py "$@" | grep -v pack
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d gpurun_out/pmcA -o a --output-format csv -- python3 tools/conv_probe.py "$@" > gpurun_out/pmcA.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM -d gpurun_out/pmcB -o b --output-format csv -- python3 tools/conv_probe.py "$@" > gpurun_out/pmcB.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmcA gpurun_out/pmcB
