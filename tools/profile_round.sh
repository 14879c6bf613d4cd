#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh [stats]  -> gpurun_out/final/{stats,pmc_fetch,pmc_write}/ + bench_*.json
# The passes behind profiles/roundN_*: kernel trace + stats of the default bench command, the two HBM counter passes
# (separately, as MI355X_MICROARCH.md prescribes), then one plain bench line per mode.
set -o pipefail
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $ROOT && export PYTHONPATH=$ROOT
OUT=gpurun_out/final
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench --output-format csv -- python3 bench.py --steps 25 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.log || exit 1
echo stats done
for MODEL in mmtrssm large; do
  rocprofv3 --kernel-trace --stats -d $OUT/stats_$MODEL -o bench --output-format csv -- python3 bench.py --model $MODEL --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_${MODEL}_under_rocprof.json 2> $OUT/stats_$MODEL.log || exit 1
  echo stats $MODEL done
done
[ "$1" = "stats" ] && { find $OUT -name "*kernel_trace.csv" -delete; ls $OUT/*; exit 0; }
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o w --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1 || exit 1
echo pmc done
python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_summary.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline" > /dev/null || exit 1
for MODEL in mmtrssm large; do
  rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_$MODEL -o f --output-format csv -- python3 bench.py --model $MODEL --steps 2 --warmup 1 --no-cpu-baseline --no-elbo-check > $OUT/pmc_fetch_$MODEL.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_$MODEL -o w --output-format csv -- python3 bench.py --model $MODEL --steps 2 --warmup 1 --no-cpu-baseline --no-elbo-check > $OUT/pmc_write_$MODEL.log 2>&1 || exit 1
  python3 tools/pmc_traffic.py $OUT/pmc_fetch_$MODEL $OUT/pmc_write_$MODEL $OUT/${MODEL}_pmc_summary.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --model $MODEL --steps 2 --warmup 1 --no-cpu-baseline --no-elbo-check" > /dev/null || exit 1
  echo pmc $MODEL done
done
timeout -k 10 300 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
echo default done
timeout -k 10 200 python3 bench.py --no-cpu-baseline --conv-mfma bf16x3 > $OUT/bench_bf16x3.json 2>/dev/null || echo "FAILED: $_"
timeout -k 10 200 python3 bench.py --no-cpu-baseline --conv-mfma bf16 > $OUT/bench_bf16.json 2>/dev/null || echo "FAILED: $_"
timeout -k 10 200 python3 bench.py --no-cpu-baseline --conv-mfma f32 > $OUT/bench_f32.json 2>/dev/null || echo "FAILED: $_"
timeout -k 10 200 python3 bench.py --no-cpu-baseline --model mmtrssm > $OUT/bench_mmtrssm.json 2>/dev/null || echo "FAILED: $_"
timeout -k 10 200 python3 bench.py --no-cpu-baseline --force-dist > $OUT/bench_force_dist.json 2>/dev/null || echo "FAILED: $_"
timeout -k 10 200 python3 bench.py --no-cpu-baseline --graph on > $OUT/bench_graph.json 2>/dev/null || echo "FAILED: $_"
timeout -k 10 400 python3 bench.py --model large --steps 5 --warmup 2 > $OUT/bench_large.json 2>/dev/null || echo "FAILED: $_"
MTRSSM_SCAN_CLUSTER=0 timeout -k 10 200 python3 bench.py --no-cpu-baseline > $OUT/bench_single_cu_scan.json 2>/dev/null || echo "FAILED single-cu"
timeout -k 10 200 python3 bench.py --gpus 2 --backend gloo --share-device --steps 6 --warmup 2 --no-elbo-check > $OUT/bench_dp2_gloo_shared.json 2>/dev/null || echo "FAILED: $_"
echo modes done
find $OUT -name "*.csv" -size +2M -delete
ls $OUT $OUT/stats/*
