#!/bin/bash
# usage (here, after tools/profile_round.sh ran on the GPU box): tools/collect_profiles.sh roundN  -> profiles/roundN_*
R=${1:?round tag}; F=gpurun_out/final; P=profiles
last() { grep '^{' "$1" | tail -1; }
cp $F/stats/bench_kernel_stats.csv $P/${R}_bench_kernel_stats.csv
for m in mmtrssm large; do
  cp $F/stats_$m/bench_kernel_stats.csv $P/${R}_${m}_bench_kernel_stats.csv
  cp $F/${m}_pmc_summary.json $P/${R}_${m}_pmc_summary.json
  last $F/bench_${m}_under_rocprof.json > $P/${R}_bench_line_${m}_under_rocprof.json
done
cp $F/pmc_summary.json $P/${R}_pmc_summary.json
last $F/bench_default.json > $P/${R}_bench_line.json
last $F/bench_under_rocprof.json > $P/${R}_bench_line_under_rocprof.json
for m in bf16 bf16x3 f32 force_dist graph large mmtrssm single_cu_scan dp2_gloo_shared; do last $F/bench_$m.json > $P/${R}_bench_line_$m.json; done
for f in default bf16x3 bf16 f32 mmtrssm force_dist graph large single_cu_scan dp2_gloo_shared; do
  python3 - "$F/bench_$f.json" "$f" <<'PY'
import json, sys
d = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
cb = d.get("cpu_baseline") or {}
print(f"{sys.argv[2]:18s} {d['value']:10.0f} {d['ms_per_step']:8.3f} ms  cpu {cb.get('value')}  traffic {d['roofline'].get('traffic')}")
PY
done
