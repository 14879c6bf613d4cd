"""Print the headline fields of a bench.py JSON line (helper for GPU-box shell one-liners): python tools/benchline.py FILE [N]"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
print(d["value"], d["ms_per_step"], d.get("ms_per_step_median"), d.get("elbo_rel_delta"))
r = d.get("roofline", {})
print(r.get("kernel"), r.get("frac"))
for k, v in list(r.get("kernels", {}).items())[:n]:
    print(k, v)
