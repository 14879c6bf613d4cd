"""Average PMC counters per kernel from rocprofv3 counter_collection CSVs (profiling helper)."""
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    if "split" not in k and "patch" not in k and "thin" not in k: continue
    print(k)
    for c, v in sorted(cs.items()): print(f"    {c:32s} n={len(v):3d} avg={sum(v)/len(v):14.1f}")
