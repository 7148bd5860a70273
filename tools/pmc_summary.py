"""Per-kernel averages of a rocprofv3 --pmc counter_collection CSV (one row per counter per dispatch)."""
import csv
import sys
from collections import defaultdict

MIN_DISPATCHES = int(sys.argv[2]) if len(sys.argv) > 2 else 5      # kernels launched fewer times are left out
acc = defaultdict(lambda: defaultdict(list))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name", "")
        if len(name) > 70:
            name = name[:70]
        acc[name][r.get("Counter_Name", "")].append(float(r.get("Counter_Value", 0) or 0))
for name, cs in acc.items():
    n = max(len(v) for v in cs.values())
    if n < MIN_DISPATCHES:
        continue
    print(f"{name}  ({n} dispatches)")
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {sum(v) / len(v):16.1f}")
