"""Compare two tap-statistics files of post_addend_corruption_probe.py --taps (execution order)."""
import json
import sys
a, b = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
bm = {r["name"]: r for r in b}
for r in a:
    q = bm.get(r["name"])
    if q is None:
        print("missing in b:", r["name"])
        continue
    rel = abs(r["norm"] - q["norm"]) / (abs(r["norm"]) + 1e-30)
    flag = "  <<<<" if (rel > 1e-2 or r["nonfinite"] != q["nonfinite"] or r["norm"] != r["norm"] or q["norm"] != q["norm"]) else ""
    print(f"{r['name']:30s} {str(r['shape']):16s} rows {str(r['rows']):6s} norm {r['norm']:.5e} vs {q['norm']:.5e}  nonfinite {r['nonfinite']} vs {q['nonfinite']}{flag}")
