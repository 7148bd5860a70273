#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r4n; mkdir -p $OUT
export TMPDIR=/tmp
P=tools/probes/post_addend_corruption_probe.py
timeout 300 python $P $OUT/a.json --graph dp --post spatial --no-wgrad-group --alias-scan --dump-graphs $OUT --taps $OUT/a.taps.json > $OUT/a.log 2>&1; echo "rc $?"
grep -E "non-finite|alias scan:|neighbours" -A0 $OUT/a.log | head; grep -A40 "neighbours of" $OUT/a.log | head -60
python - <<'PY'
import json
st=json.load(open("gpurun_out/r4n/a.taps.json"))
for r in st:
    if r.get("where"): print(r["name"], r["shape"], "rows", r["rows"], "nonfinite", r["nonfinite"], r["where"])
PY
ls -la $OUT/*.dot 2>/dev/null; for f in $OUT/*.dot; do python tools/probes/dot_shape.py $f; done
head -c 1500 $OUT/g2b_0.dot
