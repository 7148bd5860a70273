#!/bin/bash
# A/B of one environment switch under rocprofv3: per-kernel averages of the bench step for each value
#   bash tools/gpu_r6_ab.sh VAR "v1 v2 ..." "kernel-name regex" [pytest files...]
set -u
export TMPDIR=/tmp
VAR=$1; VALUES=$2; PAT=$3; shift 3
REPO=$PWD
OUT=$REPO/gpurun_out/r6_ab
mkdir -p $OUT
if [ $# -gt 0 ]; then python -m pytest "$@" -m gpu -x -q 2>&1 | tail -3; fi
for V in $VALUES; do
  rm -rf /tmp/prof && mkdir -p /tmp/prof
  (cd /tmp && env $VAR=$V timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench --output-format csv -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/prof_$V.log 2>&1; echo "rocprof exit $?")
  find /tmp/prof -name '*kernel_stats*.csv' -exec cp {} $OUT/stats_$V.csv \;
  python - <<PY
import csv, json, re
line = [l for l in open("$OUT/prof_$V.log") if l.startswith("{")][-1]
d = json.loads(line)
print("$VAR=$V", d["value"], d["ms_per_step"])
for r in csv.DictReader(open("$OUT/stats_$V.csv")):
    if re.search(r"$PAT", r["Name"]):
        print("   %-70s %4s calls  avg %7.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
