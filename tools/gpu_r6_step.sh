#!/bin/bash
# bench line + detail, then a rocprofv3 kernel trace of the same command -> one steady-state step
set -u
ulimit -c 0
TAG=${1:-r6k}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 --detail $OUT/bench_detail.json ${BENCH_FLAGS:-} > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","value_full_length_text","value_no_pad_objects")}, d["roofline"]["frac"], d.get("headline"))
PY
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench --output-format csv -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras ${BENCH_FLAGS:-} > $OUT/prof_bench.log 2>&1; echo "rocprof exit $?")
find /tmp/prof -name '*kernel_stats*.csv' -exec cp {} $OUT/bench_kernel_stats.csv \;
find /tmp/prof -name '*kernel_trace*.csv' -exec cp {} /tmp/bench_kernel_trace.csv \;
python tools/step_from_trace.py /tmp/bench_kernel_trace.csv --json $OUT/step_from_trace.json > $OUT/step_from_trace.txt 2>&1; head -70 $OUT/step_from_trace.txt | cut -c1-180
