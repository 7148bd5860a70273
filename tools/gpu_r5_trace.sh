#!/bin/bash
# rocprofv3 kernel trace of the bench command -> one steady-state step (tools/step_from_trace.py)
set -u
ulimit -c 0
TAG=${1:-r5_trace}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench --output-format csv -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras ${BENCH_FLAGS:-} > $OUT/prof_bench.log 2>&1; echo "rocprof exit $?")
find /tmp/prof -name '*kernel_stats*.csv' -exec cp {} $OUT/bench_kernel_stats.csv \;
find /tmp/prof -name '*kernel_trace*.csv' -exec cp {} /tmp/bench_kernel_trace.csv \;
head -1 /tmp/bench_kernel_trace.csv > $OUT/kernel_trace_header.txt
python tools/step_from_trace.py /tmp/bench_kernel_trace.csv --json $OUT/step_from_trace.json > $OUT/step_from_trace.txt 2>&1; head -60 $OUT/step_from_trace.txt
tail -1 $OUT/prof_bench.log | cut -c1-200
