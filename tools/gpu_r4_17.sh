#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r4_17; mkdir -p $OUT
export TMPDIR=/tmp
echo "== varlen tests"; timeout 900 python -m pytest tests/test_gpu_bert_varlen.py -q -x > $OUT/pytest_varlen.log 2>&1; grep -E "passed|failed|Error" $OUT/pytest_varlen.log | tail -3
rm -rf /tmp/prof && mkdir -p /tmp/prof
REPO=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench --output-format csv -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/prof_bench.log 2>&1; echo "rocprof exit $?")
find /tmp/prof -name '*kernel_stats*.csv' -exec cp {} $OUT/bench_kernel_stats.csv \;
tail -1 $OUT/prof_bench.log | cut -c1-200
grep "varlen_plan\|sum_kernel" $OUT/bench_kernel_stats.csv | cut -c1-160
