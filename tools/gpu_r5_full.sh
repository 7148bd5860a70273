#!/bin/bash
# full GPU test suite + bench line (+ detail) + rocprofv3 kernel stats of the bench command.  Outputs under gpurun_out/<tag>/.
set -u
TAG=${1:-r5_full}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ts() { echo "[$(date +%H:%M:%S)] $*"; }
{ echo "nproc $(nproc)"; free -g | head -2; rocm-smi --showproductname 2>&1 | grep -i -m2 "card series\|gfx"; } > $OUT/host.txt 2>&1
ts pytest; timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|exit" $OUT/pytest_gpu.log | head -30
ts bench; timeout 900 python bench.py --steps 10 --warmup 3 --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
tail -c 3000 $OUT/bench.json; tail -3 $OUT/bench.err
ts rocprof
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench --output-format csv -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --detail $OUT/bench_prof_detail.json > $OUT/prof_bench.log 2>&1; echo "rocprof exit $?")
find /tmp/prof -name '*kernel_stats*.csv' -exec cp {} $OUT/bench_kernel_stats.csv \;
head -30 $OUT/bench_kernel_stats.csv | cut -c1-150
ts done
