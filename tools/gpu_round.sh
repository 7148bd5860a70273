#!/bin/bash
# One gpurun call: GPU parity tests, bench line, rocprofv3 kernel stats of the same bench command.
# Outputs under gpurun_out/<tag>/ (merged back by gpurun, <= 64 MiB in total: raw traces stay in /tmp).
set -u
TAG=${1:-r1}
WHAT=${2:-all}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ts() { echo "[$(date +%H:%M:%S)] $*"; }
{ echo "nproc $(nproc)"; free -g | head -2; rocm-smi --showproductname 2>&1 | grep -i -m2 "card series\|gfx"; } > $OUT/host.txt 2>&1
cat $OUT/host.txt
if [[ $WHAT == all || $WHAT == *test* ]]; then
  ts pytest; timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
  grep -E "^(FAILED|ERROR)|AssertionError|passed|failed|exit" $OUT/pytest_gpu.log | head -30
fi
if [[ $WHAT == all || $WHAT == *bench* ]]; then
  ts bench; timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
  tail -c 4000 $OUT/bench.json; tail -5 $OUT/bench.err
fi
if [[ $WHAT == all || $WHAT == *kbench* ]]; then
  ts kernel_bench; timeout 300 python tools/kernel_bench.py --json $OUT/kernel_bench.json > $OUT/kernel_bench.log 2>&1; tail -3 $OUT/kernel_bench.log
fi
if [[ $WHAT == all || $WHAT == *prof* ]]; then
  ts rocprof
  rm -rf /tmp/prof && mkdir -p /tmp/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench --output-format csv -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.log 2>&1; echo "rocprof exit $?")
  mkdir -p $OUT/prof
  find /tmp/prof -name '*stats*.csv' -exec cp {} $OUT/prof/ \;
  ls -la /tmp/prof/* | head; du -sh /tmp/prof
  f=$(ls $OUT/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -45 "$f"
  tail -3 $OUT/prof_bench.log
fi
if [[ $WHAT == *pmc* ]]; then
  ts pmc
  rm -rf /tmp/pmc && mkdir -p /tmp/pmc
  (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc -o fetch --output-format csv -- python $REPO/tools/pmc_workload.py > $OUT/pmc_fetch.log 2>&1; echo "pmc fetch exit $?")
  (cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc -o write --output-format csv -- python $REPO/tools/pmc_workload.py > $OUT/pmc_write.log 2>&1; echo "pmc write exit $?")
  ls -la /tmp/pmc | head
  mkdir -p $OUT/pmc
  cp /tmp/pmc/*counter_collection.csv $OUT/pmc/ 2>/dev/null
  python tools/pmc_traffic.py /tmp/pmc/fetch_counter_collection.csv /tmp/pmc/write_counter_collection.csv $OUT/pmc/pmc_traffic.json | tail -60
fi
ts done; du -sh $REPO/gpurun_out
