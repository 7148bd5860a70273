#!/bin/bash
# round 6, final measurement call: full GPU test suite (with the [bf16-vs-fp32] lines), bench line (+ detail), rocprofv3
# kernel stats of the same bench command, PMC traffic passes, the other presets, smoke.  Outputs under gpurun_out/<tag>/.
set -u
ulimit -c 0
TAG=${1:-r6_final}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ts() { echo "[$(date +%H:%M:%S)] $*"; }
{ echo "nproc $(nproc)"; free -g | head -2; rocm-smi --showproductname 2>&1 | grep -i -m2 "card series\|gfx"; } > $OUT/host.txt 2>&1
ts pytest; timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu_full.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu_full.log
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit" $OUT/pytest_gpu_full.log | head -20
grep "bf16-vs-fp32" $OUT/pytest_gpu_full.log > $OUT/bounds_default.log; wc -l $OUT/bounds_default.log
grep -E "^(FAILED|ERROR)|passed|failed|skipped|pytest exit|warnings summary" $OUT/pytest_gpu_full.log > $OUT/pytest_gpu.log
ts bench; timeout 900 python bench.py --steps 20 --warmup 5 --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
tail -c 1500 $OUT/bench.json; tail -3 $OUT/bench.err
ts rocprof
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench --output-format csv -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --detail $OUT/bench_prof_detail.json > $OUT/prof_bench.log 2>&1; echo "rocprof exit $?")
find /tmp/prof -name '*kernel_stats*.csv' -exec cp {} $OUT/bench_kernel_stats.csv \;
find /tmp/prof -name '*kernel_trace*.csv' -exec cp {} /tmp/bench_kernel_trace.csv \;
python tools/step_from_trace.py /tmp/bench_kernel_trace.csv --json $OUT/step_from_trace.json > $OUT/step_from_trace.txt 2>&1; head -22 $OUT/step_from_trace.txt
head -1 /tmp/bench_kernel_trace.csv > $OUT/kernel_trace_header.txt
tail -2 $OUT/prof_bench.log | cut -c1-300
ts pmc
rm -rf /tmp/pmc && mkdir -p /tmp/pmc
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc -o fetch --output-format csv -- python $REPO/tools/pmc_workload.py > $OUT/pmc_fetch.log 2>&1; echo "pmc fetch exit $?")
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc -o write --output-format csv -- python $REPO/tools/pmc_workload.py > $OUT/pmc_write.log 2>&1; echo "pmc write exit $?")
python tools/pmc_traffic.py /tmp/pmc/fetch_counter_collection.csv /tmp/pmc/write_counter_collection.csv $OUT/pmc_traffic.json > $OUT/pmc_traffic.log 2>&1; grep -A3 "gemm_tn_grouped" $OUT/pmc_traffic.json | head -8
ts presets
run() { n=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $OUT/bench_$n.json; python -c "import json;d=json.load(open('$OUT/bench_$n.json'));print('$n',d['value'],d['ms_per_step'])"; }
run sa_bf16 --steps 10 --warmup 3 --sa-bf16
run graph_dp --steps 10 --warmup 3 --graph-dp
run padded --steps 10 --warmup 3 --no-varlen
run finetune --config finetune --steps 5 --warmup 2
run stress --config stress --steps 5 --warmup 2
ts smoke; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
ts done; du -sh $REPO/gpurun_out
