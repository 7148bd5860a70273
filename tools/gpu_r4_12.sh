#!/bin/bash
# round 4: XCD-local tile queues of the grouped weight-gradient launch, A/B against the single global queue
set -u
OUT=$PWD/gpurun_out/r4_12; mkdir -p $OUT
export TMPDIR=/tmp
echo "== gemm tests"; timeout 900 python -m pytest tests/test_gpu_gemm.py -q -x -k "grouped" > $OUT/pytest_gemm.log 2>&1; grep -E "passed|failed|Error" $OUT/pytest_gemm.log | tail -3
for mode in xcd one xcd one; do
  flag=""; [ $mode = one ] && flag="--wgrad-one-queue"
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras $flag --detail $OUT/detail_$mode.json 2>/dev/null | tail -1 > $OUT/bench_$mode.json
  python - <<P
import json
d=json.load(open('$OUT/bench_$mode.json')); dd=json.load(open('$OUT/detail_$mode.json'))
g=[f for f in dd['kernel_families'] if f['kernel']=='gemm_tn_grouped']
print('$mode', d['value'], d['ms_per_step'], g[0]['ms_per_step'] if g else None, g[0].get('frac') if g else None)
P
done
echo "== pmc"
rm -rf /tmp/pmc && mkdir -p /tmp/pmc
REPO=$PWD
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc -o fetch --output-format csv -- python $REPO/tools/pmc_workload.py > $OUT/pmc_fetch.log 2>&1; echo "pmc fetch exit $?")
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc -o write --output-format csv -- python $REPO/tools/pmc_workload.py > $OUT/pmc_write.log 2>&1; echo "pmc write exit $?")
python tools/pmc_traffic.py /tmp/pmc/fetch_counter_collection.csv /tmp/pmc/write_counter_collection.csv $OUT/pmc_traffic.json > $OUT/pmc_traffic.log 2>&1; grep -A3 "gemm_tn_grouped" $OUT/pmc_traffic.json | head -12
