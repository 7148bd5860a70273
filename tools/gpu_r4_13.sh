#!/bin/bash
# round 4: bias-gradient column sums of the two-group TN kernel spread over the four wave columns
set -u
OUT=$PWD/gpurun_out/r4_13; mkdir -p $OUT
export TMPDIR=/tmp
echo "== gemm tests"; timeout 900 python -m pytest tests/test_gpu_gemm.py -q -x > $OUT/pytest_gemm.log 2>&1; grep -E "passed|failed|Error" $OUT/pytest_gemm.log | tail -3
for i in 1 2; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --detail $OUT/detail_$i.json 2>/dev/null | tail -1 > $OUT/bench_$i.json
  python - <<P
import json
d=json.load(open('$OUT/bench_$i.json')); dd=json.load(open('$OUT/detail_$i.json'))
g=[f for f in dd['kernel_families'] if f['kernel']=='gemm_tn_grouped']
print('run $i', d['value'], d['ms_per_step'], g[0]['ms_per_step'] if g else None, g[0].get('frac') if g else None)
P
done
