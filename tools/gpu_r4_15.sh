#!/bin/bash
# round 4: two-group 256 x 256 kernel for the K = 768 text shapes inside the step (A/B of the default threshold)
set -u
OUT=$PWD/gpurun_out/r4_15; mkdir -p $OUT
export TMPDIR=/tmp
echo "== varlen tests"; timeout 900 python -m pytest tests/test_gpu_bert_varlen.py -q -x > $OUT/pytest_varlen.log 2>&1; grep -E "passed|failed|Error" $OUT/pytest_varlen.log | tail -3
for mode in 1536 768 1536 768; do
  GPS_GEMM_8P_MIN_K=$mode timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --detail $OUT/detail_$mode.json 2>/dev/null | tail -1 > $OUT/bench_$mode.json
  python - <<P
import json
d=json.load(open('$OUT/bench_$mode.json')); dd=json.load(open('$OUT/detail_$mode.json'))
print('min_k $mode', d['value'], d['ms_per_step'], d['headline']['native_gemms_ms_per_step'], d['headline']['native_gemms_frac_bf16_mfma'])
for f in dd['kernel_families']:
    if f['kernel'].startswith('gemm') : print('   ', f['kernel'], f['ms_per_step'], f.get('frac'))
P
done
