#!/bin/bash
# round 4: v_rcp_f32 instead of IEEE divisions in the GELU / GELU' epilogues and the spatial attention term
set -u
OUT=$PWD/gpurun_out/r4_16; mkdir -p $OUT
export TMPDIR=/tmp
echo "== tests"; timeout 1200 python -m pytest tests/test_gpu_attention.py tests/test_gpu_attention_ex.py tests/test_gpu_gemm.py tests/test_gpu_model.py tests/test_a16_vs_golden.py -q -x > $OUT/pytest.log 2>&1; grep -E "passed|failed|Error" $OUT/pytest.log | tail -3
for i in 1 2; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --detail $OUT/detail_$i.json 2>/dev/null | tail -1 > $OUT/bench_$i.json
  python - <<P
import json
d=json.load(open('$OUT/bench_$i.json')); dd=json.load(open('$OUT/detail_$i.json'))
print('run $i', d['value'], d['ms_per_step'], d['headline'])
for f in dd['kernel_families']:
    print('   ', f['kernel'], f['ms_per_step'], f.get('frac'))
for k in dd['kernels']:
    if k['kernel'].startswith('attn') and 'spatial=1' in k['kernel']: print('   ', k['kernel'], k['avg_us'])
P
done
