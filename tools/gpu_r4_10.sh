#!/bin/bash
# round 4: row-wise kernels with every load of a row issued up front (LayerNorm fwd/bwd, word-table gradient)
set -u
OUT=$PWD/gpurun_out/r4_10; mkdir -p $OUT
export TMPDIR=/tmp
echo "== unit tests"; timeout 900 python -m pytest tests/test_gpu_fused_norm.py tests/test_gpu_embedding.py tests/test_gpu_bert_varlen.py -q -x > $OUT/pytest_units.log 2>&1; grep -E "passed|failed|Error" $OUT/pytest_units.log | tail -3
echo "== bf16 bounds"; timeout 900 python -m pytest tests/test_gpu_model.py -q -x -s -k "bench_config_step or segmented" > $OUT/bf16_bounds.log 2>&1; grep -E "bf16-vs-fp32|passed|failed" $OUT/bf16_bounds.log
echo "== bench"; timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err; python - <<P
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('value','ms_per_step','value_full_length_text','value_with_device_sampler')})
dd=json.load(open('$OUT/bench_detail.json'))
for f in dd['kernel_families']:
    if any(s in f['kernel'] for s in ('layernorm','embedding','emb','adamw')): print(f['kernel'], f['ms_per_step'], f.get('frac'))
for k in dd['kernels']:
    if 'emb' in k['kernel'] or 'word' in k['kernel']: print(k['kernel'], k.get('avg_us'), k.get('frac'), k.get('frac_hbm'))
P
tail -2 $OUT/bench.err
