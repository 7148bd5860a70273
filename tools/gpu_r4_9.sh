#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r4_9; mkdir -p $OUT
export TMPDIR=/tmp
echo "== bf16 bounds"; timeout 900 python -m pytest tests/test_gpu_model.py -q -x -s -k "bench_config_step" > $OUT/bf16_bounds.log 2>&1; grep -E "bf16-vs-fp32|passed|failed" $OUT/bf16_bounds.log
echo "== objects tests"; timeout 600 python -m pytest tests/test_gpu_objects.py -q -x 2>&1 | grep -E "passed|failed" | tail -2
echo "== bench"; timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; tail -c 900 $OUT/bench.json | cut -c1-900; python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('value','ms_per_step','value_full_length_text','value_with_device_sampler')})"; tail -2 $OUT/bench.err
