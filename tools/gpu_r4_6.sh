#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r4_6; mkdir -p $OUT
export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_gemm.py -q -x -k "segmented or hip_graph or grouped or bench_n2 or fresh_dropout" > $OUT/pytest.log 2>&1; grep -E "passed|failed|Error|assert" $OUT/pytest.log | tail -8
for i in 1 2; do
echo "== one graph"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
echo "== graph-dp"; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --graph-dp 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
