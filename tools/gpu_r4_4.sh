#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r4_4; mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2; do
for rs in 1 0; do
  echo "== rowsplit=$rs run $i"; GPS_GEMM_ROWSPLIT=$rs timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
echo "== point+emb tests"; timeout 600 python -m pytest tests/test_gpu_embedding.py tests/test_gpu_optim.py -q -x 2>&1 | grep -E "passed|failed" | tail -2
