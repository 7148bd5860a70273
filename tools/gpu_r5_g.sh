#!/bin/bash
set -u
TAG=${1:-r5_g}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ts() { echo "[$(date +%H:%M:%S)] $*"; }
ts tests
timeout 1200 python -m pytest tests/test_gpu_losses.py tests/test_gpu_bert_varlen.py tests/test_gpu_gemm.py -m gpu -q > $OUT/pytest_a.log 2>&1; echo "exit $?" >> $OUT/pytest_a.log; grep -E "^(FAILED|ERROR)|passed|failed|exit" $OUT/pytest_a.log | head -20
timeout 1200 python -m pytest tests/test_gpu_model.py -m gpu -q -k "not bench_n2 and not self_spawns" > $OUT/pytest_b.log 2>&1; echo "exit $?" >> $OUT/pytest_b.log; grep -E "^(FAILED|ERROR)|passed|failed|exit" $OUT/pytest_b.log | head -20
ts bench
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>$OUT/bench.err | tail -1 > $OUT/bench.json; python -c "import json;d=json.load(open('$OUT/bench.json'));print('bench',d['value'],d['ms_per_step'],d['config']['final_loss'])" || tail -5 $OUT/bench.err
ts done
