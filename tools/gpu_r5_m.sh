#!/bin/bash
set -u
ulimit -c 0
TAG=${1:-r5_m}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sa_fused.py tests/test_gpu_point_ops.py -m gpu -q -x > $OUT/pytest_a.log 2>&1; echo "exit $?" >> $OUT/pytest_a.log
grep -E "^(FAILED|ERROR)|passed|failed|exit|assert|Error" $OUT/pytest_a.log | head -20
timeout 1200 python -m pytest tests/test_gpu_model.py -m gpu -q -x > $OUT/pytest_b.log 2>&1; echo "exit $?" >> $OUT/pytest_b.log
grep -E "^(FAILED|ERROR)|passed|failed|exit" $OUT/pytest_b.log | head
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
python -c "import json;d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]);print('bench',d['value'],d['ms_per_step'])"; tail -2 $OUT/bench.err
