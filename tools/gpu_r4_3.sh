#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r4_3; mkdir -p $OUT
export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_embedding.py tests/test_gpu_model.py -q -x > $OUT/pytest.log 2>&1; grep -E "passed|failed|Error|assert" $OUT/pytest.log | tail -8
echo "== bench"; timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; tail -c 1800 $OUT/bench.json; cp gpurun_out/bench_detail.json $OUT/bench_detail.json
