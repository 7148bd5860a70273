#!/bin/bash
# One-off hipBLASLt/rocBLAS solution tuning for the GEMM shapes of the GPS step with PyTorch's
# TunableOp (run on the GPU box; copy the CSV into profiles/ -- bench.py loads it read-only).
set -u
OUT=${1:-gpurun_out/tunableop}
mkdir -p $OUT
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_VERBOSE=0
export PYTORCH_TUNABLEOP_FILENAME=$PWD/$OUT/tunableop_gfx950.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=${TUNE_MS:-15} PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=2
export GPS_NO_TUNABLEOP_FILE=1
date
timeout ${TUNE_TIMEOUT:-420} python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > $OUT/bench_tuning.json 2> $OUT/bench_tuning.err
echo "tuning run exit $?"; date
ls -la $OUT; wc -l $OUT/*.csv 2>/dev/null; head -5 $OUT/*.csv 2>/dev/null
