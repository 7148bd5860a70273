#!/bin/bash
# round 6: full GPU test suite, then the glue attribution of one eager step
set -u
ulimit -c 0
TAG=${1:-r6_full}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu_full.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu_full.log
grep -E "^(FAILED|ERROR)|passed|failed|pytest exit" $OUT/pytest_gpu_full.log | head -20
timeout 600 python tools/step_attrib.py --steps 2 --out $OUT/step_attrib.txt > $OUT/step_attrib.log 2>&1; head -80 $OUT/step_attrib.txt | cut -c1-230
