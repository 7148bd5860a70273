#!/bin/bash
# glue pass: the tests of the touched paths, bench line + one-step trace (gpu_r6_step.sh), eager attribution of the torch launches
set -u
ulimit -c 0
TAG=${1:-r6glue}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_joint_compact.py tests/test_gpu_losses.py tests/test_gpu_model.py tests/test_gpu_bert_varlen.py -m gpu -x -q 2>&1 | tail -5
bash tools/gpu_r6_step.sh $TAG
timeout 600 python tools/step_attrib.py --out $OUT/step_attrib.txt > /dev/null 2> $OUT/step_attrib.err; echo "attrib exit $?"
