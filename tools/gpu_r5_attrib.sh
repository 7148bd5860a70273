#!/bin/bash
set -u
TAG=${1:-r5_attrib}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/step_attrib.py --steps 2 --out $OUT/step_attrib.txt > $OUT/step_attrib.log 2>&1; echo "exit $?"; tail -3 $OUT/step_attrib.log
grep -n "total device time" $OUT/step_attrib.txt
