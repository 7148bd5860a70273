#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r4_20; mkdir -p $OUT
export TMPDIR=/tmp
timeout 500 python tools/step_attrib.py --out $OUT/step_attrib.txt > $OUT/step_attrib.log 2>&1; echo "attrib exit $?"; tail -3 $OUT/step_attrib.log
