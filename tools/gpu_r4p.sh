#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r4p; mkdir -p $OUT
export TMPDIR=/tmp
P=$PWD/tools/probes/post_addend_corruption_probe.py
mkdir -p /tmp/dots && cd /tmp/dots
DEBUG_HIP_GRAPH_DOT_PRINT=1 timeout 300 python $P $OUT/dot.json --graph dp --post spatial --no-wgrad-group > $OUT/dot.log 2>&1; echo "rc $?"
for f in /tmp/dots/graph_*; do python /root/repo/tools/probes/dot_shape.py $f; cp $f $OUT/$(basename $f).dot; done 2>&1 | head -20
cd /root/repo
