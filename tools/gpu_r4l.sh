#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r4l; mkdir -p $OUT
export TMPDIR=/tmp
P=tools/probes/post_addend_corruption_probe.py
run() { tag=$1; shift; timeout 300 python $P $OUT/$tag.json --save-grads /tmp/$tag.pt "$@" > $OUT/$tag.log 2>&1; echo "== $tag: rc $? $(grep -E 'non-finite|loss' $OUT/$tag.log | tr '\n' ' ')"; grep -A40 "watch:" $OUT/$tag.log | head -60; }
run dp_sp_classic_watch --graph dp --post spatial --no-wgrad-group --watch
