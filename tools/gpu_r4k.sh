#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r4k; mkdir -p $OUT
export TMPDIR=/tmp
P=tools/probes/post_addend_corruption_probe.py
run() { tag=$1; shift; timeout 300 python $P $OUT/$tag.json --save-grads /tmp/$tag.pt --taps $OUT/$tag.taps.json "$@" > $OUT/$tag.log 2>&1; echo "== $tag: rc $? $(grep -E 'non-finite|loss' $OUT/$tag.log | tr '\n' ' ')"; }
run eager_sp --graph off --post spatial --no-wgrad-group
run dp_sp_classic --graph dp --post spatial --no-wgrad-group
run dp_sp_classic_split --graph dp --post spatial --no-wgrad-group --split-bottom
echo "-- taps eager_sp vs dp_sp_classic"; python tools/probes/tap_diff.py $OUT/eager_sp.taps.json $OUT/dp_sp_classic.taps.json | grep -n "" | grep "<<<<" | head -40
echo "-- taps eager_sp vs dp_sp_classic_split"; python tools/probes/tap_diff.py $OUT/eager_sp.taps.json $OUT/dp_sp_classic_split.taps.json | grep -n "" | grep "<<<<" | head -10
