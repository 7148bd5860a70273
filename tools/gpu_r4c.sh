#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r4c; mkdir -p $OUT
export TMPDIR=/tmp
P=tools/probes/post_addend_corruption_probe.py
run() { tag=$1; shift; timeout 300 python $P $OUT/$tag.json --save-grads /tmp/$tag.pt "$@" > $OUT/$tag.log 2>&1; echo "== $tag: $(grep -E 'non-finite|CHANGED|loss' $OUT/$tag.log | tr '\n' ' ')"; }
run eager_off --graph off --post off
run dp_sp_split --graph dp --post spatial --split-bottom
run dp_sp_novarlen --graph dp --post spatial --no-varlen
run dp_sp_noclstail --graph dp --post spatial --no-cls-tail
for t in dp_sp_split dp_sp_novarlen dp_sp_noclstail; do
  echo "-- diff eager_off vs $t"; timeout 120 python tools/probes/dp_graph_grad_diff_probe.py diff /tmp/eager_off.pt /tmp/$t.pt 2>&1 | tail -3
done
tail -60 $OUT/dp_sp_split.log
