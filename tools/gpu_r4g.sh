#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r4g; mkdir -p $OUT
export TMPDIR=/tmp
P=tools/probes/post_addend_corruption_probe.py
run() { tag=$1; shift; timeout 300 python $P $OUT/$tag.json --save-grads /tmp/$tag.pt "$@" > $OUT/$tag.log 2>&1; echo "== $tag: rc $? $(grep -E 'non-finite|loss|stage' $OUT/$tag.log | tr '\n' ' ')"; }
run eager_off --graph off --post off
run objfirst_eager_sp --graph off --post spatial --obj-first
run objfirst_dp_off --graph dp --post off --obj-first
run objfirst_one_sp --graph one --post spatial --obj-first
run objfirst_dp_sp_split --graph dp --post spatial --obj-first --split-bottom
run objfirst_dp_sp_fill --graph dp --post spatial --obj-first --fill-nan
for t in objfirst_eager_sp objfirst_dp_off objfirst_one_sp objfirst_dp_sp_split objfirst_dp_sp_fill; do
  echo "-- grads eager_off vs $t"; timeout 120 python tools/probes/dp_graph_grad_diff_probe.py diff /tmp/eager_off.pt /tmp/$t.pt 2>&1 | tail -1
done
