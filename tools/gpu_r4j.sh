#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r4j; mkdir -p $OUT
export TMPDIR=/tmp
P=tools/probes/post_addend_corruption_probe.py
run() { tag=$1; shift; timeout 300 python $P $OUT/$tag.json --save-grads /tmp/$tag.pt "$@" > $OUT/$tag.log 2>&1; echo "== $tag: rc $? $(grep -E 'non-finite|loss|stage' $OUT/$tag.log | tr '\n' ' ')"; }
run eager_off --graph off --post off
run of_st --graph dp --post off --obj-first --single-thread-backward
run sp_classic_st --graph dp --post spatial --no-wgrad-group --single-thread-backward
run of_sp_classic_st --graph dp --post spatial --obj-first --no-wgrad-group --single-thread-backward
run of_mt --graph dp --post off --obj-first
for t in of_st sp_classic_st of_sp_classic_st; do
  echo "-- grads eager_off vs $t"; timeout 120 python tools/probes/dp_graph_grad_diff_probe.py diff /tmp/eager_off.pt /tmp/$t.pt 2>&1 | tail -1
done
