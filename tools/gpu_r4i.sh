#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r4i; mkdir -p $OUT
export TMPDIR=/tmp
P=tools/probes/post_addend_corruption_probe.py
run() { tag=$1; shift; timeout 300 python $P $OUT/$tag.json --save-grads /tmp/$tag.pt "$@" > $OUT/$tag.log 2>&1; echo "== $tag: rc $? $(grep -E 'non-finite|loss|stage' $OUT/$tag.log | tr '\n' ' ')"; }
run eager_off --graph off --post off
run of_global --graph dp --post off --obj-first --capture-mode global
run of_relaxed --graph dp --post off --obj-first --capture-mode relaxed
run of_tl --graph dp --post off --obj-first --capture-mode thread_local
run sp_classic_global --graph dp --post spatial --no-wgrad-group --capture-mode global
run sp_classic_relaxed --graph dp --post spatial --no-wgrad-group --capture-mode relaxed
for t in of_global of_relaxed sp_classic_global sp_classic_relaxed; do
  echo "-- grads eager_off vs $t"; timeout 120 python tools/probes/dp_graph_grad_diff_probe.py diff /tmp/eager_off.pt /tmp/$t.pt 2>&1 | tail -1
done
