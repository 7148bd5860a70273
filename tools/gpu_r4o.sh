#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r4o; mkdir -p $OUT
export TMPDIR=/tmp
P=tools/probes/post_addend_corruption_probe.py
run() { tag=$1; shift; timeout 300 python $P $OUT/$tag.json --save-grads /tmp/$tag.pt "$@" > $OUT/$tag.log 2>&1; echo "== $tag: rc $? $(grep -E 'non-finite|loss' $OUT/$tag.log | tr '\n' ' ' | cut -c1-200)"; }
run eager_off --graph off --post off
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run sp_classic_nopkt --graph dp --post spatial --no-wgrad-group
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 run of_nopkt --graph dp --post off --obj-first
DEBUG_HIP_FORCE_GRAPH_QUEUES=1 run sp_classic_q1 --graph dp --post spatial --no-wgrad-group
DEBUG_HIP_FORCE_GRAPH_QUEUES=1 run of_q1 --graph dp --post off --obj-first
DEBUG_HIP_GRAPH_BATCH_SIZE=1 run sp_classic_b1 --graph dp --post spatial --no-wgrad-group
for t in sp_classic_nopkt of_nopkt sp_classic_q1 of_q1 sp_classic_b1; do
  echo "-- grads eager_off vs $t"; timeout 120 python tools/probes/dp_graph_grad_diff_probe.py diff /tmp/eager_off.pt /tmp/$t.pt 2>&1 | tail -1
done
