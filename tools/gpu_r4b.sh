#!/bin/bash
# round 4, call B: is the "post-addend" failure an uninitialised read?  NaN-filled torch.empty in every mode
set -u
OUT=$PWD/gpurun_out/r4b; mkdir -p $OUT
export TMPDIR=/tmp
P=tools/probes/post_addend_corruption_probe.py
run() { tag=$1; shift; timeout 300 python $P $OUT/$tag.json --save-grads /tmp/$tag.pt "$@" > $OUT/$tag.log 2>&1; echo "== $tag: $(grep -E 'non-finite|CHANGED|loss' $OUT/$tag.log | tr '\n' ' ')"; }
run eager_off --graph off --post off
run eager_off_fill --graph off --post off --fill-nan
run eager_sp_fill --graph off --post spatial --fill-nan
run dp_off_fill --graph dp --post off --fill-nan
run dp_sp_fill --graph dp --post spatial --fill-nan
run one_off_fill --graph one --post off --fill-nan
run dp_sp --graph dp --post spatial
for t in eager_off_fill eager_sp_fill dp_off_fill dp_sp_fill one_off_fill dp_sp; do
  echo "-- diff eager_off vs $t"; timeout 120 python tools/probes/dp_graph_grad_diff_probe.py diff /tmp/eager_off.pt /tmp/$t.pt 2>&1 | tail -4
done
tail -40 $OUT/dp_sp.log
