#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r4d; mkdir -p $OUT
export TMPDIR=/tmp
P=tools/probes/post_addend_corruption_probe.py
run() { tag=$1; shift; timeout 300 python $P $OUT/$tag.json --save-grads /tmp/$tag.pt --taps /tmp/$tag.taps "$@" > $OUT/$tag.log 2>&1; echo "== $tag: $(grep -E 'non-finite|loss|taps' $OUT/$tag.log | tr '\n' ' ')"; }
run eager_off --graph off --post off
run dp_off --graph dp --post off
run dp_sp --graph dp --post spatial
run dp_sp_objfirst --graph dp --post spatial --obj-first
for t in dp_off dp_sp dp_sp_objfirst; do
  echo "-- grads eager_off vs $t"; timeout 120 python tools/probes/dp_graph_grad_diff_probe.py diff /tmp/eager_off.pt /tmp/$t.pt 2>&1 | tail -3
done
echo "-- taps eager_off vs dp_off"; python tools/probes/tap_diff.py /tmp/eager_off.taps /tmp/dp_off.taps > $OUT/taps_dp_off.txt; grep -c "<<<<" $OUT/taps_dp_off.txt
echo "-- taps eager_off vs dp_sp"; python tools/probes/tap_diff.py /tmp/eager_off.taps /tmp/dp_sp.taps > $OUT/taps_dp_sp.txt; cat $OUT/taps_dp_sp.txt
