#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r4f; mkdir -p $OUT
export TMPDIR=/tmp
P=tools/probes/post_addend_corruption_probe.py
run() { tag=$1; shift; timeout 300 python $P $OUT/$tag.json --save-grads /tmp/$tag.pt "$@" > $OUT/$tag.log 2>&1; echo "== $tag: rc $? $(grep -E 'non-finite|loss' $OUT/$tag.log | tr '\n' ' ')"; }
run eager_off --graph off --post off
run sp_group --graph dp --post spatial
run sp_classic --graph dp --post spatial --no-wgrad-group
run sp_classic_fresh --graph dp --post spatial --no-wgrad-group --ws-mode fresh
run sp_classic_prealloc --graph dp --post spatial --no-wgrad-group --ws-mode prealloc
run sp_classic_nosplit --graph dp --post spatial --no-wgrad-group --no-split
run sp_group_objfirst --graph dp --post spatial --obj-first
run sp_classic_objfirst_prealloc --graph dp --post spatial --obj-first --no-wgrad-group --ws-mode prealloc
for t in sp_group sp_classic sp_classic_fresh sp_classic_prealloc sp_classic_nosplit sp_group_objfirst sp_classic_objfirst_prealloc; do
  echo "-- grads eager_off vs $t"; timeout 120 python tools/probes/dp_graph_grad_diff_probe.py diff /tmp/eager_off.pt /tmp/$t.pt 2>&1 | tail -1
done
echo "== model tests"; timeout 900 python -m pytest tests/test_gpu_model.py -q -x > $OUT/pytest_model.log 2>&1; grep -E "passed|failed|Error|assert" $OUT/pytest_model.log | tail -8
echo "== bq sched"; for s in 0 1 2 3 4 5 6; do GPS_BQ_SCHED=$s timeout 120 python tools/bq_sched_bench.py 2>&1 | tail -1; done
