#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r4q; mkdir -p $OUT
export TMPDIR=/tmp
P=$PWD/tools/probes/post_addend_corruption_probe.py
run() { tag=$1; shift; timeout 300 python $P $OUT/$tag.json --save-grads /tmp/$tag.pt "$@" > $OUT/$tag.log 2>&1; echo "== $tag: rc $? $(grep -E 'non-finite|loss' $OUT/$tag.log | tr '\n' ' ' | cut -c1-200)"; }
run eager_off --graph off --post off
run sp_classic --graph dp --post spatial --no-wgrad-group
run sp_group --graph dp --post spatial
run all_group --graph dp --post all
run of_off --graph dp --post off --obj-first
run of_sp_classic --graph dp --post spatial --obj-first --no-wgrad-group
run one_all --graph one --post all
for t in sp_classic sp_group all_group of_off of_sp_classic one_all; do
  echo "-- grads eager_off vs $t"; timeout 120 python tools/probes/dp_graph_grad_diff_probe.py diff /tmp/eager_off.pt /tmp/$t.pt 2>&1 | tail -1
done
mkdir -p /tmp/dots && cd /tmp/dots
DEBUG_HIP_GRAPH_DOT_PRINT=1 timeout 300 python $P $OUT/dot.json --graph dp --post spatial --no-wgrad-group > $OUT/dot.log 2>&1; echo "rc $?"
for f in /tmp/dots/graph_*; do python /root/repo/tools/probes/dot_shape.py $f; done 2>&1 | head -20
cd /root/repo
echo "== model tests"; timeout 900 python -m pytest tests/test_gpu_model.py -q -x > $OUT/pytest_model.log 2>&1; grep -E "passed|failed|Error|assert" $OUT/pytest_model.log | tail -8
