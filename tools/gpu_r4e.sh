#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r4e; mkdir -p $OUT
export TMPDIR=/tmp
P=tools/probes/post_addend_corruption_probe.py
run() { tag=$1; shift; timeout 300 python $P $OUT/$tag.json --save-grads /tmp/$tag.pt "$@" > $OUT/$tag.log 2>&1; echo "== $tag: rc $? $(grep -E 'non-finite|loss' $OUT/$tag.log | tr '\n' ' ')"; }
run eager_off --graph off --post off
GPS_POST_VARIANT=noattr run dp_sp_noattr --graph dp --post spatial
GPS_POST_VARIANT=nobf16 run dp_sp_nobf16 --graph dp --post spatial
run dp_plain --graph dp --post plain
run dp_all --graph dp --post all
for t in dp_sp_noattr dp_sp_nobf16 dp_plain dp_all; do
  echo "-- grads eager_off vs $t"; timeout 120 python tools/probes/dp_graph_grad_diff_probe.py diff /tmp/eager_off.pt /tmp/$t.pt 2>&1 | tail -2
done
echo "== gemm tests"; timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x -k "grouped or wgrad or deferred or linear_and_packed or ffn_autograd" 2>&1 | tail -5
echo "== model tests"; timeout 900 python -m pytest tests/test_gpu_model.py -q -x 2>&1 | tail -5
