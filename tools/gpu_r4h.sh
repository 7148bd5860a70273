#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r4h; mkdir -p $OUT
export TMPDIR=/tmp
P=tools/probes/post_addend_corruption_probe.py
run() { tag=$1; shift; timeout 300 python $P $OUT/$tag.json --save-grads /tmp/$tag.pt "$@" > $OUT/$tag.log 2>&1; echo "== $tag: rc $? $(grep -E 'non-finite|loss|stage' $OUT/$tag.log | tr '\n' ' ')"; }
run eager_off --graph off --post off
run of_lang_only --graph dp --post off --obj-first --bottom-inputs lang_encoder
run of_point_only --graph dp --post off --obj-first --bottom-inputs point_encoder
run of_eager_g2b --graph dp --post off --obj-first --eager-g2b
run of_clone_roots --graph dp --post off --obj-first --clone-roots
run of_classic --graph dp --post off --obj-first --no-wgrad-group
for t in of_eager_g2b of_clone_roots; do
  echo "-- grads eager_off vs $t"; timeout 120 python tools/probes/dp_graph_grad_diff_probe.py diff /tmp/eager_off.pt /tmp/$t.pt 2>&1 | tail -1
done
echo "== point ops"; timeout 600 python -m pytest tests/test_gpu_point_ops.py tests/test_gpu_vs_reference_ext.py -q -x 2>&1 | grep -E "passed|failed" | tail -3
echo "== bq sched"; for s in 0 2; do GPS_BQ_SCHED=$s timeout 120 python tools/bq_sched_bench.py 2>&1 | tail -1; done
