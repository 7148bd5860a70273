#!/bin/bash
# round 4, call A: trainer drop-in on the GPU, root-cause probe of the post-addend corruption
set -u
OUT=$PWD/gpurun_out/r4a; mkdir -p $OUT
export TMPDIR=/tmp
ts() { echo "[$(date +%H:%M:%S)] $*"; }
ts dropin; timeout 600 python -m pytest tests/test_reference_trainer_dropin.py tests/test_gpu_bert_varlen.py -m gpu -q -x > $OUT/pytest_dropin.log 2>&1; tail -5 $OUT/pytest_dropin.log
ts probe_off; timeout 300 python tools/probes/post_addend_corruption_probe.py $OUT/probe_off.json --post off --save-grads /tmp/g_off.pt > $OUT/probe_off.log 2>&1; tail -5 $OUT/probe_off.log
ts probe_spatial; timeout 300 python tools/probes/post_addend_corruption_probe.py $OUT/probe_spatial.json --post spatial --save-grads /tmp/g_sp.pt > $OUT/probe_spatial.log 2>&1; tail -60 $OUT/probe_spatial.log
ts diff; timeout 120 python tools/probes/dp_graph_grad_diff_probe.py diff /tmp/g_off.pt /tmp/g_sp.pt > $OUT/diff_off_spatial.log 2>&1; tail -12 $OUT/diff_off_spatial.log
ts probe_one; timeout 300 python tools/probes/post_addend_corruption_probe.py $OUT/probe_one.json --post spatial --graph one --save-grads /tmp/g_one.pt > $OUT/probe_one.log 2>&1; tail -5 $OUT/probe_one.log
timeout 120 python tools/probes/dp_graph_grad_diff_probe.py diff /tmp/g_off.pt /tmp/g_one.pt > $OUT/diff_off_one.log 2>&1; tail -5 $OUT/diff_off_one.log
ts done
