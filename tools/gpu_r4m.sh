#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r4m; mkdir -p $OUT
export TMPDIR=/tmp
P=tools/probes/post_addend_corruption_probe.py
timeout 300 python $P $OUT/eager_off.json --graph off --post off --save-grads /tmp/eager_off.pt > $OUT/eager_off.log 2>&1
timeout 300 python $P $OUT/alias.json --graph dp --post spatial --no-wgrad-group --alias-scan --save-grads /tmp/alias.pt > $OUT/alias.log 2>&1; echo "rc $?"; grep -E "non-finite|alias|ALIAS" $OUT/alias.log | head -40; grep "region " $OUT/alias.log | head -30
timeout 120 python tools/probes/dp_graph_grad_diff_probe.py diff /tmp/eager_off.pt /tmp/alias.pt 2>&1 | tail -2
