#!/bin/bash
# round 5, call 1: the new spatial attention kernels (parity + timing, both register budgets of the backward), the
# phase-0 fixes (graph-chain topology test, prefix-mask poison, resume), host info.
set -u
TAG=${1:-r5_1}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ts() { echo "[$(date +%H:%M:%S)] $*"; }
ts attention tests
timeout 900 python -m pytest tests/test_gpu_attention.py -m gpu -q -x > $OUT/pytest_attn.log 2>&1; echo "exit $?" >> $OUT/pytest_attn.log; tail -5 $OUT/pytest_attn.log
ts bench occ4
timeout 300 python tools/attn_bench.py --json $OUT/attn_bench_occ4.json > $OUT/attn_bench_occ4.log 2>&1; tail -6 $OUT/attn_bench_occ4.log
ts bench occ3
GPS_ATTN_SP_BWD_OCC=3 timeout 300 python tools/attn_bench.py --json $OUT/attn_bench_occ3.json > $OUT/attn_bench_occ3.log 2>&1; grep spatial_L80 $OUT/attn_bench_occ3.log
ts phase0 tests
timeout 900 python -m pytest tests/test_gpu_graph_chain.py tests/test_gpu_resume.py tests/test_gpu_bert_varlen.py tests/test_gpu_embedding.py -m gpu -q > $OUT/pytest_phase0.log 2>&1; echo "exit $?" >> $OUT/pytest_phase0.log; tail -15 $OUT/pytest_phase0.log
ts rocprof attn
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o attn --output-format csv -- python $REPO/tools/attn_bench.py --iters 5 > $OUT/prof_attn.log 2>&1; echo "rocprof exit $?")
find /tmp/prof -name '*kernel_stats*.csv' -exec cp {} $OUT/attn_kernel_stats.csv \;
grep -i "attn" $OUT/attn_kernel_stats.csv | cut -c1-200 | head -20
ts done
