#!/bin/bash
# round 5, call 2: spatial plane kernels (two-sweep backward) + block-streaming plain kernels: parity, timing, rocprof.
set -u
TAG=${1:-r5_2}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ts() { echo "[$(date +%H:%M:%S)] $*"; }
ts attention tests
timeout 1200 python -m pytest tests/test_gpu_attention.py tests/test_gpu_attention_blocks.py tests/test_gpu_attention_ex.py -m gpu -q > $OUT/pytest_attn.log 2>&1; echo "exit $?" >> $OUT/pytest_attn.log; grep -E "^(FAILED|ERROR)|passed|failed|exit" $OUT/pytest_attn.log | head -30
ts bench occ4
timeout 400 python tools/attn_bench.py --json $OUT/attn_bench_occ4.json > $OUT/attn_bench_occ4.log 2>&1; grep -v "^  " $OUT/attn_bench_occ4.log | tail -12
ts bench occ3
GPS_ATTN_SP_BWD_OCC=3 timeout 300 python tools/attn_bench.py --only spatial_L80 --json $OUT/attn_bench_occ3.json > $OUT/attn_bench_occ3.log 2>&1; grep "^spatial_L80 {" $OUT/attn_bench_occ3.log
ts phase0 tests
timeout 900 python -m pytest tests/test_gpu_graph_chain.py tests/test_gpu_resume.py tests/test_gpu_bert_varlen.py -m gpu -q > $OUT/pytest_phase0.log 2>&1; echo "exit $?" >> $OUT/pytest_phase0.log; grep -E "^(FAILED|ERROR)|passed|failed|exit" $OUT/pytest_phase0.log | head
ts rocprof attn
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o attn --output-format csv -- python $REPO/tools/attn_bench.py --iters 5 > $OUT/prof_attn.log 2>&1; echo "rocprof exit $?")
find /tmp/prof -name '*kernel_stats*.csv' -exec cp {} $OUT/attn_kernel_stats.csv \;
grep -i "attn" $OUT/attn_kernel_stats.csv | cut -c1-160 | head -24
ts done
