#!/bin/bash
# round 5: attention families after the dK/dV register restructure: tests + microbench + rocprof rows
set -u
TAG=${1:-r5_e}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ts() { echo "[$(date +%H:%M:%S)] $*"; }
ts attention tests
timeout 1200 python -m pytest tests/test_gpu_attention.py tests/test_gpu_attention_blocks.py tests/test_gpu_attention_ex.py tests/test_gpu_bert_varlen.py -m gpu -q > $OUT/pytest_attn.log 2>&1; echo "exit $?" >> $OUT/pytest_attn.log; grep -E "^(FAILED|ERROR)|passed|failed|exit" $OUT/pytest_attn.log | head -30
ts rocprof attn
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && GPS_BENCH_WARM=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o attn --output-format csv -- python $REPO/tools/attn_bench.py --only text --iters 10 --json $OUT/attn_bench_text.json > $OUT/prof_attn.log 2>&1; echo "rocprof exit $?")
find /tmp/prof -name '*kernel_stats*.csv' -exec cp {} $OUT/attn_kernel_stats.csv \;
grep -i "attn" $OUT/attn_kernel_stats.csv | cut -c1-160 | head -24
grep "^text" $OUT/prof_attn.log | cut -c1-900
ts done
