#!/bin/bash
# A/B of the attention kernel families inside the bf16-vs-fp32 whole-step comparison + new resident plain kernels' tests/timing
set -u
TAG=${1:-r5_ab}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ts() { echo "[$(date +%H:%M:%S)] $*"; }
ts attention tests
timeout 1200 python -m pytest tests/test_gpu_attention.py tests/test_gpu_attention_blocks.py tests/test_gpu_attention_ex.py -m gpu -q > $OUT/pytest_attn.log 2>&1; echo "exit $?" >> $OUT/pytest_attn.log; grep -E "^(FAILED|ERROR)|passed|failed|exit" $OUT/pytest_attn.log | head -30
ts bench
GPS_BENCH_WARM=1 timeout 400 python tools/attn_bench.py --only joint --json $OUT/attn_bench_joint.json > $OUT/attn_bench_joint.log 2>&1; grep -v "^  " $OUT/attn_bench_joint.log | tail -4
for V in "default::" "general:0:0" "planes_only:1:0" "plain_only:0:5"; do
  IFS=: read NAME PL PM <<< "$V"
  ts "step bounds $NAME"
  GPS_TEST_SPATIAL_PLANES=$PL GPS_TEST_PLAIN_MODE=$PM timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -s -k fp32_oracle_port > $OUT/bounds_$NAME.log 2>&1
  grep "bf16-vs-fp32\|passed\|failed" $OUT/bounds_$NAME.log | sed 's/\[bf16-vs-fp32\] //' | tr '\n' ';' | cut -c1-1800; echo
done
ts rocprof attn
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && GPS_BENCH_WARM=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o attn --output-format csv -- python $REPO/tools/attn_bench.py --only joint --iters 10 > $OUT/prof_attn.log 2>&1; echo "rocprof exit $?")
find /tmp/prof -name '*kernel_stats*.csv' -exec cp {} $OUT/attn_kernel_stats.csv \;
grep -i "attn" $OUT/attn_kernel_stats.csv | cut -c1-160 | head -24
ts done
