#!/bin/bash
# round 5, call 3: root-cause the attn_bench segfault (rocgdb backtrace), then the timings.
set -u
TAG=${1:-r5_3}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ts() { echo "[$(date +%H:%M:%S)] $*"; }
ts bench warm
GPS_BENCH_WARM=1 timeout 400 python tools/attn_bench.py --json $OUT/attn_bench_occ4.json > $OUT/attn_bench_occ4.log 2>&1; grep -v "^  " $OUT/attn_bench_occ4.log | tail -12
ts bench occ3
GPS_BENCH_WARM=1 GPS_ATTN_SP_BWD_OCC=3 timeout 300 python tools/attn_bench.py --only spatial_L80 --json $OUT/attn_bench_occ3.json > $OUT/attn_bench_occ3.log 2>&1; grep "^spatial_L80 {" $OUT/attn_bench_occ3.log
ts rocprof attn
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && GPS_BENCH_WARM=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o attn --output-format csv -- python $REPO/tools/attn_bench.py --iters 10 > $OUT/prof_attn.log 2>&1; echo "rocprof exit $?")
find /tmp/prof -name '*kernel_stats*.csv' -exec cp {} $OUT/attn_kernel_stats.csv \;
grep -i "attn" $OUT/attn_kernel_stats.csv | cut -c1-160 | head -24
ts done
