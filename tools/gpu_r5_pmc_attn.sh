#!/bin/bash
# rocprofv3 --pmc passes over tools/attn_bench.py (counters only, with --kernel-trace): instruction mix and matrix-pipe
# busy cycles of the attention kernels of the bench shapes
set -u
ulimit -c 0
OUT=$PWD/gpurun_out/${1:-r5_pmc_attn}
mkdir -p $OUT
export TMPDIR=/tmp GPS_BENCH_WARM=1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  (cd /tmp && timeout 200 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_$i -o p --output-format csv -- python $OLDPWD/tools/attn_bench.py --iters 6 > $OUT/pmc_$i.log 2>&1; echo "pmc $i exit $?")
  f=$(find /tmp/pmc_$i -name '*counter_collection.csv' | head -1)
  python tools/pmc_summary.py $f 5 | grep -A12 "gps_attn" > $OUT/pmc_attn_g$i.txt
done
head -60 $OUT/pmc_attn_g1.txt
