#!/bin/bash
# round 4: hardware counters of the grouped weight-gradient launch (and everything else in tools/pmc_workload.py)
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/${1:-r4_14}; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); tag=g$i
  rm -rf /tmp/pmc_$tag
  (cd /tmp && timeout 300 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_$tag -o p --output-format csv -- python $REPO/tools/pmc_workload.py > $OUT/pmc_$tag.log 2>&1; echo "pmc $tag exit $?")
  f=$(find /tmp/pmc_$tag -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python $REPO/tools/pmc_summary.py "$f" 3 > $OUT/pmc_$tag.txt 2>&1
  k=$(find /tmp/pmc_$tag -name '*kernel_trace.csv' | head -1)
  [ -n "$k" ] && python - "$k" > $OUT/durations_$tag.txt <<'P'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print(f"{sum(v)/len(v):10.2f} us x{len(v):4d}  {k[:110]}")
P
done
grep -A9 "wgrad_grouped_kernel" $OUT/pmc_g*.txt | cut -c1-120
grep "wgrad_grouped" $OUT/durations_g*.txt
