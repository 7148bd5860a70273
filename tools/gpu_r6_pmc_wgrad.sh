#!/bin/bash
# PMC passes over the grouped weight-gradient launch (tools/probes/gemm_probe wgrad) and, for comparison, the NN form of the
# same 256 x 256 kernel on a long reduction
set -u
ulimit -c 0
OUT=$PWD/gpurun_out/${1:-r6_pmc_wgrad}; mkdir -p $OUT
export TMPDIR=/tmp
P=$PWD/${PROBE:-tools/probes/gemm_probe}
: > $OUT/pmc_wgrad.txt
g=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  g=$((g+1))
  for what in "wgrad" "loop 1 0 12608 768 2304 12 8"; do
    rm -rf /tmp/pmc_$g
    (cd /tmp && timeout 120 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_$g -o p --output-format csv -- $P $what > $OUT/pmc_$g.log 2>&1; echo "pmc $g exit $?")
    f=$(find /tmp/pmc_$g -name '*counter_collection.csv' | head -1)
    { echo "## gemm_probe $what -- $grp"; python3 tools/pmc_summary.py $f 2 | grep -A16 "wgrad_grouped_kernel\|gemm8p_kernel" ; } >> $OUT/pmc_wgrad.txt
  done
done
cat $OUT/pmc_wgrad.txt
