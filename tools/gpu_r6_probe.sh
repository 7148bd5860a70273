#!/bin/bash
# One gpurun call of round 6: the standalone GEMM probe (tools/probes/gemm_probe, no python): cold / warm per-shape times
# of the tile variants, workgroup timelines, and PMC passes (matrix-pipe busy cycles, wave cycles) of the heaviest shapes.
#   tools/gpu_r6_probe.sh TAG [bench] [trace] [pmc]
set -u
ulimit -c 0
TAG=${1:-r6_probe}; shift
WHAT=${*:-bench trace pmc}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
P=$PWD/tools/probes/gemm_probe
VARIANTS=${VARIANTS:--1,6,7,12}
ts() { echo "[$(date +%H:%M:%S)] $*"; }
{ nproc; rocm-smi --showproductname 2>&1 | grep -i -m2 "card series\|gfx"; } > $OUT/host.txt 2>&1
if [[ $WHAT == *bench* ]]; then
  ts bench cold; timeout 600 $P bench --variants $VARIANTS > $OUT/gemm_probe_cold.json 2> $OUT/gemm_probe_cold.err; echo "exit $?"
  ts bench warm; timeout 600 $P bench --warm --variants $VARIANTS > $OUT/gemm_probe_warm.json 2> $OUT/gemm_probe_warm.err; echo "exit $?"
  python3 tools/gemm_probe_table.py $OUT/gemm_probe_cold.json $OUT/gemm_probe_warm.json | tee $OUT/gemm_probe_table.txt
fi
if [[ $WHAT == *trace* ]]; then
  ts trace
  : > $OUT/gemm_probe_trace.jsonl
  for cfg in ${TRACES:-"0 0 12608 2304 768 7" "0 0 12608 2304 768 12" "1 0 12608 768 2304 12" "1 0 12608 768 2304 7" "0 8 12608 3072 768 7" "0 8 12608 3072 768 12" "1 9 12608 3072 768 7" "1 0 5120 768 2376 6" "1 0 5120 768 2376 7" "0 8 5120 2048 768 7" "0 0 8320 768 768 6" "1 4 8320 2048 768 7"}; do
    timeout 120 $P trace $cfg | tr '\n' ' ' >> $OUT/gemm_probe_trace.jsonl; echo >> $OUT/gemm_probe_trace.jsonl
  done
  cat $OUT/gemm_probe_trace.jsonl
fi
if [[ $WHAT == *pmc* ]]; then
  ts pmc
  i=0
  for cfg in ${PMCS:-"1 9 12608 3072 768 -1" "1 0 12608 768 2304 -1" "0 8 12608 3072 768 -1" "1 0 12608 768 3072 -1" "1 4 8320 2048 768 -1"}; do
    i=$((i+1)); g=0
    for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM GRBM_GUI_ACTIVE"; do
      g=$((g+1))
      rm -rf /tmp/pmc_${i}_$g
      (cd /tmp && timeout 120 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_${i}_$g -o p --output-format csv -- $P loop $cfg 12 > $OUT/pmc_${i}_$g.log 2>&1; echo "pmc $i/$g exit $?")
      f=$(find /tmp/pmc_${i}_$g -name '*counter_collection.csv' | head -1)
      { echo "## gemm_probe loop $cfg  -- $grp"; python3 tools/pmc_summary.py $f 5 | grep -v "^fill_\|^diff_" ; } >> $OUT/pmc_gemm_shapes.txt
    done
  done
  cat $OUT/pmc_gemm_shapes.txt
fi
ts done
