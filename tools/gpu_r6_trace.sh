#!/bin/bash
# workgroup timelines of single GEMM launches: tools/gpu_r6_trace.sh TAG "FORM EPI M N K VARIANT" ...
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
: > $OUT/gemm_probe_trace.jsonl
for cfg in "$@"; do
  timeout 120 ${PROBE:-tools/probes/gemm_probe} trace $cfg | tr '\n' ' ' >> $OUT/gemm_probe_trace.jsonl; echo >> $OUT/gemm_probe_trace.jsonl
done
python3 tools/gemm_trace_table.py $OUT/gemm_probe_trace.jsonl
