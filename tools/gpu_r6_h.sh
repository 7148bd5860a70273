#!/bin/bash
OUT=$PWD/gpurun_out/r6h; mkdir -p $OUT
for cfg in "225 0 0 12608 2304 768" "150 1 0 12608 768 2304" "256 0 0 12608 2304 768" "256 1 0 12608 768 2304" "256 1 0 5120 768 2376"; do set -- $cfg
  echo "== grid $1: form $2 $4x$5x$6"
  GPS_GEMM_SK_GRID=$1 timeout 60 tools/probes/gemm_probe trace $2 $3 $4 $5 $6 13 $OUT/t.bin | tr '\n' ' ' | grep -o '"workgroup": [0-9.]*\|"end_us": {[^}]*}' | tr '\n' ' '; echo
  python3 tools/sk_trace_segments.py $OUT/t.bin
done
VARIANTS=7,12,13 bash tools/gpu_r6_probe.sh r6h bench
