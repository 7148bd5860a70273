#!/bin/bash
OUT=$PWD/gpurun_out/r6g; mkdir -p $OUT
for e in e1 e2 e3 e4; do
  for cfg in "225 0 0 12608 2304 768" "150 1 0 12608 768 2304"; do set -- $cfg
    echo "== $e grid $1: form $2 $4x$5x$6"
    GPS_GEMM_SK_GRID=$1 timeout 60 tools/probes/gemm_probe_$e trace $2 $3 $4 $5 $6 13 $OUT/t.bin | tr '\n' ' ' | grep -o '"workgroup": [0-9.]*\|"end_us": {[^}]*}' | tr '\n' ' '; echo
    python3 tools/sk_trace_segments.py $OUT/t.bin 2050 | tail -n +2
  done
done
echo "== v12"; timeout 60 tools/probes/gemm_probe_e1 trace 1 0 12608 768 2304 12 | tr '\n' ' ' | grep -o '"us": {[^}]*}\|"end_us": {[^}]*}'
