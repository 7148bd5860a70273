#!/bin/bash
OUT=$PWD/gpurun_out/r6f; mkdir -p $OUT
for grid in 225 150 256; do
  echo "== GPS_GEMM_SK_GRID=$grid nt 12608x2304x768"
  GPS_GEMM_SK_GRID=$grid timeout 60 tools/probes/gemm_probe trace 0 0 12608 2304 768 13 $OUT/sk_$grid.bin | tr '\n' ' ' | grep -o '"wall_us": [0-9.]*\|"workgroup": [0-9.]*\|"end_us": {[^}]*}' | tr '\n' ' '; echo
  python3 tools/sk_trace_segments.py $OUT/sk_$grid.bin 2050
done
echo "== v12 for reference"; timeout 60 tools/probes/gemm_probe trace 0 0 12608 2304 768 12 | tr '\n' ' ' | grep -o '"us": {[^}]*}\|"end_us": {[^}]*}'
