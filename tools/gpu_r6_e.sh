#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r6e; mkdir -p $OUT
# correctness first, under a short timeout (a hang must not take the box down)
timeout 60 tools/probes/gemm_probe bench --set small --variants 7,13 --rounds 2 --inner 2 > $OUT/small.json 2> $OUT/small.err; echo "small exit $?"; cat $OUT/small.err | head -5
python3 tools/gemm_probe_table.py $OUT/small.json
VARIANTS=${VARIANTS:-7,12,13} timeout 300 bash tools/gpu_r6_probe.sh r6e bench
for t in "0 0 12608 2304 768 13 sk_nt" "1 0 12608 768 2304 13 sk_nn" "1 0 5120 768 2376 13 sk_c" "1 4 8320 2048 768 13 sk_b"; do set -- $t; timeout 60 tools/probes/gemm_probe trace $1 $2 $3 $4 $5 $6 $OUT/$7.bin | tr '\n' ' '; echo; python3 tools/sk_trace_segments.py $OUT/$7.bin 2050; done
