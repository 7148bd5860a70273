#!/bin/bash
# probe bench + traces + the GEMM parity tests
set -u
OUT=$PWD/gpurun_out/r6c; mkdir -p $OUT
VARIANTS=${VARIANTS:-7,12,13} bash tools/gpu_r6_probe.sh r6c bench
bash tools/gpu_r6_trace.sh r6c_trace '0 0 12608 2304 768 12' '0 0 12608 2304 768 7' '1 9 12608 3072 768 12' '0 8 12608 3072 768 12' '0 0 12608 2304 768 13'
grep -h "v7_far" $OUT/gemm_probe_cold.json | grep -o '"what": "[^"]*"\|"v7_far_from_naive": [0-9]*' | paste - - | head -30
timeout 900 python -m pytest tests/test_gpu_gemm.py -m gpu -x -q 2>&1 | tail -5
