#!/bin/bash
# A/B of whole bench runs in ONE box: plain attention mode 5 vs 7, SA x3 vs single product, GELU factor on/off
set -u
TAG=${1:-r5_f}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() { name=$1; shift; timeout 600 env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras ${EXTRA:-} 2>/dev/null | tail -1 > $OUT/bench_$name.json; python -c "import json;d=json.load(open('$OUT/bench_$name.json'));print('$name',d['value'],d['ms_per_step'],d['headline']['attention_core_ms_per_step'],d['headline']['native_gemms_ms_per_step'])"; }
run default_1 A=1
run plain7 GPS_ATTN_PLAIN_MODE=7
run plain0 GPS_ATTN_PLAIN_MODE=0
EXTRA=--sa-bf16 run sa_bf16 A=1
run default_2 A=1
