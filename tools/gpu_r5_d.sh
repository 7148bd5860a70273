#!/bin/bash
# round 5: N = 2 bench tests (self-check), SA single-product mode (test + bench both modes), gemm tests of the factor epilogues
set -u
TAG=${1:-r5_d}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ts() { echo "[$(date +%H:%M:%S)] $*"; }
ts tests
timeout 1200 python -m pytest tests/test_gpu_model.py -m gpu -q -k "bench_n2 or self_spawns" > $OUT/pytest_n2.log 2>&1; echo "exit $?" >> $OUT/pytest_n2.log; grep -E "^(FAILED|ERROR)|passed|failed|exit" $OUT/pytest_n2.log | head
timeout 600 python -m pytest tests/test_gpu_sa_fused.py -m gpu -q -s > $OUT/pytest_sa.log 2>&1; echo "exit $?" >> $OUT/pytest_sa.log; grep -E "^(FAILED|ERROR)|passed|failed|exit|sa-bf16" $OUT/pytest_sa.log | head
ts bench sa-bf16
timeout 600 python bench.py --steps 10 --warmup 3 --sa-bf16 --no-cpu-baseline --no-extras --detail $OUT/bench_sa_bf16_detail.json 2>/dev/null | tail -1 > $OUT/bench_sa_bf16.json; python -c "import json;d=json.load(open('$OUT/bench_sa_bf16.json'));print('sa-bf16',d['value'],d['ms_per_step'],d['config']['final_loss'])"
python - <<PY
import json
d=json.load(open('$OUT/bench_sa_bf16_detail.json'))
for k in d['kernels']:
    if k['kernel'].startswith('sa_mlp'): print(k['kernel'][:70], k['avg_us'], k.get('frac'))
PY
ts done
