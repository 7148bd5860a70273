#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r4_8; mkdir -p $OUT
export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_model.py -q -x -k "grouped or segmented or hip_graph or bench_config" > $OUT/pytest.log 2>&1; grep -E "passed|failed|Error|assert" $OUT/pytest.log | tail -5
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err; python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])
det=json.load(open("$OUT/bench_detail.json"))
for k in det['kernel_families'][:6]: print(k['kernel'][:60], k['ms_per_step'], k.get('frac'))
PY
rm -rf /tmp/pmc && mkdir -p /tmp/pmc
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc -o fetch --output-format csv -- python $PWD/tools/pmc_workload.py > $OUT/pmc_fetch.log 2>&1; echo "pmc fetch exit $?")
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc -o write --output-format csv -- python $PWD/tools/pmc_workload.py > $OUT/pmc_write.log 2>&1; echo "pmc write exit $?")
python tools/pmc_traffic.py /tmp/pmc/fetch_counter_collection.csv /tmp/pmc/write_counter_collection.csv $OUT/pmc_traffic.json > $OUT/pmc_traffic.log 2>&1; grep -A4 '"gemm_tn_grouped' $OUT/pmc_traffic.json | head -12
