#!/bin/bash
# round 4: index plan of the variable-length text path in one launch; roofline families by kernel symbol
set -u
OUT=$PWD/gpurun_out/r4_11; mkdir -p $OUT
export TMPDIR=/tmp
echo "== unit tests"; timeout 900 python -m pytest tests/test_gpu_bert_varlen.py tests/test_capi.py tests/test_gpu_point_ops.py::test_library_is_the_hip_one -q -x > $OUT/pytest_units.log 2>&1; grep -E "passed|failed|Error" $OUT/pytest_units.log | tail -3
echo "== model tests"; timeout 900 python -m pytest tests/test_gpu_model.py -q -x > $OUT/pytest_model.log 2>&1; grep -E "passed|failed|Error" $OUT/pytest_model.log | tail -3
echo "== bench"; timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err; python - <<P
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('value','ms_per_step','value_full_length_text','value_with_device_sampler')}); print(d['roofline'])
dd=json.load(open('$OUT/bench_detail.json'))
for f in dd['kernel_families']: print(f['kernel'], f['ms_per_step'], f.get('frac'))
P
tail -2 $OUT/bench.err
echo "== graph-dp"; timeout 600 python bench.py --steps 10 --warmup 3 --graph-dp --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $OUT/bench_graph_dp.json; python -c "import json;d=json.load(open('$OUT/bench_graph_dp.json'));print('graph-dp',d['value'],d['ms_per_step'])"
