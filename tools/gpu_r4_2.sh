#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r4_2; mkdir -p $OUT
export TMPDIR=/tmp
echo "== bench"; timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; tail -c 2500 $OUT/bench.json; cp gpurun_out/bench_detail.json $OUT/bench_detail.json
echo "== bench classic wgrad"; timeout 900 python bench.py --steps 10 --warmup 3 --no-wgrad-group --no-cpu-baseline --no-extras 2> /dev/null | tail -c 600
echo "== bench graph-dp"; timeout 900 python bench.py --steps 10 --warmup 3 --graph-dp --no-cpu-baseline --no-extras 2> /dev/null | tail -c 600
echo "== step_attrib"; timeout 600 python tools/step_attrib.py --out $OUT/step_attrib.txt > $OUT/step_attrib.log 2>&1; tail -3 $OUT/step_attrib.log
echo "== chain tests"; timeout 900 python -m pytest tests/test_gpu_graph_chain.py -q -x 2>&1 | grep -E "passed|failed|skipped" | tail -3
