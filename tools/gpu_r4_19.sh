#!/bin/bash
# round 4: full GPU suite + smoke + default bench line on the last commit
set -u
OUT=$PWD/gpurun_out/r4_19; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|exit" $OUT/pytest_gpu.log | head -10
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json; python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print({k:d.get(k) for k in ('value','ms_per_step','value_full_length_text','value_with_device_sampler')}, d['roofline']['frac'])"
