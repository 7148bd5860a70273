#!/bin/bash
set -u
ulimit -c 0
TAG=${1:-r5_n}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for v in "distinct:" "all_slots:--all-object-slots" "distinct2:"; do
  n=${v%%:*}; f=${v#*:}
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras $f 2>/dev/null | tail -1 > $OUT/bench_$n.json
  python -c "import json;d=json.load(open('$OUT/bench_$n.json'));print('$n',d['value'],d['ms_per_step'])"
done
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_extras.json
python -c "import json;d=json.load(open('$OUT/bench_extras.json'));print({k:v for k,v in d.items() if k.startswith('value') or k.startswith('ms_')}, d['config'].get('pad_object_fraction'))"
