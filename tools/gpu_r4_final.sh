#!/bin/bash
# round 4, final measurement call: full GPU test suite, bench line (+ detail), kernel bench, rocprofv3 kernel stats of
# the same bench command, PMC traffic passes, the other presets.  Outputs under gpurun_out/<tag>/.
set -u
TAG=${1:-r4_final}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ts() { echo "[$(date +%H:%M:%S)] $*"; }
{ echo "nproc $(nproc)"; free -g | head -2; rocm-smi --showproductname 2>&1 | grep -i -m2 "card series\|gfx"; } > $OUT/host.txt 2>&1
ts pytest; timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|exit" $OUT/pytest_gpu.log | head -20
ts bench; timeout 900 python bench.py --steps 10 --warmup 3 --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
tail -c 3500 $OUT/bench.json; tail -3 $OUT/bench.err
ts kernel_bench; timeout 300 python tools/kernel_bench.py --json $OUT/kernel_bench.json > $OUT/kernel_bench.log 2>&1; tail -3 $OUT/kernel_bench.log
ts rocprof
rm -rf /tmp/prof && mkdir -p /tmp/prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench --output-format csv -- python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --detail $OUT/bench_prof_detail.json > $OUT/prof_bench.log 2>&1; echo "rocprof exit $?")
find /tmp/prof -name '*kernel_stats*.csv' -exec cp {} $OUT/bench_kernel_stats.csv \;
tail -2 $OUT/prof_bench.log | cut -c1-400
ts pmc
rm -rf /tmp/pmc && mkdir -p /tmp/pmc
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc -o fetch --output-format csv -- python $REPO/tools/pmc_workload.py > $OUT/pmc_fetch.log 2>&1; echo "pmc fetch exit $?")
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc -o write --output-format csv -- python $REPO/tools/pmc_workload.py > $OUT/pmc_write.log 2>&1; echo "pmc write exit $?")
python tools/pmc_traffic.py /tmp/pmc/fetch_counter_collection.csv /tmp/pmc/write_counter_collection.csv $OUT/pmc_traffic.json > $OUT/pmc_traffic.log 2>&1; grep -A3 "gemm_tn_grouped\|gemm_nn(M=22400,N=768,K=3072" $OUT/pmc_traffic.json | head -20
ts presets
timeout 600 python bench.py --steps 10 --warmup 3 --graph-dp --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $OUT/bench_graph_dp.json; python -c "import json;d=json.load(open('$OUT/bench_graph_dp.json'));print('graph-dp',d['value'],d['ms_per_step'])"
timeout 600 python bench.py --steps 10 --warmup 3 --no-varlen --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $OUT/bench_padded.json; python -c "import json;d=json.load(open('$OUT/bench_padded.json'));print('padded',d['value'],d['ms_per_step'])"
timeout 600 python bench.py --config finetune --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $OUT/bench_finetune.json; python -c "import json;d=json.load(open('$OUT/bench_finetune.json'));print('finetune',d['value'],d['ms_per_step'])"
timeout 600 python bench.py --config stress --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $OUT/bench_stress.json; python -c "import json;d=json.load(open('$OUT/bench_stress.json'));print('stress',d['value'],d['ms_per_step'])"
timeout 600 python bench.py --config stress --fp8 --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $OUT/bench_stress_fp8.json; python -c "import json;d=json.load(open('$OUT/bench_stress_fp8.json'));print('stress fp8',d['value'],d['ms_per_step'])"
ts smoke; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
ts done; du -sh $REPO/gpurun_out
