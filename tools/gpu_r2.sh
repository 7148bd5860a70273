#!/bin/bash
# Round-2 gpurun driver: stages selected by name; outputs under gpurun_out/<tag>/.
set -u
TAG=${1:-r2a}
WHAT=${2:-gemm}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ts() { echo "[$(date +%H:%M:%S)] $*"; }
{ echo "nproc $(nproc)"; free -g | head -2; } > $OUT/host.txt 2>&1
if [[ $WHAT == *trprobe* ]]; then
  ts trprobe; timeout 120 python tools/probes/run_tr_b16_probe.py > $OUT/tr_probe.txt 2>&1; echo "tr probe exit $?"; head -20 $OUT/tr_probe.txt
fi
if [[ $WHAT == *gemmtest* ]]; then
  ts gemmtest; timeout 900 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x > $OUT/pytest_gemm.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gemm.log
  tail -25 $OUT/pytest_gemm.log
fi
if [[ $WHAT == *gemmbench* ]]; then
  ts gemmbench; timeout 600 python tools/gemm_bench.py --json $OUT/gemm_bench.json > $OUT/gemm_bench.log 2>&1; echo "gemm bench exit $?"; tail -20 $OUT/gemm_bench.log
fi
if [[ $WHAT == *gpmc2* ]]; then
  ts gpmc2
  for cfg in ${GPMC_CFGS:-nt:7 nt:4 nt:11}; do
    for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
      tag=$(echo ${cfg}_${grp%% *} | tr ':' '_')
      rm -rf /tmp/pmc_$tag
      (cd /tmp && timeout 120 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_$tag -o p --output-format csv -- python $REPO/tools/gemm_bench.py --only 8 --pmc-loop $cfg > $OUT/pmc_$tag.log 2>&1; echo "pmc $tag exit $?")
      f=$(find /tmp/pmc_$tag -name '*counter_collection.csv' | head -1)
      [ -n "$f" ] && python $REPO/tools/pmc_summary.py "$f" > $OUT/pmc_$tag.txt 2>&1
    done
  done
  tail -n 12 $OUT/pmc_*.txt
fi
if [[ $WHAT == *bqpmc* ]]; then
  ts bqpmc
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES"; do
    tag=bq_${grp%% *}
    rm -rf /tmp/pmc_$tag
    (cd /tmp && timeout 120 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_$tag -o p --output-format csv -- python $REPO/tools/bq_pmc_loop.py > $OUT/pmc_$tag.log 2>&1; echo "pmc $tag exit $?")
    f=$(find /tmp/pmc_$tag -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python $REPO/tools/pmc_summary.py "$f" > $OUT/pmc_$tag.txt 2>&1
  done
  tail -n 22 $OUT/pmc_bq_*.txt
fi
if [[ $WHAT == *benchoverlap* ]]; then
  ts benchoverlap
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --detail $OUT/bench_overlap_detail.json > $OUT/bench_overlap.json 2> $OUT/bench_overlap.err; echo "bench overlap exit $?"; tail -c 1200 $OUT/bench_overlap.json | head -c 700; echo; tail -2 $OUT/bench_overlap.err
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-wgrad-overlap > $OUT/bench_nooverlap.json 2> $OUT/bench_nooverlap.err; echo "bench no-overlap exit $?"; tail -c 1200 $OUT/bench_nooverlap.json | head -c 400; echo; tail -2 $OUT/bench_nooverlap.err
fi
if [[ $WHAT == *varlen* ]]; then
  ts varlen
  timeout 900 python -m pytest tests/test_gpu_bert_varlen.py tests/test_gpu_fused_norm.py tests/test_gpu_model.py tests/test_gpu_attention.py tests/test_gpu_gemm.py tests/test_gpu_losses.py -m gpu -q -x > $OUT/pytest_varlen.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_varlen.log
  grep -E "^(FAILED|ERROR)|passed|failed|exit|Error|assert" $OUT/pytest_varlen.log | head -30 | cut -c1-400
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --detail $OUT/bench_varlen_detail.json > $OUT/bench_varlen.json 2> $OUT/bench_varlen.err; echo "bench varlen exit $?"; tail -c 2200 $OUT/bench_varlen.json; echo; tail -2 $OUT/bench_varlen.err
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-varlen > $OUT/bench_padded.json 2> $OUT/bench_padded.err; echo "bench padded exit $?"; tail -c 2200 $OUT/bench_padded.json | head -c 300; echo; tail -2 $OUT/bench_padded.err
fi
if [[ $WHAT == *gemmquick* ]]; then
  ts gemmquick
  for i in 8 10 11 4 0; do timeout 200 python tools/gemm_bench.py --only $i --rounds 8 >> $OUT/gemm_quick.log 2>&1; done; echo "gemm quick exit $?"; grep "^{" $OUT/gemm_quick.log | cut -c1-1200
fi
if [[ $WHAT == *newtests* ]]; then
  ts newtests; timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_optim.py tests/test_a16_vs_golden.py tests/test_gpu_attention.py tests/test_gpu_model.py -m gpu -q -x > $OUT/pytest_new.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_new.log
  tail -15 $OUT/pytest_new.log | cut -c1-300
fi
if [[ $WHAT == *benchnative* ]]; then
  ts bench-native
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_native.json 2> $OUT/bench_native.err; echo "bench native exit $?"
  tail -c 600 $OUT/bench_native.err
  python - <<PYEOF
import json
d=json.loads(open("$OUT/bench_native.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"])
for k in d["kernels"][:14]: print(k["kernel"][:60], k["launches_per_step"], k["avg_us"], k["ms_per_step"], k["frac"])
for k in d["step_kernels"][:16]: print(k["name"][:90], k["ms_per_step"], k["launches_per_step"])
PYEOF
fi
if [[ $WHAT == *pointtest* ]]; then
  ts pointtest; timeout 900 python -m pytest tests/test_gpu_point_ops.py tests/test_gpu_vs_reference_ext.py tests/test_gpu_sa_fused.py -m gpu -q > $OUT/pytest_point.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_point.log
  tail -8 $OUT/pytest_point.log
fi
if [[ $WHAT == *gemmpmc* ]]; then
  ts gemmpmc
  rocprofv3 -L > $OUT/pmc_list.txt 2>&1
  for cfg in nt:0 nt:1 nt:2 nt:5; do
    for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
      tag=$(echo ${cfg}_${grp%% *} | tr ':' '_')
      rm -rf /tmp/pmc_$tag
      (cd /tmp && timeout 120 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_$tag -o p --output-format csv -- python $REPO/tools/gemm_bench.py --only 8 --pmc-loop $cfg > $OUT/pmc_$tag.log 2>&1; echo "pmc $tag exit $?")
      f=$(find /tmp/pmc_$tag -name '*counter_collection.csv' | head -1)
      [ -n "$f" ] && python $REPO/tools/pmc_summary.py "$f" > $OUT/pmc_$tag.txt 2>&1
    done
  done
  tail -n 12 $OUT/pmc_*.txt
fi
if [[ $WHAT == *losstest* ]]; then
  ts losstest; timeout 600 python -m pytest tests/test_gpu_losses.py tests/test_gpu_model.py tests/test_gpu_optim.py tests/test_gpu_gemm.py -m gpu -q -x > $OUT/pytest_loss.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_loss.log; tail -8 $OUT/pytest_loss.log | cut -c1-300
fi
if [[ $WHAT == *attntest* ]]; then
  ts attntest; timeout 600 python -m pytest tests/test_gpu_attention.py tests/test_gpu_model.py tests/test_a16_vs_golden.py -m gpu -q -x > $OUT/pytest_attn.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_attn.log; tail -8 $OUT/pytest_attn.log | cut -c1-300
fi
if [[ $WHAT == *attnpmc* ]]; then
  ts attnpmc
  for cfg in "300 0.1" "130 0.1"; do
    set -- $cfg
    for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
      tag=L$1_${grp%% *}
      rm -rf /tmp/pmc_$tag
      (cd /tmp && timeout 120 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_$tag -o p --output-format csv -- python $REPO/tools/attn_pmc_loop.py --L $1 --p $2 > $OUT/pmc_$tag.log 2>&1; echo "pmc $tag exit $?")
      f=$(find /tmp/pmc_$tag -name '*counter_collection.csv' | head -1)
      [ -n "$f" ] && python $REPO/tools/pmc_summary.py "$f" > $OUT/pmc_$tag.txt 2>&1
    done
  done
  tail -n 14 $OUT/pmc_L*.txt | cut -c1-120
fi
if [[ $WHAT == *alltests* ]]; then
  ts pytest; timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
  grep -E "^(FAILED|ERROR)|passed|failed|exit" $OUT/pytest_gpu.log | head -40
fi
if [[ $WHAT == *splitsweep* ]]; then
  ts splitsweep; timeout 600 python tools/gemm_bench.py --split-sweep --rounds 8 --json $OUT/split_sweep.json > $OUT/split_sweep.log 2>&1; echo "sweep exit $?"; tail -20 $OUT/split_sweep.log | cut -c1-400
fi
if [[ $WHAT == *lntest* ]]; then
  ts lntest; timeout 600 python -m pytest tests/test_gpu_fused_norm.py tests/test_gpu_model.py tests/test_gpu_sa_fused.py tests/test_gpu_gemm.py -m gpu -q -x > $OUT/pytest_ln.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_ln.log; tail -8 $OUT/pytest_ln.log | cut -c1-300
fi
if [[ $WHAT == *attrib* ]]; then
  ts attrib; timeout 600 python tools/step_attrib.py --out $OUT/step_attrib.txt > $OUT/step_attrib.log 2>&1; echo "attrib exit $?"; tail -5 $OUT/step_attrib.log
fi
if [[ $WHAT == *stepbench* ]]; then
  ts bench; timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
  tail -c 3000 $OUT/bench.json; tail -5 $OUT/bench.err
fi
if [[ $WHAT == *benchab* ]]; then
  ts bench-ab
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_native.json 2> $OUT/bench_native.err; echo "bench native exit $?"
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-native-gemm --no-extras > $OUT/bench_lib.json 2> $OUT/bench_lib.err; echo "bench lib exit $?"
  python - <<PYEOF
import json
for n in ("native","lib"):
    try:
        d=json.loads(open("$OUT/bench_%s.json"%n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["roofline"] and (d["roofline"]["kernel"], d["roofline"]["frac"]))
    except Exception as e:
        print(n, "failed", e); print(open("$OUT/bench_%s.err"%n).read()[-1500:])
PYEOF
fi
if [[ $WHAT == *kbench* ]]; then
  ts kernel_bench; timeout 300 python tools/kernel_bench.py --json $OUT/kernel_bench.json > $OUT/kernel_bench.log 2>&1; tail -5 $OUT/kernel_bench.log
fi
if [[ $WHAT == *pmctraffic* ]]; then
  ts pmc
  rm -rf /tmp/pmc && mkdir -p /tmp/pmc
  (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc -o fetch --output-format csv -- python $REPO/tools/pmc_workload.py > $OUT/pmc_fetch.log 2>&1; echo "pmc fetch exit $?")
  (cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc -o write --output-format csv -- python $REPO/tools/pmc_workload.py > $OUT/pmc_write.log 2>&1; echo "pmc write exit $?")
  mkdir -p $OUT/pmc
  f=$(find /tmp/pmc -name 'fetch_counter_collection.csv' | head -1); w=$(find /tmp/pmc -name 'write_counter_collection.csv' | head -1)
  python tools/pmc_traffic.py $f $w $OUT/pmc/pmc_traffic.json | tail -40
fi
if [[ $WHAT == *fullbench* ]]; then
  ts fullbench; timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; tail -c 1500 $OUT/bench.json; tail -3 $OUT/bench.err
  timeout 600 python bench.py --config finetune --steps 5 --warmup 2 > $OUT/bench_finetune.json 2> $OUT/bench_finetune.err; echo "finetune exit $?"; tail -c 800 $OUT/bench_finetune.json | head -c 600; tail -3 $OUT/bench_finetune.err
  timeout 600 python bench.py --config stress --steps 5 --warmup 2 > $OUT/bench_stress.json 2> $OUT/bench_stress.err; echo "stress exit $?"; tail -c 800 $OUT/bench_stress.json | head -c 600; tail -3 $OUT/bench_stress.err
fi
if [[ $WHAT == *prof* ]]; then
  ts rocprof
  rm -rf /tmp/prof && mkdir -p /tmp/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench --output-format csv -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.log 2>&1; echo "rocprof exit $?")
  mkdir -p $OUT/prof
  find /tmp/prof -name '*stats*.csv' -exec cp {} $OUT/prof/ \;
  f=$(ls $OUT/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -40 "$f" | cut -c1-200
  tail -3 $OUT/prof_bench.log
fi
if [[ $WHAT == *attnex* ]]; then
  ts attnex; timeout 900 python -m pytest tests/test_gpu_attention_ex.py tests/test_gpu_attention.py tests/test_a16_vs_golden.py tests/test_gpu_model.py -m gpu -q > $OUT/pytest_attnex.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_attnex.log
  grep -E "^(FAILED|ERROR)|passed|failed|exit" $OUT/pytest_attnex.log | head -40 | cut -c1-400
fi
if [[ $WHAT == *stressfp8* ]]; then
  ts stressfp8
  timeout 600 python bench.py --config stress --steps 5 --warmup 2 --detail $OUT/bench_stress_detail.json > $OUT/bench_stress.json 2> $OUT/bench_stress.err; echo "stress exit $?"; tail -c 1500 $OUT/bench_stress.json; tail -3 $OUT/bench_stress.err
  timeout 600 python bench.py --config stress --fp8 --steps 5 --warmup 2 --detail $OUT/bench_stress_fp8_detail.json > $OUT/bench_stress_fp8.json 2> $OUT/bench_stress_fp8.err; echo "stress fp8 exit $?"; tail -c 1500 $OUT/bench_stress_fp8.json; tail -3 $OUT/bench_stress_fp8.err
fi
ts done; du -sh $REPO/gpurun_out
if [[ $WHAT == *dpgraph* ]]; then
  ts dpgraph
  timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x > $OUT/pytest_dpgraph.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_dpgraph.log
  grep -E "^(FAILED|ERROR)|passed|failed|exit|Error|assert" $OUT/pytest_dpgraph.log | head -30 | cut -c1-400
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --graph-dp > $OUT/bench_dpgraph.json 2> $OUT/bench_dpgraph.err; echo "bench dp exit $?"; tail -c 2200 $OUT/bench_dpgraph.json | head -c 400; echo; tail -3 $OUT/bench_dpgraph.err
fi
if [[ $WHAT == *g8p* ]]; then
  ts g8p
  timeout 600 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x -k "12 or forced" > $OUT/pytest_g8p.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_g8p.log
  tail -8 $OUT/pytest_g8p.log | cut -c1-300
  for i in ${G8P_LAYERS:-8 10 11 4}; do timeout 200 python tools/gemm_bench.py --only $i --rounds 8 >> $OUT/gemm_quick.log 2>&1; done; echo "gemm quick exit $?"; grep "^{" $OUT/gemm_quick.log | cut -c1-1300
fi
if [[ $WHAT == *g8tn* ]]; then
  ts g8tn
  timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_bert_varlen.py -m gpu -q -x -k "wgrad or extent or determin" > $OUT/pytest_g8tn.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_g8tn.log
  tail -8 $OUT/pytest_g8tn.log | cut -c1-400
  for i in ${G8P_LAYERS:-10 11 8 4 6 0}; do timeout 200 python tools/gemm_bench.py --only $i --rounds 8 >> $OUT/gemm_quick.log 2>&1; done; echo "gemm quick exit $?"; grep "^{" $OUT/gemm_quick.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['tokens'], r['in'], r['out'], 'wgrad', r['wgrad_us'], 'fwd12', r['fwd_us']['v12'], 'lib', r['fwd_us']['lib'])"
fi
if [[ $WHAT == *fullcheck* ]]; then
  ts fullcheck
  timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
  grep -E "^(FAILED|ERROR)|passed|failed|exit" $OUT/pytest_gpu.log | head -20 | cut -c1-300
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; tail -c 2200 $OUT/bench.json; echo; tail -2 $OUT/bench.err
fi
if [[ $WHAT == *attrib* ]]; then
  ts attrib
  timeout 600 python tools/step_attrib.py --steps 2 --out $OUT/step_attrib.txt > $OUT/step_attrib.log 2>&1; echo "attrib exit $?"; tail -5 $OUT/step_attrib.log
fi
if [[ $WHAT == *embcheck* ]]; then
  ts embcheck
  timeout 1200 python -m pytest ${EMB_TESTS:-tests/test_gpu_embedding.py tests/test_gpu_bert_varlen.py tests/test_gpu_model.py tests/test_gpu_point_ops.py tests/test_gpu_sa_fused.py} -m gpu -q -x > $OUT/pytest_emb.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_emb.log
  grep -E "^(FAILED|ERROR)|passed|failed|exit|Error|assert " $OUT/pytest_emb.log | head -20 | cut -c1-400
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; head -c 330 $OUT/bench.json; echo; tail -2 $OUT/bench.err
fi
if [[ $WHAT == *dbgdp* ]]; then
  ts dbgdp
  for v in "" "GPS_NO_BF16_ATTR=1" "GPS_NO_POST=1"; do
    echo "== $v"; env $v timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "hip_graph_step_matches" 2>&1 | grep -E "passed|failed|AssertionError: \(" | cut -c1-300
  done
fi
if [[ $WHAT == *stagedprobe* ]]; then
  ts stagedprobe; timeout 300 python tools/probes/${PROBE:-staged_backward_probe.py} 2>&1 | grep -v "amdgpu.ids" | tail -40
fi
if [[ $WHAT == *dbgbench* ]]; then
  ts dbgbench
  for v in "X=1" "GPS_NO_BF16_ATTR=1" "GPS_NO_POST=1"; do
    for extra in "--no-extras" ""; do
      echo "== $v $extra"; env $v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $extra > $OUT/b.json 2> $OUT/b.err; echo "exit $?"; head -c 200 $OUT/b.json; echo; grep -v "amdgpu.ids\|UserWarning\|_warn_once" $OUT/b.err | tail -3
    done
  done
fi
if [[ $WHAT == *poisonprobe* ]]; then
  ts poisonprobe
  for v in "POISON=40" "POISON=40 GPS_POST_TORCH_DPOST=1" "POISON=40 GPS_NO_POST=1"; do
    echo "== $v"; env $v timeout 300 python tools/probes/dp_graph_nan_probe.py 2>&1 | grep -E "^step|NONFINITE|nonfinite params|flat grad" | head -24
  done
fi
if [[ $WHAT == *modes* ]]; then
  ts modes
  for extra in "--no-graph" "--graph-dp"; do
    echo "== $extra"; timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $extra > $OUT/m.json 2> $OUT/m.err; echo "exit $?"; head -c 230 $OUT/m.json; echo; grep -v "amdgpu.ids\|UserWarning\|_warn_once" $OUT/m.err | tail -2
  done
fi
if [[ $WHAT == *smoke* ]]; then
  ts smoke; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -5 $OUT/smoke.log | cut -c1-300
fi
if [[ $WHAT == *postsplit* ]]; then
  ts postsplit
  for v in "POISON=40 FUSE_POST=1 GPS_POST_ONLY=spatial" "POISON=40 FUSE_POST=1 GPS_POST_ONLY=plain" "POISON=40"; do
    echo "== $v"; env $v timeout 300 python tools/probes/dp_graph_nan_probe.py 2>&1 | grep -E "^step" | head -8
  done
fi
if [[ $WHAT == *graddiff* ]]; then
  ts graddiff
  env ${GD_ENV:-X=1} PROBE_B=${GD_B:-64} PROBE_OBJ=${GD_OBJ:-80} timeout 300 python tools/probes/dp_graph_grad_diff_probe.py run /tmp/g_one.pt one 2>&1 | grep -E "^loss"
  env ${GD_ENV:-X=1} PROBE_B=${GD_B:-64} PROBE_OBJ=${GD_OBJ:-80} timeout 300 python tools/probes/dp_graph_grad_diff_probe.py run /tmp/g_dp.pt dp 2>&1 | grep -v "amdgpu.ids" | tail -12
  timeout 300 python tools/probes/dp_graph_grad_diff_probe.py diff /tmp/g_one.pt /tmp/g_dp.pt 2>&1 | tail -8
fi
if [[ $WHAT == *strictalloc* ]]; then
  ts strictalloc
  for v in "X=1" "FUSE_POST=1"; do
    echo "== $v"; env $v PYTORCH_NO_CUDA_MEMORY_CACHING=1 HIP_LAUNCH_BLOCKING=1 AMD_SERIALIZE_KERNEL=3 timeout 400 python tools/probes/strict_alloc_probe.py 2>&1 | grep -v "amdgpu.ids" | tail -14 | cut -c1-300
  done
fi
if [[ $WHAT == *gemmtable* ]]; then
  ts gemmtable; timeout 700 python tools/gemm_bench.py --rounds 6 --json $OUT/gemm_bench.json > $OUT/gemm_bench.log 2>&1; echo "gemm bench exit $?"; grep -c "^{" $OUT/gemm_bench.log
fi
if [[ $WHAT == *extenttest* ]]; then
  ts extenttest; timeout 600 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x -k "device_side_row_extent" 2>&1 | tail -6 | cut -c1-300
fi
