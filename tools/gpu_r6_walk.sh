#!/bin/bash
# A/B of the 256 x 256 kernel's phase walks (gemm_probe = B fragments read inside the MFMA segments, gemm_probe_e1 = the previous schedule)
OUT=$PWD/gpurun_out/r6_walk; mkdir -p $OUT
for b in gemm_probe_e1 gemm_probe; do
  echo "== $b"
  timeout 300 tools/probes/$b bench --variants 7,12 --rounds 5 > $OUT/$b.json 2>/dev/null; python3 tools/gemm_probe_table.py $OUT/$b.json | tail -26
  timeout 200 tools/probes/$b wgrad | head -1
  timeout 200 tools/probes/$b group | tail -10
done
for b in gemm_probe_e1 gemm_probe; do
  echo "== $b"; PROBE=tools/probes/$b bash tools/gpu_r6_trace.sh r6_walk_trace_$b '2 5 3072 768 12608 12' '1 0 12608 768 3072 12' | grep -v "start "
done
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_twin.py tests/test_gpu_joint_compact.py -m gpu -x -q 2>&1 | tail -5
