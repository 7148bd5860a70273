#!/bin/bash
# A/B of the 256 x 256 kernel's phase walks (gemm_probe = row-half-major, gemm_probe_e1 = quadrant walk of rounds 3 - 5)
OUT=$PWD/gpurun_out/r6_walk; mkdir -p $OUT
for b in gemm_probe_e1 gemm_probe; do
  echo "== $b"
  timeout 300 tools/probes/$b bench --variants 7,12 --rounds 5 > $OUT/$b.json 2>/dev/null; python3 tools/gemm_probe_table.py $OUT/$b.json | tail -26
  timeout 200 tools/probes/$b wgrad | head -1
  timeout 200 tools/probes/$b group | tail -10
done
bash tools/gpu_r6_trace.sh r6_walk_trace '2 5 3072 768 12608 12' '1 0 12608 768 3072 12' '0 0 12608 3072 768 12' | grep -v "start "
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_twin.py -m gpu -x -q 2>&1 | tail -3
