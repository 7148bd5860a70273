#!/bin/bash
OUT=$PWD/gpurun_out/r6i; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_twin.py tests/test_gpu_graph_chain.py tests/test_gpu_model.py tests/test_gpu_point_ops.py::test_library_is_the_hip_one -m gpu -q 2>&1 | tail -15
