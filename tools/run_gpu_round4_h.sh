#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "plane_gemm_from_three" > gpurun_out/h_tests.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/h_tests.log
timeout -k 10 300 python tools/gemm_x3_bench.py 2>&1 | tee gpurun_out/h_gemm.log | grep -v amdgpu.ids | sed -e 's/  f32 .*x3 0/ x3 0/' | cut -c1-520
