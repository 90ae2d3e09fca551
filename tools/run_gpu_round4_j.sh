#!/bin/bash
mkdir -p gpurun_out
export X3_BRIEF=1
timeout -k 10 300 python tools/gemm_x3_bench.py 2>&1 | tee gpurun_out/j_gemm.log | grep -v amdgpu.ids
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "plane_gemm_from_three" 2>&1 | tail -2
timeout -k 10 300 python tools/ab_step.py ssd_tune_set_x3_big 0 1 2>&1 | tee gpurun_out/j_ab_big.log | tail -2
