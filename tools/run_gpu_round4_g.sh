#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python tools/gemm_x3_bench.py 2>&1 | tee gpurun_out/g_gemm.log | grep -v amdgpu.ids | cut -c1-420
