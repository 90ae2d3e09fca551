#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_path.py -m gpu -q -x -k "bf16" > gpurun_out/af_tests.log 2>&1
rc=$?
tail -3 gpurun_out/af_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/conv_bf16_bench.py 32 2>&1 | grep -E "c_7|c_4|conv5|total"
