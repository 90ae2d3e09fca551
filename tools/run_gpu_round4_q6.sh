#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_path.py -m gpu -q -x -k "loss or match or golden or hard or train_step or graphed or ddp or data_parallel" > gpurun_out/q_tests2.log 2>&1
rc=$?
tail -4 gpurun_out/q_tests2.log
if [ $rc -ne 0 ]; then exit $rc; fi
bash tools/run_gpu_round4_q2.sh
