#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "multibox_loss_three_launch" > gpurun_out/q_tests.log 2>&1
rc=$?
tail -5 gpurun_out/q_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
bash tools/run_gpu_round4_q2.sh
