#!/bin/bash
bash tools/run_gpu_round4_q4.sh
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "multibox_loss_three_launch" > gpurun_out/q_tests.log 2>&1
grep -v "^img" gpurun_out/q_tests.log | tail -5
