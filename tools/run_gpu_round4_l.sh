#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q --maxfail=12 -k "conv1_1_written or first_layer_into or pool_gradient_formed or decision_pinned" > gpurun_out/l_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/l_tests.log
grep -E "passed|failed|^FAILED|^E  " gpurun_out/l_tests.log | tail -8
timeout -k 10 300 python tools/ab_step.py engine.first_wino 0 1 2>&1 | tee gpurun_out/l_ab_first.log | tail -2
