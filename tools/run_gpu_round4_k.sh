#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=12 > gpurun_out/k_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/k_tests.log
grep -E "passed|failed|^FAILED|^E  " gpurun_out/k_tests.log | tail -12
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python tools/ab_step.py engine.first_wino 0 1 2>&1 | tee gpurun_out/k_ab_first.log | tail -2
