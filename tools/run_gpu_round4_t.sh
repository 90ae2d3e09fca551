#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q --maxfail=8 > gpurun_out/t_tests.log 2>&1
rc=$?
grep -E "passed|failed|^FAILED|^E  " gpurun_out/t_tests.log | tail -12
exit $rc
