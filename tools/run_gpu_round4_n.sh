#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -q --maxfail=8 -k "bf16" > gpurun_out/n_tests.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED" gpurun_out/n_tests.log | tail -5
timeout -k 10 800 bash tools/cb_variants.sh "" "-DCB_RING3" "" "-DCB_RING3" 2>&1 | tee gpurun_out/n_cb.log | grep -E "flags|total|step"
