#!/bin/bash
mkdir -p gpurun_out
python - > gpurun_out/q_stamps.log 2>&1 <<'PY'
import sys, os
sys.argv = ['x']
exec(open('tools/loss_bench.py').read().replace('range(200)', 'range(2)').replace('range(10)', 'range(1)').replace('range(3)', 'range(1)'))
PY
grep "^img" gpurun_out/q_stamps.log | tail -8
