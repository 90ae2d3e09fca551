#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_path.py -m gpu -q -x -k "bf16 or weight" > gpurun_out/s_tests.log 2>&1
rc=$?
tail -4 gpurun_out/s_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --conv-dtype bf16 --no-cpu-baseline --layers > gpurun_out/s_bench.json 2> gpurun_out/s_layers.txt
python - <<'PY'
import json
d = json.loads(open('gpurun_out/s_bench.json').read().strip().splitlines()[-1]); c = d['config']
print('bf16 step', d['value'], d['ms_per_step'], c['shader_clock_mhz_during_timed_steps'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
PY
grep -E "weight|loss|jobs" gpurun_out/s_layers.txt | head
