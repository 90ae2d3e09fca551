#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_path.py -m gpu -q -x -k "bf16 or wgrad or weight_gradient" > gpurun_out/ai_tests.log 2>&1
rc=$?
tail -3 gpurun_out/ai_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --spinup-seconds 2 --live-traffic off --no-cpu-baseline --conv-dtype bf16 --layers > gpurun_out/ai_b.json 2> gpurun_out/ai_l.txt
grep -E "^wgrad model.features.(2|7|14|21|28) |^wgrad c_4" gpurun_out/ai_l.txt | cut -c1-110
python - <<'PY'
import json
d = json.loads(open('gpurun_out/ai_b.json').read().strip().splitlines()[-1]); c = d['config']
print('bf16 step', d['value'], d['ms_per_step'], c['shader_clock_mhz_during_timed_steps'])
PY
