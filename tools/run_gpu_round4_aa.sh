#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "wgrad or weight_gradient" > gpurun_out/aa_tests.log 2>&1
rc=$?
tail -3 gpurun_out/aa_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --spinup-seconds 2 --live-traffic off --no-cpu-baseline --conv-dtype bf16 --layers > gpurun_out/aa_b.json 2> gpurun_out/aa_l.txt
grep -E "^wgrad (c_4|c_7|conv_fc7|model.features.28|model.features.21) " gpurun_out/aa_l.txt
python - <<'PY'
import json
d = json.loads(open('gpurun_out/aa_b.json').read().strip().splitlines()[-1]); c = d['config']
print('bf16 step', d['value'], d['ms_per_step'], c['shader_clock_mhz_during_timed_steps'])
PY
