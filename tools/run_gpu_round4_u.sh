#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q --maxfail=8 -k "bf16" > gpurun_out/u_tests.log 2>&1
rc=$?
grep -E "passed|failed|^FAILED|^E  " gpurun_out/u_tests.log | tail -12
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/u_bench.json 2> gpurun_out/u_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/u_bench.json').read().strip().splitlines()[-1]); c = d['config']
print('value', d['value'], 'ms', d['ms_per_step'], 'clock', c['shader_clock_mhz_during_timed_steps'], 'host', c['host_enqueue_ms_per_step'])
print('graph', {k: v for k, v in c.get('graph_step', {}).items() if k != 'note'})
print('f32only', c.get('f32_mfma_only', {}).get('ms_per_step'), 'bf16', c.get('bf16_operand_mode', {}).get('ms_per_step'), c.get('bf16_operand_mode', {}).get('roofline', {}).get('frac'))
print('loss', d.get('loss_delta_vs_cpu'))
PY
