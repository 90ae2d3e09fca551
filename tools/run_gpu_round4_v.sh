#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/v_bench.json 2> gpurun_out/v_bench.err
echo "rc=$?"
grep "live traffic\|timed" gpurun_out/v_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/v_bench.json').read().strip().splitlines()[-1]); c = d['config']
print('value', d['value'], 'ms', d['ms_per_step'], 'clock', c['shader_clock_mhz_during_timed_steps'])
r = d['roofline']
print('roofline', r['frac'], r['avg_launch_ms'], r['traffic'], r['traffic_unit'][:400])
b = c.get('bf16_operand_mode', {})
print('bf16', b.get('ms_per_step'), b.get('roofline', {}).get('traffic'), b.get('roofline', {}).get('traffic_unit', '')[:120])
PY
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --conv-dtype bf16 --no-cpu-baseline > gpurun_out/v_bench16.json 2> gpurun_out/v_bench16.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/v_bench16.json').read().strip().splitlines()[-1]); c = d['config']
r = d['roofline']
print('bf16 run', d['ms_per_step'], c['shader_clock_mhz_during_timed_steps'], r['frac'], r['traffic'], r['traffic_unit'][:300])
PY
