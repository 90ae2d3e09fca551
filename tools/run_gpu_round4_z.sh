#!/bin/bash
# final records of the round: default bench line, then both profile sets
mkdir -p gpurun_out
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/z_bench.json 2> gpurun_out/z_bench.err
echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/z_bench.json').read().strip().splitlines()[-1]); c = d['config']
print('value', d['value'], 'ms', d['ms_per_step'], 'clock', c['shader_clock_mhz_during_timed_steps'], 'host', c['host_enqueue_ms_per_step'])
print('graph', {k: v for k, v in c.get('graph_step', {}).items() if k != 'note'})
print('f32only', c.get('f32_mfma_only', {}).get('ms_per_step'), 'bf16', c.get('bf16_operand_mode', {}).get('ms_per_step'), c.get('bf16_operand_mode', {}).get('roofline', {}).get('frac'))
r = d['roofline']; print('roofline', r['frac'], r['avg_launch_ms'], r['traffic'], r['traffic_unit'][:60])
print('cpu', {k: v for k, v in d['cpu_baseline'].items() if k in ('value', 'cores')})
PY
timeout -k 10 600 bash tools/profile_round.sh r04 2>&1 | tail -2
timeout -k 10 600 bash tools/profile_bf16.sh r04 2>&1 | tail -2
