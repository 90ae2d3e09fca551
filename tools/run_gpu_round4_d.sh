#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 700 python bench.py --steps 20 --warmup 5 > gpurun_out/d_bench.json
rc=$?
echo "bench rc=$rc"
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/d_bench.json').read().strip().splitlines()[-1])
    c = d['config']
    print('value', d['value'], 'ms', d['ms_per_step'], 'host', c['host_enqueue_ms_per_step'], c.get('step_launch'))
    print('graph', c.get('graph_step'))
    print('f32only', c.get('f32_mfma_only', {}).get('ms_per_step'), 'bf16', c.get('bf16_operand_mode', {}).get('ms_per_step'))
    r = d['roofline']
    print('roofline', r['frac'], r['kernel'], r['avg_launch_ms'])
    for k, v in r['by_kernel'].items(): print('   ', k, v)
except Exception as e:
    print('parse failed', e)
PY
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 900 bash tools/profile_round.sh r04 2>&1 | tail -5
