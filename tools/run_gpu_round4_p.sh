#!/bin/bash
# the default bench line with the spin-up phase, twice (does the first timed window now agree with the later legs?), then the profiles of the round
mkdir -p gpurun_out
for i in 1 2; do
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/p_bench$i.json
rc=$?
echo "bench$i rc=$rc"
python - $i <<'PY'
import json, sys
try:
    d = json.loads(open(f'gpurun_out/p_bench{sys.argv[1]}.json').read().strip().splitlines()[-1])
    c = d['config']
    print('value', d['value'], 'ms', d['ms_per_step'], 'clock', c['shader_clock_mhz_during_timed_steps'], 'host', c['host_enqueue_ms_per_step'], 'spin', c['spinup']['untimed_steps'])
    print('graph', {k: v for k, v in c.get('graph_step', {}).items() if k != 'note'})
    print('f32only', c.get('f32_mfma_only', {}).get('ms_per_step'), 'bf16', c.get('bf16_operand_mode', {}).get('ms_per_step'))
    r = d['roofline']
    print('roofline', r['frac'], r['avg_launch_ms'])
except Exception as e:
    print('parse failed', e)
PY
if [ $rc -ge 124 ]; then exit $rc; fi
done
timeout -k 10 900 bash tools/profile_round.sh r04 2>&1 | tail -3
