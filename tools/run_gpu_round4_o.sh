#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=12 > gpurun_out/o_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/o_tests.log
grep -E "passed|failed|^FAILED|^E  " gpurun_out/o_tests.log | tail -8
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/o_bench.json
rc=$?
echo "bench rc=$rc"
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/o_bench.json').read().strip().splitlines()[-1])
    c = d['config']
    print('value', d['value'], 'ms', d['ms_per_step'], 'clock', c['shader_clock_mhz_during_timed_steps'], 'host', c['host_enqueue_ms_per_step'])
    print('graph', {k: v for k, v in c.get('graph_step', {}).items() if k != 'note'})
    print('f32only', c.get('f32_mfma_only', {}).get('ms_per_step'), 'bf16', c.get('bf16_operand_mode', {}).get('ms_per_step'))
    r = d['roofline']
    print('roofline', r['frac'], r['kernel'], r['avg_launch_ms'], r['traffic'])
    print('cpu', {k: v for k, v in d['cpu_baseline'].items() if k in ('value', 'cores', 'all_host_cores_leg')})
    print('loss', d.get('loss_delta_vs_cpu', {}).get('within_tolerance'))
except Exception as e:
    print('parse failed', e)
PY
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 900 bash tools/profile_round.sh r04 2>&1 | tail -3
