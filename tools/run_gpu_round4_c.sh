#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q --maxfail=12 -k "second_stream or plane_gemm_from_three or rounding_pinned or adjoint or decision_pinned" -s > gpurun_out/c_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/c_tests.log
grep -E "passed|failed|rounding-pinned|adjoint dgrad|decision-pinned" gpurun_out/c_tests.log | tail -14
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 200 python tools/gemm_x3_bench.py 2>&1 | tee gpurun_out/c_gemm.log | tail -8
timeout -k 10 700 python bench.py --steps 20 --warmup 5 > gpurun_out/c_bench.json
rc=$?
echo "bench rc=$rc"
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/c_bench.json').read().strip().splitlines()[-1])
    c = d['config']
    print('value', d['value'], 'ms', d['ms_per_step'], 'host', c['host_enqueue_ms_per_step'], c.get('step_launch'))
    print('f32only', c.get('f32_mfma_only', {}).get('ms_per_step'), 'bf16', c.get('bf16_operand_mode', {}).get('ms_per_step'))
    print('roofline', d['roofline']['frac'], d['roofline']['kernel'])
    print('cpu', {k: v for k, v in d['cpu_baseline'].items() if k in ('value', 'cores', 'by_threads', 'all_host_cores_leg')})
    print('loss', d.get('loss_delta_vs_cpu'))
except Exception as e:
    print('parse failed', e)
PY
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python tools/ab_step.py engine.adjoint_dgrad 0 1 2>&1 | tee gpurun_out/c_ab_adj.log | tail -3
timeout -k 10 300 python tools/ab_step.py engine.adjoint_chain 0 1 2>&1 | tee gpurun_out/c_ab_chain.log | tail -3
timeout -k 10 300 python tools/ab_step.py ssd_tune_set_x3_mfma 32 16 2>&1 | tee gpurun_out/c_ab_m16.log | tail -3
