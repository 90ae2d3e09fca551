#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=12 > gpurun_out/f_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/f_tests.log
grep -E "passed|failed|^FAILED" gpurun_out/f_tests.log | tail -8
if [ $rc -ge 124 ]; then exit $rc; fi
# two ranks on the one GPU over gloo: the N > 1 code path of bench.py (default overlapped exchange, exposed-time events), f32 and bf16 payload
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --one-device --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/f_dp2.json 2> gpurun_out/f_dp2.err
echo "dp2 rc=$?"; tail -c 400 gpurun_out/f_dp2.err
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --one-device --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --conv-dtype bf16 > gpurun_out/f_dp2_bf16.json 2> gpurun_out/f_dp2_bf16.err
echo "dp2 bf16 rc=$?"; tail -c 400 gpurun_out/f_dp2_bf16.err
python - <<'PY'
import json
for f in ('gpurun_out/f_dp2.json', 'gpurun_out/f_dp2_bf16.json'):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['n_gpus'], d['value'], d['ms_per_step'], d['config']['parallelism'], d['config']['gradient_exchange'])
    except Exception as e:
        print(f, 'parse failed', e)
PY
