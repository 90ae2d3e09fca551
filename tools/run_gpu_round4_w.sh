#!/bin/bash
# two ranks on the one GPU over gloo: the N > 1 code path of bench.py with the spin-up phase (one decision for all ranks), f32 and bf16 payload
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --one-device --steps 4 --warmup 2 --spinup-seconds 1.5 --no-cpu-baseline --no-roofline > gpurun_out/w_dp2.json 2> gpurun_out/w_dp2.err
echo "dp2 rc=$?"; grep "bench +" gpurun_out/w_dp2.err | tail -4
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --one-device --steps 4 --warmup 2 --spinup-seconds 1.5 --no-cpu-baseline --no-roofline --conv-dtype bf16 > gpurun_out/w_dp2_bf16.json 2> gpurun_out/w_dp2_bf16.err
echo "dp2 bf16 rc=$?"
python - <<'PY'
import json
for f in ('gpurun_out/w_dp2.json', 'gpurun_out/w_dp2_bf16.json'):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['n_gpus'], d['value'], d['ms_per_step'], d['config']['spinup']['untimed_steps'], d['config']['gradient_exchange'].get('exposed_ms_per_step'))
    except Exception as e:
        print(f, 'parse failed', e)
PY
