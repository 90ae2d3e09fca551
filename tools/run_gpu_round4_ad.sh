#!/bin/bash
mkdir -p gpurun_out
for ks in 1 4 2 1 4; do
SSD_TN_X3_KS=$ks timeout -k 10 300 python bench.py --steps 10 --warmup 3 --spinup-seconds 2 --live-traffic off --no-cpu-baseline --no-bf16-leg --layers > gpurun_out/ad_b.json 2> gpurun_out/ad_l.txt || exit 1
echo "ks=$ks $(python - <<'PY'
import json
d = json.loads(open('gpurun_out/ad_b.json').read().strip().splitlines()[-1]); c = d['config']
print('step', d['ms_per_step'], c['shader_clock_mhz_during_timed_steps'])
PY
)"
grep -E "^wgrad model.features.(17|19|21|24) " gpurun_out/ad_l.txt | cut -c1-110
done
