#!/bin/bash
mkdir -p gpurun_out
for rep in 1 2; do
for flag in "" "--overlap-wgrad"; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --spinup-seconds 2 --live-traffic off --no-cpu-baseline --conv-dtype bf16 --no-roofline $flag > gpurun_out/al.json 2> gpurun_out/al.err || { tail -3 gpurun_out/al.err; exit 1; }
echo "bf16 '$flag': $(python -c "import json; d=json.loads(open('gpurun_out/al.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['shader_clock_mhz_during_timed_steps'])")"
done
done
for flag in "" "--overlap-wgrad"; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --spinup-seconds 2 --live-traffic off --no-cpu-baseline --no-bf16-leg --no-roofline $flag > gpurun_out/al.json 2> gpurun_out/al.err || { tail -3 gpurun_out/al.err; exit 1; }
echo "f32 '$flag': $(python -c "import json; d=json.loads(open('gpurun_out/al.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['shader_clock_mhz_during_timed_steps'])")"
done
