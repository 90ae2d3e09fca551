#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "conv1_1" > gpurun_out/ah_tests.log 2>&1
rc=$?
tail -3 gpurun_out/ah_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for m in f32 bf16; do
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --spinup-seconds 1 --live-traffic off --no-cpu-baseline --no-bf16-leg --conv-dtype $m --layers > gpurun_out/ah_b.json 2> gpurun_out/ah_l.txt
echo "$m: $(grep 'wgrad model.features.0 ' gpurun_out/ah_l.txt | cut -c60-)"
done
