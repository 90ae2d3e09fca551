#!/bin/bash
mkdir -p gpurun_out
for g in 256 512 640 1024 1280 1536 2048 3072 4096 512 1024; do
  SSD_FIRST_GRID=$g timeout -k 10 300 python bench.py --steps 5 --warmup 2 --spinup-seconds 1 --live-traffic off --no-cpu-baseline --no-bf16-leg --layers > gpurun_out/x_b.json 2> gpurun_out/x_l.txt || exit 1
  echo "f32 grid $g: $(grep 'features.0 ' gpurun_out/x_l.txt | head -1 | cut -c60-)"
done
for g in 256 512 1024 2048 4096 100000000; do
SSD_FIRST_GRID=$g timeout -k 10 300 python bench.py --steps 5 --warmup 2 --spinup-seconds 1 --live-traffic off --no-cpu-baseline --conv-dtype bf16 --layers > gpurun_out/x_b.json 2> gpurun_out/x_l.txt
echo "bf16 grid $g: $(grep 'features.0 ' gpurun_out/x_l.txt | head -1 | cut -c60-)"
done
