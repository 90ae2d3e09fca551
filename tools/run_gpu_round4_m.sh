#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-bf16-leg --layers > gpurun_out/m_bench.json 2> gpurun_out/m_layers.txt
grep -E "features\.0 |features\.2 |features\.5 " gpurun_out/m_layers.txt
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-bf16-leg --layers --no-first-wino > gpurun_out/m_bench0.json 2> gpurun_out/m_layers0.txt
grep -E "features\.0 |features\.2 |features\.5 " gpurun_out/m_layers0.txt
