#!/bin/bash
mkdir -p gpurun_out
for f7 in 0 1 0 1; do
echo "== flat7 $f7"
SSD_CONV_BF16_FLAT7=$f7 timeout -k 10 300 python tools/conv_bf16_bench.py 32 2>&1 | grep -E "conv3|total" || exit 1
done
