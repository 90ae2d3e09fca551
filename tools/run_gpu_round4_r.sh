#!/bin/bash
mkdir -p gpurun_out
for gap in 2 1 2 1; do
echo "== gap $gap"
SSD_CONV_BF16_GAP=$gap timeout -k 10 300 python tools/conv_bf16_bench.py 32 2>&1 | grep -E "conv4|conv5|c_4|c_7|total" || exit 1
done
