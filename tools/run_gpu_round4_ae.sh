#!/bin/bash
for bn in -1 64 -1 64; do
echo "== bn $bn"
timeout -k 10 300 python tools/conv_bf16_bench.py 32 -1 $bn 2>&1 | grep -E "c_7|c_4|conv5" || exit 1
done
