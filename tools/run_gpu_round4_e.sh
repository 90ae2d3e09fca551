#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python tools/ab_graph.py 2>&1 | tee gpurun_out/e_ab_graph.log | tail -3
AB_TWO_STREAMS=1 timeout -k 10 300 python tools/ab_graph.py 2>&1 | tee gpurun_out/e_ab_graph2.log | tail -3
AB_CONV_DTYPE=bf16 timeout -k 10 300 python tools/ab_graph.py 2>&1 | tee gpurun_out/e_ab_graph_bf16.log | tail -3
timeout -k 10 600 bash tools/x3_variants.sh "" "-DX3_PROBE_NO_SPLIT" "-DX3_PROBE_NO_SPLIT -DX3_PROBE_NO_ALOAD" 2>&1 | tee gpurun_out/e_x3_variants.log | grep -E "flags|conv3_2|conv4_2|conv5_2|fc6"
