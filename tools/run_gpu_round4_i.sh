#!/bin/bash
mkdir -p gpurun_out
export X3_BRIEF=1 X3_FILTER=cat
timeout -k 10 800 bash tools/x3_variants.sh "" "-DX3S_DIRECT_EPI" "-DX3S_NO_SPLIT" "-DX3S_NO_SETPRIO" "-DX3S_FREE_SCHED -DX3S_NO_SETPRIO" "-DX3S_NO_SPLIT -DX3S_DIRECT_EPI" 2>&1 | tee gpurun_out/i_variants.log | grep -v amdgpu.ids
