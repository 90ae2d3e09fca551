#!/bin/bash
# Where the persistent K = 64 kernel (csrc/conv_bf16.hip conv3x3_bf16_k64_kernel) spends a patch: rebuilds the file with its stores and / or
# its halo DMA compiled out (-DK64_NO_STORE / -DK64_NO_DMA: results are then wrong, times only) and times conv1_2 at batch 32.
set -e
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for flags in "" "-DK64_NO_STORE" "-DK64_NO_DMA" "-DK64_NO_STORE -DK64_NO_DMA" ""; do
  touch objectdetection_ssd_amd/csrc/conv_bf16.hip
  SSD_HIPCC_FLAGS="$flags" python -c "from objectdetection_ssd_amd import build; build.build()"
  echo "flags: $flags"
  python tools/conv_bf16_bench.py 32 -1 -1 1 2>&1 | grep "conv1_2"
done
