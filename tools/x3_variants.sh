#!/bin/bash
# A/B of compile-time variants of csrc/gemm_x3.hip on one device: rebuild with each flag set and run tools/gemm_x3_bench.py.
#   bash tools/x3_variants.sh "" "-DX3_SETPRIO=1" "-DX3_SETPRIO=2"
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for flags in "$@"; do
  touch objectdetection_ssd_amd/csrc/gemm_x3.hip objectdetection_ssd_amd/csrc/gemm_x3v2.hip
  SSD_HIPCC_FLAGS="$flags" python3 -c "from objectdetection_ssd_amd import build; build.build()" > /dev/null
  echo "== flags: [$flags]"
  python3 tools/gemm_x3_bench.py | ${X3_FILTER:-cut -c1-128}
done
touch objectdetection_ssd_amd/csrc/gemm_x3.hip objectdetection_ssd_amd/csrc/gemm_x3v2.hip
python3 -c "from objectdetection_ssd_amd import build; build.build()" > /dev/null
