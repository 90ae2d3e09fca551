#!/bin/bash
# A/B of compile-time variants of csrc/conv_bf16.hip on one device: rebuild with each flag set, run tools/conv_bf16_bench.py and (last) the step.
#   bash tools/cb_variants.sh "" "-DCB_RING4"
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for flags in "$@"; do
  touch objectdetection_ssd_amd/csrc/conv_bf16.hip
  SSD_HIPCC_FLAGS="$flags" python3 -c "from objectdetection_ssd_amd import build; build.build()" > /dev/null
  echo "== flags: [$flags]"
  python3 tools/conv_bf16_bench.py 2>&1 | grep -v amdgpu.ids
  python3 bench.py --conv-dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['ms_per_step'], 'ms at', d['config']['shader_clock_mhz_during_timed_steps'], 'MHz')"
done
touch objectdetection_ssd_amd/csrc/conv_bf16.hip
python3 -c "from objectdetection_ssd_amd import build; build.build()" > /dev/null
