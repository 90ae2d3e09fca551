#!/bin/bash
# Where nms_kernel's time goes: rebuild csrc/nms.hip with one phase compiled out at a time (results are then meaningless) and time the decode.
#   bash tools/nms_probe.sh       (GPU box)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for flags in "" "-DNMS_PROBE_NO_A" "-DNMS_PROBE_NO_B" "-DNMS_PROBE_NO_C" "-DNMS_PROBE_NO_A -DNMS_PROBE_NO_B -DNMS_PROBE_NO_C"; do
  touch objectdetection_ssd_amd/csrc/nms.hip
  SSD_HIPCC_FLAGS="$flags" python3 -c "from objectdetection_ssd_amd import build; build.build()" > /dev/null
  echo "== flags: [$flags]"
  python3 tools/decode_bench.py | grep batch
done
touch objectdetection_ssd_amd/csrc/nms.hip
python3 -c "from objectdetection_ssd_amd import build; build.build()" > /dev/null
