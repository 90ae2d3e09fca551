#!/bin/bash
# Kernel stats of the decode workload at batch 32 and for a single image:
#   bash tools/profile_decode.sh r03   -> gpurun_out/prof_r03_decode/{b32_kernel_stats.csv,b32_bench.json,b1_kernel_stats.csv,b1_bench.json}
set -e
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_decode
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for B in 32 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s$B -- python3 $ROOT/bench.py --workload decode --batch $B --steps 50 --warmup 5 --no-cpu-baseline > $OUT/b${B}_bench.json 2> $OUT/b${B}.err
  cp $(ls $OUT/s$B/*/*kernel_stats.csv | head -1) $OUT/b${B}_kernel_stats.csv
  rm -rf $OUT/s$B
done
python3 $ROOT/bench.py --workload decode --batch 32 --steps 50 --warmup 5 > $OUT/b32_bench_plain.json 2> $OUT/b32_plain.err
ls -la $OUT
