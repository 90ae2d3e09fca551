#!/bin/bash
# Profiles of one round on the GPU box: kernel stats of the default bench run + three PMC passes (counters in their own runs).
#   bash tools/profile_round.sh r02      -> gpurun_out/prof_r02/{kernel_stats.csv,bench.json,pmc_p1.txt,pmc_p2.txt,pmc_p3.txt}
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# --no-overlap-tail: every kernel on ONE stream, so a kernel duration (and its counters) in the stats is its own rate (with the tail group on its second stream those kernels overlap others)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 10 --warmup 3 --live-traffic off --no-cpu-baseline --no-bf16-leg --no-overlap-tail > $OUT/bench.json 2> $OUT/bench.err
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
echo "stats done"
python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-leg --layers > $OUT/bench_plain.json 2> $OUT/layers.txt
echo "plain + layers done"
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $OUT/p1 -- python3 $ROOT/bench.py --steps 1 --warmup 1 --spinup-seconds 0 --live-traffic off --no-cpu-baseline --no-bf16-leg --no-overlap-tail > /dev/null 2> $OUT/p1.err
echo "p1 done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p2 -- python3 $ROOT/bench.py --steps 1 --warmup 1 --spinup-seconds 0 --live-traffic off --no-cpu-baseline --no-bf16-leg --no-overlap-tail > /dev/null 2> $OUT/p2.err
echo "p2 done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p3 -- python3 $ROOT/bench.py --steps 1 --warmup 1 --spinup-seconds 0 --live-traffic off --no-cpu-baseline --no-bf16-leg --no-overlap-tail > /dev/null 2> $OUT/p3.err
echo "p3 done"
for p in p1 p2 p3; do
  f=$(ls $OUT/$p/*/*counter_collection.csv | head -1)
  python3 $ROOT/tools/pmc_summary.py ${f%_counter_collection.csv} > $OUT/pmc_$p.txt
done
rm -rf $OUT/stats $OUT/p1 $OUT/p2 $OUT/p3
ls -la $OUT
