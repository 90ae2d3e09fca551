#!/bin/bash
# bf16-operand mode (BASELINE configs[2] per-GPU leg) under the same evidence as the f32 headline:
#   bash tools/profile_bf16.sh r03   -> gpurun_out/prof_r03_bf16/{kernel_stats.csv,bench.json,layers.txt,pmc_p1.txt,pmc_p2.txt,pmc_p3.txt}
# Every run passes --no-overlap-tail: all kernels on one stream, so a kernel's duration / counters are its own.
set -e
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_bf16
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--conv-dtype bf16 --live-traffic off --no-cpu-baseline --no-overlap-tail"
python3 $ROOT/bench.py --steps 10 --warmup 3 --conv-dtype bf16 --no-cpu-baseline --layers > $OUT/bench_plain.json 2> $OUT/layers.txt
echo "plain done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 10 --warmup 3 $ARGS > $OUT/bench.json 2> $OUT/bench.err
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
echo "stats done"
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $P1 --output-format csv -d $OUT/p1 -- python3 $ROOT/bench.py --steps 1 --warmup 1 --spinup-seconds 0 $ARGS > /dev/null 2> $OUT/p1.err
echo "p1 done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p2 -- python3 $ROOT/bench.py --steps 1 --warmup 1 --spinup-seconds 0 $ARGS > /dev/null 2> $OUT/p2.err
echo "p2 done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p3 -- python3 $ROOT/bench.py --steps 1 --warmup 1 --spinup-seconds 0 $ARGS > /dev/null 2> $OUT/p3.err
echo "p3 done"
for p in p1 p2 p3; do
  f=$(ls $OUT/$p/*/*counter_collection.csv | head -1)
  python3 $ROOT/tools/pmc_summary.py ${f%_counter_collection.csv} > $OUT/pmc_$p.txt
done
rm -rf $OUT/stats $OUT/p1 $OUT/p2 $OUT/p3
ls -la $OUT
