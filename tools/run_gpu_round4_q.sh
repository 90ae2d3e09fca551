#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "multibox_loss_two_launch" > gpurun_out/q_tests.log 2>&1
rc=$?
tail -5 gpurun_out/q_tests.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 900 python -m pytest tests/test_gpu_path.py -m gpu -q -x -k "loss or match or golden or hard or train_step or graphed" > gpurun_out/q_tests2.log 2>&1
rc=$?
tail -5 gpurun_out/q_tests2.log
if [ $rc -ge 124 ]; then exit $rc; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/q_prof -- python3 $GRAFT_REPO_ROOT/tools/loss_bench.py > $GRAFT_REPO_ROOT/gpurun_out/q_loss.log 2>&1
cd $GRAFT_REPO_ROOT
cat gpurun_out/q_loss.log | tail -8
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/q_prof/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('loss_image', 'finalize', 'match_ce', 'hard_negative', 'best_prior')):
        print(f"{float(r['AverageNs'])/1e3:8.1f} us  x{r['Calls']}  {r['Name'][:90]}")
PY
rm -rf gpurun_out/q_prof
