#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/q_prof -- python3 $GRAFT_REPO_ROOT/tools/loss_bench.py > $GRAFT_REPO_ROOT/gpurun_out/q_loss.log 2>&1 || exit 1
grep "^round" $GRAFT_REPO_ROOT/gpurun_out/q_loss.log
python3 - <<'PY'
import csv, glob, sys, os
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/q_prof/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('loss_', 'finalize', 'match_ce', 'hard_negative', 'best_prior')):
        print(f"{float(r['AverageNs'])/1e3:8.1f} us  x{r['Calls']}  {r['Name'][:60]}")
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/q_prof
