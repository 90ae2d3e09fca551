#!/bin/bash
# kernel traces (grid sizes, LDS, registers per dispatch) of one step in both modes, for tools/grid_audit.py
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for mode in f32 bf16; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$mode -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --spinup-seconds 0 --live-traffic off --no-cpu-baseline --no-bf16-leg --no-overlap-tail --no-roofline --graph-step off --conv-dtype $mode > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/ab_$mode.err || exit 1
  cp $(ls /tmp/trace_$mode/*/*kernel_trace.csv | head -1) $GRAFT_REPO_ROOT/gpurun_out/ab_trace_$mode.csv
  ls -la $GRAFT_REPO_ROOT/gpurun_out/ab_trace_$mode.csv
done
