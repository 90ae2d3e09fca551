#!/bin/bash
# end of round 4: the full GPU suite, smoke, then the records (default line + both profile sets) on the same device
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=8 > gpurun_out/final_tests.log 2>&1
rc=$?
grep -E "passed|failed|^FAILED|^E  " gpurun_out/final_tests.log | tail -12
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/run_gpu_round4_z.sh
