#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q --maxfail=8 > gpurun_out/y_tests.log 2>&1
rc=$?
grep -E "passed|failed|^FAILED|^E  " gpurun_out/y_tests.log | tail -12
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
