#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
for k in 12 13 14; do
timeout 500 python -m pytest tests -x -q -m gpu --timeout=200 --durations=5 > $O/pytest_gpu_${k}_full_suite.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu_${k}_full_suite.log
grep -E "passed|failed|pytest rc" $O/pytest_gpu_${k}_full_suite.log | tail -2
done
