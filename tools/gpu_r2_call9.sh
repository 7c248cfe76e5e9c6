#!/bin/bash
# round 2, GPU call 9: the whole GPU suite twice (incl. config 4 reference-rule test, config 5, C++ multi-GPU driver)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
for k in 9a 9b; do
timeout 1500 python -m pytest tests -x -q -m gpu --durations=12 -s > $O/pytest_$k.log 2>&1; echo "pytest rc $?" >> $O/pytest_$k.log
grep -E "passed|failed|pytest rc|^iterations|CPU restatement|per-lane kernel mean|tile kernel mean|config 5|multi-GPU driver|frame guard|native:" $O/pytest_$k.log | head -40
done
tail -16 $O/pytest_9a.log
