#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke_final3.log 2>&1; tail -2 $O/smoke_final3.log
timeout 500 python -m pytest tests -x -q -m gpu --timeout=200 --durations=8 > $O/pytest_gpu_11_full_suite.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu_11_full_suite.log
grep -E "passed|failed|pytest rc" $O/pytest_gpu_11_full_suite.log | tail -2
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_final3.json 2> $O/bench_final3.err; echo "bench rc $?"; head -c 200 $O/bench_final3.json; echo
