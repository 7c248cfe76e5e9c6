#!/bin/bash
# final validation, part B: the driver's bench command (with the CPU baseline leg), the whole schedule with stress,
# smoke(), and the whole GPU suite three more times
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc $?"; head -c 400 $O/bench_final.json; echo
timeout 120 python bench.py --steps 30 --warmup 0 --cpu-seconds 0 --stress > $O/bench_final_whole.json 2> $O/bench_final_whole.err; echo "whole rc $?"
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke_final.log 2>&1; tail -2 $O/smoke_final.log
for k in 06 07 08; do
timeout 500 python -m pytest tests -x -q -m gpu --timeout=200 --durations=8 > $O/pytest_gpu_${k}_full_suite.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu_${k}_full_suite.log
grep -E "passed|failed|pytest rc" $O/pytest_gpu_${k}_full_suite.log | tail -2
done
