#!/bin/bash
# final validation, part A: rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes of the bench command, then the
# whole GPU suite twice
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout 90 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_final_check.json 2> $O/bench_final_check.err || { echo "bench failed"; tail -3 $O/bench_final_check.err; exit 1; }
bash tools/profile_bench.sh > $O/profile_bench_final.log 2>&1; tail -4 $O/profile_bench_final.log
for k in 04 05; do
timeout 500 python -m pytest tests -x -q -m gpu --timeout=200 --durations=8 > $O/pytest_gpu_${k}_full_suite.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu_${k}_full_suite.log
grep -E "passed|failed|pytest rc" $O/pytest_gpu_${k}_full_suite.log | tail -2
done
