#!/bin/bash
# round 2, GPU call 2: snapshot records + far-update outbox — parity, bench, kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest_2.log 2>&1; echo "pytest rc $?" >> $O/pytest_2.log
tail -25 $O/pytest_2.log
timeout 120 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_outbox_v1.json 2> $O/bench_outbox_v1.err
timeout 120 python bench.py --steps 28 --warmup 2 --cpu-seconds 0 --stress > $O/bench_outbox_v1_whole.json 2>> $O/bench_outbox_v1.err
cat $O/bench_outbox_v1.json $O/bench_outbox_v1_whole.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_outbox_v1 -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $GRAFT_REPO_ROOT/$O/prof_outbox_v1.log 2>&1)
find $O/prof_outbox_v1 -name "*kernel_stats*" | head -3 | xargs -I{} sh -c 'head -12 {}'
find $O/prof_outbox_v1 -name "*.db" -delete; find $O/prof_outbox_v1 -name "*kernel_trace.csv" -size +20M -delete
