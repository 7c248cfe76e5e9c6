#!/bin/bash
# round 2, GPU call 1: the reordered suite, the baseline bench, tile-kernel ablations, machine ceilings for the new far path
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu --durations=25 > $O/pytest_1.log 2>&1; echo "pytest rc $?" >> $O/pytest_1.log
timeout 120 python bench.py --steps 20 --warmup 5 > $O/bench_base.json 2> $O/bench_base.err
for a in 1 2 3 4; do
  PGSGD_TILE_ABLATE=$a timeout 120 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_abl$a.json 2> $O/bench_abl$a.err
done
MICROBENCH_R2=1 timeout 300 odgi_amd/lib/microbench > $O/microbench_r2.jsonl 2>&1
tail -3 $O/pytest_1.log; cat $O/bench_abl*.json | cut -c1-400; cat $O/microbench_r2.jsonl
