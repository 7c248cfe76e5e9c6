#!/bin/bash
# round 2, GPU call 3: write-granularity microbench, occupancy sensitivity, parity after the pool-size fix
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
MICROBENCH_R2B=1 timeout 300 odgi_amd/lib/microbench > $O/microbench_r2b.jsonl 2>&1; cat $O/microbench_r2b.jsonl
timeout 900 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest_3.log 2>&1; echo "pytest rc $?" >> $O/pytest_3.log
tail -15 $O/pytest_3.log
for gridsz in 1024 1536; do
  PGSGD_TILE_GRID=$gridsz timeout 120 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_outbox_v1_grid$gridsz.json 2> $O/bench_outbox_v1_grid$gridsz.err
  python - <<PY
import json
j=json.load(open("$O/bench_outbox_v1_grid$gridsz.json"))
print("grid $gridsz", j["ms_per_step"], j["roofline"]["phases"])
PY
done
