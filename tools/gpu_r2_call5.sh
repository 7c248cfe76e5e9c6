#!/bin/bash
# round 2, GPU call 5: tile kernel v3 (rings for the window's buckets) — parity, bench, no-message bound, SQ counters
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout 600 python -m pytest tests -x -q -m gpu --durations=4 -k "one_workgroup or terms_bit_exact or million or outbox or unsorted or tandem" > $O/pytest_5.log 2>&1; echo "pytest rc $?" >> $O/pytest_5.log
tail -8 $O/pytest_5.log
show() { python - "$1" <<'PY'
import json, sys
try:
    j=json.load(open(sys.argv[1])); r=j["roofline"]
    print(sys.argv[1].split("/")[-1], "ms/step", round(j["ms_per_step"],2), "frac", round(r["frac"],3), "all", round(r["frac_all_kernels"],3), "aux", {k: round(v,2) for k,v in r["aux_kernels_ms_per_step"].items()}, {k:(round(v["update_kernel_ms_per_step"],2), round(v["frac_all_kernels"],3)) for k,v in r["phases"].items()}, j.get("stress_sampled"))
except Exception as e: print(sys.argv[1], "failed", e)
PY
}
timeout 120 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --stress > $O/bench_v3.json 2> $O/bench_v3.err; show $O/bench_v3.json
PGSGD_TILE_EXP=1 timeout 120 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_v3_nomsg.json 2> $O/bench_v3_nomsg.err; show $O/bench_v3_nomsg.json
SKIP_TCC=1 bash tools/profile_sq.sh v3 > $O/profile_sq_v3.log 2>&1; tail -6 $O/profile_sq_v3.log
cp gpurun_out/prof_sq_v3/sq_tcc_summary.json $O/sq_summary_v3.json
