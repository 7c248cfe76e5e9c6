#!/bin/bash
# round 2, GPU call 6: tile kernel v4 (unified rings, COOLING/LOCAL templates, trimmed sampler) — parity, bench, counters
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout 600 python -m pytest tests -x -q -m gpu --durations=4 -k "one_workgroup or terms_bit_exact or million or outbox or unsorted or tandem or frame" > $O/pytest_6.log 2>&1; echo "pytest rc $?" >> $O/pytest_6.log
tail -8 $O/pytest_6.log
show() { python - "$1" <<'PY'
import json, sys
try:
    j=json.load(open(sys.argv[1])); r=j["roofline"]
    print(sys.argv[1].split("/")[-1], "ms/step", round(j["ms_per_step"],2), "frac", round(r["frac"],3), "all", round(r["frac_all_kernels"],3), "aux", {k: round(v,2) for k,v in r["aux_kernels_ms_per_step"].items()}, {k:(round(v["update_kernel_ms_per_step"],2), round(v["frac_all_kernels"],3)) for k,v in r["phases"].items()}, j.get("stress_sampled"))
except Exception as e: print(sys.argv[1], "failed", e)
PY
}
timeout 120 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_v4.json 2> $O/bench_v4.err; show $O/bench_v4.json
PGSGD_TILE_EXP=1 timeout 120 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_v4_nomsg.json 2> $O/bench_v4_nomsg.err; show $O/bench_v4_nomsg.json
timeout 120 python bench.py --steps 28 --warmup 2 --cpu-seconds 0 --stress > $O/bench_v4_whole.json 2> $O/bench_v4_whole.err; show $O/bench_v4_whole.json
SKIP_TCC=1 bash tools/profile_sq.sh v4 > $O/profile_sq_v4.log 2>&1; tail -4 $O/profile_sq_v4.log
cp gpurun_out/prof_sq_v4/sq_tcc_summary.json $O/sq_summary_v4.json
