#!/bin/bash
# round 2, GPU call 14: the whole GPU suite twice in a row, bench (driver flags), rocprofv3 kernel trace + PMC passes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
for k in 14a 14b; do
timeout 1500 python -m pytest tests -x -q -m gpu --durations=10 > $O/pytest_$k.log 2>&1; echo "pytest rc $?" >> $O/pytest_$k.log
tail -3 $O/pytest_$k.log
done
grep -E "^[0-9.]+s call" $O/pytest_14a.log | head -12
show() { python - "$1" <<'PY'
import json, sys
try:
    j=json.load(open(sys.argv[1])); r=j["roofline"]
    print(sys.argv[1].split("/")[-1], "ms/step", round(j["ms_per_step"],2), "frac", round(r["frac"],3), "all", round(r["frac_all_kernels"],3), "aux", {k: round(v,2) for k,v in r["aux_kernels_ms_per_step"].items()}, {k:(round(v["update_kernel_ms_per_step"],2), round(v["frac_all_kernels"],3)) for k,v in r["phases"].items()}, j.get("stress_sampled"))
except Exception as e: print(sys.argv[1], "failed", e)
PY
}
timeout 200 python bench.py --steps 20 --warmup 5 > $O/bench_v8.json 2> $O/bench_v8.err; show $O/bench_v8.json
timeout 120 python bench.py --steps 28 --warmup 2 --cpu-seconds 0 --stress > $O/bench_v8_whole.json 2> $O/bench_v8_whole.err; show $O/bench_v8_whole.json
bash tools/profile_bench.sh > $O/profile_bench_v8.log 2>&1
python tools/summarize_prof.py gpurun_out/prof $O v8 > $O/summarize_v8.log 2>&1; head -14 $O/rocprof_kernel_stats_v8.csv | cut -c1-200
