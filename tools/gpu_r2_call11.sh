#!/bin/bash
# round 2, GPU call 11: whole GPU suite (relax 0.5, snapshot per launch while warm), bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu --durations=12 -s > $O/pytest_11.log 2>&1; echo "pytest rc $?" >> $O/pytest_11.log
grep -E "passed|failed|pytest rc|^iterations|CPU restatement|per-lane kernel mean|tile kernel mean|config 5|multi-GPU driver|frame guard|native:" $O/pytest_11.log | head -40
tail -16 $O/pytest_11.log
show() { python - "$1" <<'PY'
import json, sys
try:
    j=json.load(open(sys.argv[1])); r=j["roofline"]
    print(sys.argv[1].split("/")[-1], "ms/step", round(j["ms_per_step"],2), "frac", round(r["frac"],3), "all", round(r["frac_all_kernels"],3), "aux", {k: round(v,2) for k,v in r["aux_kernels_ms_per_step"].items()}, {k:(round(v["update_kernel_ms_per_step"],2), round(v["frac_all_kernels"],3)) for k,v in r["phases"].items()}, j.get("stress_sampled"))
except Exception as e: print(sys.argv[1], "failed", e)
PY
}
timeout 200 python bench.py --steps 20 --warmup 5 > $O/bench_v7.json 2> $O/bench_v7.err; show $O/bench_v7.json
timeout 120 python bench.py --steps 28 --warmup 2 --cpu-seconds 0 --stress > $O/bench_v7_whole.json 2> $O/bench_v7_whole.err; show $O/bench_v7_whole.json
