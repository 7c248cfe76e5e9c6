#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout 600 python -m pytest tests -x -q -m gpu -k "one_workgroup or outbox or unsorted or million" > $O/pytest_15.log 2>&1; echo "pytest rc $?" >> $O/pytest_15.log; tail -3 $O/pytest_15.log
show() { python - "$1" <<'PY'
import json, sys
try:
    j=json.load(open(sys.argv[1])); r=j["roofline"]
    print(sys.argv[1].split("/")[-1], "ms/step", round(j["ms_per_step"],2), "frac", round(r["frac"],3), "all", round(r["frac_all_kernels"],3), "aux", {k: round(v,2) for k,v in r["aux_kernels_ms_per_step"].items()}, {k:(round(v["update_kernel_ms_per_step"],2), round(v["frac_all_kernels"],3)) for k,v in r["phases"].items()}, j.get("stress_sampled"))
except Exception as e: print(sys.argv[1], "failed", e)
PY
}
timeout 120 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_v11.json 2> $O/bench_v11.err; show $O/bench_v11.json
