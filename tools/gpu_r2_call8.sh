#!/bin/bash
# round 2, GPU call 8: whole GPU suite on tile kernel v6 (snapshot once per iteration), bench, rocprofv3 evidence
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu --durations=12 --deselect tests/test_gpu_parity.py::test_tile_kernel_against_the_reference_rule_at_config4 > $O/pytest_8.log 2>&1; echo "pytest rc $?" >> $O/pytest_8.log
tail -22 $O/pytest_8.log
show() { python - "$1" <<'PY'
import json, sys
try:
    j=json.load(open(sys.argv[1])); r=j["roofline"]
    print(sys.argv[1].split("/")[-1], "ms/step", round(j["ms_per_step"],2), "frac", round(r["frac"],3), "all", round(r["frac_all_kernels"],3), "aux", {k: round(v,2) for k,v in r["aux_kernels_ms_per_step"].items()}, {k:(round(v["update_kernel_ms_per_step"],2), round(v["frac_all_kernels"],3)) for k,v in r["phases"].items()}, j.get("stress_sampled"))
except Exception as e: print(sys.argv[1], "failed", e)
PY
}
timeout 200 python bench.py --steps 20 --warmup 5 > $O/bench_v6.json 2> $O/bench_v6.err; show $O/bench_v6.json
timeout 120 python bench.py --steps 28 --warmup 2 --cpu-seconds 0 --stress > $O/bench_v6_whole.json 2> $O/bench_v6_whole.err; show $O/bench_v6_whole.json
bash tools/profile_bench.sh > $O/profile_bench_v6.log 2>&1; tail -30 $O/profile_bench_v6.log
python tools/summarize_prof.py gpurun_out/prof $O v6 > $O/summarize_v6.log 2>&1; tail -40 $O/summarize_v6.log
