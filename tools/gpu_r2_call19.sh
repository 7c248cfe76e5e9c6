#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
show() { python - "$1" <<'PY'
import json, sys
try:
    j=json.load(open(sys.argv[1])); r=j["roofline"]
    print(sys.argv[1].split("/")[-1], "ms/step", round(j["ms_per_step"],2), "frac", round(r["frac"],3), "all", round(r["frac_all_kernels"],3), "aux", {k: round(v,2) for k,v in r["aux_kernels_ms_per_step"].items()}, {k:(round(v["update_kernel_ms_per_step"],2), round(v["frac_all_kernels"],3)) for k,v in r["phases"].items()}, j.get("stress_sampled"))
except Exception as e: print(sys.argv[1], "failed", e)
PY
}
for sh in 14 15; do
PGSGD_OUTBOX_SHIFT=$sh timeout 120 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_v11_shift$sh.json 2> $O/bench_v11_shift$sh.err; show $O/bench_v11_shift$sh.json
done
PGSGD_TILE_REGION=512 timeout 120 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_v11_r512.json 2> $O/bench_v11_r512.err; show $O/bench_v11_r512.json
