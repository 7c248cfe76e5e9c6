#!/bin/bash
# round 2, GPU call 4: tile kernel v2 (per-lane streams, pipelined gather, staged 64-byte outbox lines, tabled Zipf)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout 600 python -m pytest tests -x -q -m gpu --durations=6 -k "tile or tiled or million or outbox or sampler or one_stream" > $O/pytest_4.log 2>&1; echo "pytest rc $?" >> $O/pytest_4.log
tail -12 $O/pytest_4.log
timeout 120 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_v2.json 2> $O/bench_v2.err
timeout 120 python bench.py --steps 28 --warmup 2 --cpu-seconds 0 --stress > $O/bench_v2_whole.json 2>> $O/bench_v2.err
tail -3 $O/bench_v2.err
python - <<PY
import json
for n in ("bench_v2","bench_v2_whole"):
    try:
        j=json.load(open("$O/%s.json"%n)); r=j["roofline"]
        print(n, "ms/step", round(j["ms_per_step"],2), "frac", round(r["frac"],3), "all", round(r["frac_all_kernels"],3), "aux", r["aux_kernels_ms_per_step"], {k:(round(v["update_kernel_ms_per_step"],2), round(v["frac_all_kernels"],3)) for k,v in r["phases"].items()}, j.get("stress_sampled"))
    except Exception as e: print(n, "failed", e)
PY
