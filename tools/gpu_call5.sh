#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v "amdgpu.ids\|2D path-guided SGD: iteration" | tail -6 | tee gpurun_out/pytest_gpu.log
echo "== e2e"; timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/e2e.jsonl
import json, sys, time
sys.path.insert(0, '.')
import odgi_amd as oa
g = oa.Graph.synthetic(1_000_000, 50, seed=42)
X0, Y0 = oa.initial_layout(g, "d", seed=42)
for rep in range(2):
    p = oa.LayoutParams.defaults(g, device=0)
    X, Y = X0.copy(), Y0.copy()
    t = time.perf_counter(); st = oa.path_linear_sgd_layout_gpu(g, p, X, Y); wall = time.perf_counter() - t
    print(json.dumps({"exp": "e2e", "rep": rep, "python_wall_s": wall, "lib_wall_ms": st["wall_ms"], "kernel_ms": st["kernel_ms"],
                      "terms": st["term_updates"], "e2e_terms_per_s": st["term_updates"] / wall, "kernel_terms_per_s": 1e3 * st["term_updates"] / st["kernel_ms"]}))
PY
bash tools/profile_bench.sh > gpurun_out/profile.log 2>&1; tail -5 gpurun_out/profile.log
