#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== e2e"; timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/e2e.jsonl
import json, sys, time
sys.path.insert(0, '.')
import odgi_amd as oa
g = oa.Graph.synthetic(1_000_000, 50, seed=42)
X0, Y0 = oa.initial_layout(g, "d", seed=42)
for rep in range(3):
    p = oa.LayoutParams.defaults(g, device=0)
    X, Y = X0.copy(), Y0.copy()
    t = time.perf_counter(); st = oa.path_linear_sgd_layout_gpu(g, p, X, Y); wall = time.perf_counter() - t
    print(json.dumps({"exp": "e2e_tiled", "rep": rep, "python_wall_s": wall, "lib_wall_ms": st["wall_ms"], "kernel_ms": st["kernel_ms"],
                      "e2e_terms_per_s": st["term_updates"] / wall, "kernel_terms_per_s": 1e3 * st["term_updates"] / st["kernel_ms"],
                      "stress": oa.path_stress(g, X, Y, 2_000_000)}))
PY
echo "== bench"; timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; cat gpurun_out/bench.json; tail -2 gpurun_out/bench.err
echo "== bench 2 ranks sharing the GPU (gloo)"
PGSGD_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --stress 2>&1 | grep -v amdgpu.ids | tail -2
