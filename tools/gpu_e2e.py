#!/usr/bin/env python3
"""pgsgd_layout_run end to end (host buffers in, host buffers out) at config 4, with the PGSGD_TIMING
phase breakdown on stderr.  usage: gpu_e2e.py [reps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PGSGD_TIMING"] = "1"
import odgi_amd as oa
g = oa.Graph.synthetic(1_000_000, 50, seed=42)
X0, Y0 = oa.initial_layout(g, "d", seed=42)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    p = oa.LayoutParams.defaults(g, device=0)
    X, Y = X0.copy(), Y0.copy()
    t0 = time.perf_counter()
    st = oa.path_linear_sgd_layout_gpu(g, p, X, Y)
    wall = time.perf_counter() - t0
    terms = p.min_term_updates * p.iter_max
    print(json.dumps(dict(exp="e2e_tiled", rep=rep, python_wall_s=wall, lib_wall_ms=st["wall_ms"], kernel_ms=st["kernel_ms"],
                          e2e_terms_per_s=terms / wall, kernel_terms_per_s=terms / (st["kernel_ms"] * 1e-3))), flush=True)
    print(f"--- rep {rep} done", file=sys.stderr, flush=True)
