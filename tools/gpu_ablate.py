#!/usr/bin/env python3
"""Ablation of the update kernel on the 1M-node synthetic graph (profiling only; results of the
ablated variants are not layouts)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import odgi_amd as oa
from odgi_amd import _lib
g = oa.Graph.synthetic(1_000_000, 50, seed=42)
X0, Y0 = oa.initial_layout(g, "d", seed=42)
p0 = oa.LayoutParams.defaults(g, device=0)
etas = oa.path_linear_sgd_layout_schedule(p0)
names = {0: "full", 1: "no_atomics", 3: "no_coord_loads", 4: "no_atomics_no_coord_loads"}
FORMATS = (("q32_packed_u64_atomics", 0), ("q32_hogwild_stores", _lib.FLAG_HOGWILD_STORES),
           ("f32_hogwild_stores", _lib.FLAG_FP32_ATOMICS | _lib.FLAG_HOGWILD_STORES), ("f32_atomics", _lib.FLAG_FP32_ATOMICS))
for fmt_name, fmt_flag in FORMATS:
    for abl in (0,):
        for ns in ((0, 524288) if abl == 0 else (0,)):
            p = oa.LayoutParams.defaults(g, device=0, flags=(abl << 8) | fmt_flag, n_streams=ns)
            with oa.LayoutSession(g, p) as s:
                s.upload(X0, Y0)
                res = {}
                for tag, it in (("warm", 0), ("cool", 20)):
                    s.iteration(etas[it], it >= 15, p.min_term_updates); s.sync()
                    s.kernel_time(reset=True)
                    s.iteration(etas[it], it >= 15, p.min_term_updates); s.sync()
                    ms, _ = s.kernel_time()
                    res[tag] = 1e3 * p.min_term_updates / ms
                print(json.dumps({"exp": "ablate", "format": fmt_name, "variant": names[abl], "n_streams": s.n_streams, "terms_per_s": res}), flush=True)
# quality of the full default schedule in both formats
for fmt_name, fmt_flag in FORMATS:
    p = oa.LayoutParams.defaults(g, device=0, flags=fmt_flag)
    X, Y = X0.copy(), Y0.copy()
    st = oa.path_linear_sgd_layout_gpu(g, p, X, Y)
    print(json.dumps({"exp": "full30", "format": fmt_name, "n_streams": st["n_streams"], "kernel_ms": st["kernel_ms"],
                      "terms_per_s": 1e3 * st["term_updates"] / st["kernel_ms"], "stress": oa.path_stress(g, X, Y, 2_000_000),
                      "path_distance": oa.path_distance(g, X, Y)}), flush=True)
