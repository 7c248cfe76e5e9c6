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
for m in (1, 2, 4, 8, 16):
    for ns in (0, 524288):
        p = oa.LayoutParams.defaults(g, device=0, terms_per_anchor=m, n_streams=ns)
        with oa.LayoutSession(g, p) as s:
            s.upload(X0, Y0)
            res = {}
            for tag, it in (("warm", 0), ("cool", 20)):
                s.iteration(etas[it], it >= 15, p.min_term_updates); s.sync()
                s.kernel_time(reset=True)
                s.iteration(etas[it], it >= 15, p.min_term_updates); s.sync()
                ms, _ = s.kernel_time()
                res[tag] = 1e3 * p.min_term_updates / ms
            print(json.dumps({"exp": "anchor_speed", "terms_per_anchor": m, "n_streams": s.n_streams, "terms_per_s": res}), flush=True)
for m in (1, 2, 4, 8, 16):
    p = oa.LayoutParams.defaults(g, device=0, terms_per_anchor=m)
    X, Y = X0.copy(), Y0.copy()
    st = oa.path_linear_sgd_layout_gpu(g, p, X, Y)
    print(json.dumps({"exp": "anchor_full30", "terms_per_anchor": m, "n_streams": st["n_streams"], "kernel_ms": st["kernel_ms"],
                      "terms_per_s": 1e3 * st["term_updates"] / st["kernel_ms"], "stress": oa.path_stress(g, X, Y, 2_000_000),
                      "path_distance": oa.path_distance(g, X, Y)}), flush=True)
# small graphs: quality vs terms_per_anchor against the CPU oracle
from oracle import oracle as orc
for name in ("DRB1-3123", "chr6.C4", "LPA"):
    gs = oa.Graph.from_gfa(os.path.join(ROOT, "tests", "golden", name + ".gfa"))
    og = orc.Graph.from_product(gs)
    Xs, Ys = oa.initial_layout(gs, "d", seed=11)
    ps = oa.LayoutParams.defaults(gs, device=0)
    Xo, Yo, _ = orc.layout_hogwild(og, orc.params_from(ps), 8, Xs, Ys)
    row = {"exp": "anchor_small", "graph": name, "cpu_oracle": orc.path_stress_sampled(og, Xo, Yo, 1_000_000)}
    for m in (1, 2, 4, 8, 16):
        ps = oa.LayoutParams.defaults(gs, device=0, terms_per_anchor=m)
        X, Y = Xs.copy(), Ys.copy()
        oa.path_linear_sgd_layout_gpu(gs, ps, X, Y)
        row[f"m{m}"] = orc.path_stress_sampled(og, X, Y, 1_000_000)
    print(json.dumps(row), flush=True)
