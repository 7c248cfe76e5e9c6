#!/usr/bin/env python3
"""Convergence curves (stress after every iteration) and the BASELINE config 3 sweep (theta, -K on chr6.C4)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import odgi_amd as oa
from odgi_amd import _lib

def emit(**kw):
    print(json.dumps(kw), flush=True)

def curve(g, X0, Y0, flags, label, pairs=1_000_000, iters=30):
    p = oa.LayoutParams.defaults(g, device=0, flags=flags, iter_max=iters)
    etas = oa.path_linear_sgd_layout_schedule(p)
    out = []
    with oa.LayoutSession(g, p) as s:
        s.upload(X0, Y0)
        for it in range(p.iter_max):
            s.iteration(etas[it], it >= p.first_cooling_iteration(), p.min_term_updates)
            s.sync()
            X, Y = s.download()
            out.append(round(oa.path_stress(g, X, Y, pairs, seed=1), 5))
        ms, n = s.kernel_time()
    emit(exp="convergence", graph=label, kernel={0: "tiled", _lib.FLAG_NO_TILES: "per_lane", _lib.FLAG_NO_FAR_CAP: "tiled_no_far_cap"}[flags],
         iter_max=p.iter_max, stress_after_iteration=out, kernel_ms=ms)

which = sys.argv[1:] or ["config4", "config3", "config5"]
if "config4" in which:
    g = oa.Graph.synthetic(1_000_000, 50, seed=42)
    X0, Y0 = oa.initial_layout(g, "d", seed=42)
    emit(exp="convergence", graph="config4 synthetic 1M", stress_initial=oa.path_stress(g, X0, Y0, 1_000_000, seed=1))
    for iters in (30, 10):
        curve(g, X0, Y0, 0, "config4 synthetic 1M", iters=iters)
        curve(g, X0, Y0, _lib.FLAG_NO_FAR_CAP, "config4 synthetic 1M", iters=iters)
        curve(g, X0, Y0, _lib.FLAG_NO_TILES, "config4 synthetic 1M", iters=iters)
    del g
if "config3" in which:
    from oracle import oracle as orc
    g = oa.Graph.from_gfa(os.path.join(ROOT, "tests", "golden", "chr6.C4.gfa"))
    og = orc.Graph.from_product(g)
    X0, Y0 = oa.initial_layout(g, "h")
    for theta in (0.5, 0.9, 0.99, 0.999):
        for K in (0.25, 0.5, 0.75):
            p = oa.LayoutParams.defaults(g, device=0, theta=theta, cooling_start=K)
            X, Y = X0.copy(), Y0.copy()
            st = oa.path_linear_sgd_layout_gpu(g, p, X, Y)
            Xo, Yo, ho = orc.layout_hogwild(og, orc.params_from(p), 8, X0, Y0)
            emit(exp="config3_sweep", graph="chr6.C4", theta=theta, cooling_start=K, n_streams=st["n_streams"], kernel_ms=st["kernel_ms"],
                 gpu_terms_per_s=1e3 * st["term_updates"] / st["kernel_ms"], stress_gpu=orc.path_stress_sampled(og, X, Y, 1_000_000),
                 stress_cpu_oracle=orc.path_stress_sampled(og, Xo, Yo, 1_000_000), cpu_terms_per_s=ho["terms"] / ho["seconds"])
if "config5" in which:
    t0 = time.time()
    g = oa.Graph.synthetic(10_000_000, 50, seed=42)
    X0, Y0 = oa.initial_layout(g, "d", seed=42)
    emit(exp="convergence", graph="config5 synthetic 10M", N=g.n_nodes, S=g.n_steps, build_s=time.time() - t0,
         stress_initial=oa.path_stress(g, X0, Y0, 1_000_000, seed=1))
    curve(g, X0, Y0, 0, "config5 synthetic 10M")
