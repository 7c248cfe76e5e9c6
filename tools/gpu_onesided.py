#!/usr/bin/env python3
"""Tile kernel, far partners: the reference's two-sided update vs PGSGD_FLAG_ONE_SIDED_FAR.
Speed at config 4, stress replicates at config 4 and on a 300k-node graph (3*S and 10*S terms),
short schedules (-x 10), a graph with window-less tiles."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import odgi_amd as oa
from odgi_amd import _lib

MODES = (("two_sided", 0), ("one_sided", _lib.FLAG_ONE_SIDED_FAR))

def emit(**kw):
    print(json.dumps(kw), flush=True)

def run(g, X0, Y0, flags, seed, iters=30, mult=10):
    p = oa.LayoutParams.defaults(g, device=0, flags=flags, iter_max=iters, seed=seed, min_term_updates=mult * g.n_steps)
    etas = oa.path_linear_sgd_layout_schedule(p)
    with oa.LayoutSession(g, p) as s:
        s.upload(X0, Y0)
        per_iter = []
        for it in range(p.iter_max):
            s.kernel_time(reset=True)
            s.iteration(etas[it], it >= p.first_cooling_iteration(), p.min_term_updates)
            s.sync()
            per_iter.append(s.kernel_time()[0])
        X, Y = s.download()
        return dict(tiled=s.tile_info()["tiled"], nonlocal_tiles=s.tile_info()["n_nonlocal_tiles"], n_streams=s.n_streams,
                    ms_warm=per_iter[min(1, iters - 1)], ms_cool=per_iter[-1],
                    terms_per_s=1e3 * p.min_term_updates * p.iter_max / sum(per_iter),
                    finite=bool(np.isfinite(X).all() and np.isfinite(Y).all()),
                    stress=oa.path_stress(g, X, Y, 2_000_000, seed=1), path_distance=oa.path_distance(g, X, Y)[0])

exps = sys.argv[1:] or ["big", "mid", "short", "unsorted"]
if "big" in exps:
    g = oa.Graph.synthetic(1_000_000, 50, seed=42)
    for rep in range(3):
        X0, Y0 = oa.initial_layout(g, "d", seed=42 + rep)
        for name, flags in MODES:
            emit(exp="one_sided_far", graph="synthetic1M", rep=rep, mode=name, **run(g, X0, Y0, flags, 9399220 + 7919 * rep))
    if "short" in exps:
        X0, Y0 = oa.initial_layout(g, "d", seed=42)
        for name, flags in MODES:
            emit(exp="one_sided_far", graph="synthetic1M_x10", mode=name, **run(g, X0, Y0, flags, 9399220, iters=10))
if "mid" in exps:
    g3 = oa.Graph.synthetic(300_000, 24, seed=7)
    for mult in (3, 10):
        rows = {}
        for rep in range(5):
            X0, Y0 = oa.initial_layout(g3, "d", seed=70 + rep)
            for name, flags in MODES + (("per_lane", _lib.FLAG_NO_TILES),):
                r = run(g3, X0, Y0, flags, 9399220 + 7919 * rep, mult=mult)
                rows.setdefault(name, []).append((r["stress"], r["path_distance"], r["terms_per_s"]))
        out = {"exp": "one_sided_far", "graph": "synthetic300k", "terms_per_iter": f"{mult}S"}
        for k, v in rows.items():
            a = np.array(v)
            out[k] = {"stress": [round(x, 4) for x in a[:, 0]], "stress_mean": float(a[:, 0].mean()), "pd_mean": float(a[:, 1].mean()),
                      "terms_per_s": float(a[:, 2].mean())}
        emit(**out)
if "unsorted" in exps:
    # relabel the nodes of two stretches at random: their tiles no longer fit a window (kFarOneSided instance)
    g3 = oa.Graph.synthetic(300_000, 24, seed=7)
    rs = np.random.RandomState(5)
    perm = np.arange(g3.n_nodes)
    for a, b in ((50_000, 60_000), (200_000, 203_000)):
        perm[a:b] = a + rs.permutation(b - a)
    inv = np.empty_like(perm); inv[perm] = np.arange(g3.n_nodes)
    new_len = np.empty_like(g3.node_len); new_len[inv] = g3.node_len
    h = g3.step_handle
    gu = oa.Graph.from_arrays(new_len, g3.path_first, (inv[h >> 1].astype(np.uint32) << 1) | (h & 1))
    for rep in range(3):
        X0, Y0 = oa.initial_layout(gu, "d", seed=70 + rep)
        for name, flags in MODES:
            emit(exp="one_sided_far", graph="synthetic300k_unsorted", rep=rep, mode=name, **run(gu, X0, Y0, flags, 9399220 + 7919 * rep))
