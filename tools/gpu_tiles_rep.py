#!/usr/bin/env python3
"""Replicates: tiled vs per-lane kernel, stress spread over seeds (300k nodes, 3*S and 10*S terms)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import odgi_amd as oa
from odgi_amd import _lib
g = oa.Graph.synthetic(300_000, 24, seed=7)
for mult in (3, 10):
    rows = {}
    for rep in range(5):
        X0, Y0 = oa.initial_layout(g, "d", seed=70 + rep)
        for name, flags, K in (("per_lane", _lib.FLAG_NO_TILES, 1), ("tiled", 0, 1), ("tiled_k2", 0, 2)):
            os.environ["PGSGD_TILE_SUBSTEPS"] = str(K)
            p = oa.LayoutParams.defaults(g, device=0, flags=flags, min_term_updates=mult * g.n_steps, seed=9399220 + 7919 * rep)
            X, Y = X0.copy(), Y0.copy()
            oa.path_linear_sgd_layout_gpu(g, p, X, Y)
            rows.setdefault(name, []).append((oa.path_stress(g, X, Y, 2_000_000, seed=1), oa.path_distance(g, X, Y)[0]))
    out = {"exp": "tiles_rep", "graph": "synthetic300k", "terms_per_iter": f"{mult}S"}
    for k, v in rows.items():
        a = np.array(v)
        out[k] = {"stress": [round(x, 4) for x in a[:, 0]], "stress_mean": float(a[:, 0].mean()), "pd_mean": float(a[:, 1].mean())}
    print(json.dumps(out), flush=True)
