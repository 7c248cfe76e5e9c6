#!/usr/bin/env python3
"""Short schedules (`-x N`): final sampled stress of the tile kernel against the per-lane kernel (the reference's rule term by
term) at BASELINE config 4 for iter_max = 3 .. 30 — where does the tile kernel's gentle start stop costing layout quality?"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import odgi_amd as oa
from odgi_amd import _lib

g = oa.Graph.synthetic(int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 50, seed=42)
for it in (3, 4, 5, 6, 8, 10, 12, 15, 20, 30):
    row = dict(exp="short_schedules", nodes=g.n_nodes, iter_max=it)
    for mode, flags in (("tile", 0), ("per_lane", _lib.FLAG_NO_TILES)):
        res = []
        for rep in range(2):
            X, Y = oa.initial_layout(g, "d", seed=42 + rep)
            p = oa.LayoutParams.defaults(g, device=0, iter_max=it, flags=flags, seed=9399220 + 7919 * rep)
            st = oa.path_linear_sgd_layout_gpu(g, p, X, Y)
            res.append(oa.path_stress(g, X, Y, 2_000_000, seed=1))
        row[mode] = [float("%.5g" % v) for v in res]
    print(json.dumps(row), flush=True)
