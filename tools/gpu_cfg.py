#!/usr/bin/env python3
"""Tile kernel at a given region size / workgroup size on a synthetic graph: speed and stress replicates.
usage: gpu_cfg.py NODES PATHS REPS R:B [R:B ...]   (R:B = region nodes : lanes per workgroup; 0:0 = per-lane kernel)"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import odgi_amd as oa
from odgi_amd import _lib
nodes, paths, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g = oa.Graph.synthetic(nodes, paths, seed=42 if nodes == 1_000_000 else 7)
for cfg in sys.argv[4:]:
    R, B = (int(x) for x in cfg.split(":"))
    rows = []
    for rep in range(reps):
        X0, Y0 = oa.initial_layout(g, "d", seed=70 + rep)
        if R:
            os.environ["PGSGD_TILE_REGION"], os.environ["PGSGD_TILE_BLOCK"] = str(R), str(B)
        p = oa.LayoutParams.defaults(g, device=0, flags=0 if R else _lib.FLAG_NO_TILES, seed=9399220 + 7919 * rep)
        etas = oa.path_linear_sgd_layout_schedule(p)
        with oa.LayoutSession(g, p) as s:
            info = s.tile_info()
            s.upload(X0, Y0)
            for it in range(p.iter_max):
                s.iteration(etas[it], it >= p.first_cooling_iteration(), p.min_term_updates)
            s.sync()
            ms, _ = s.kernel_time()
            aux = s.aux_time() if R else (0.0, 0.0)
            X, Y = s.download()
            lanes = s.n_streams
        rows.append((1e3 * p.min_term_updates * p.iter_max / ms, oa.path_stress(g, X, Y, 2_000_000, seed=1)))
        os.environ.pop("PGSGD_TILE_REGION", None); os.environ.pop("PGSGD_TILE_BLOCK", None)
    a = np.array(rows)
    print(json.dumps(dict(exp="tile_cfg", nodes=nodes, paths=paths, region=R, block=B, tiled=info["tiled"], work_items=info["n_work_items"], lanes=lanes,
                          terms_per_s=float(a[:, 0].mean()), stress=[round(x, 4) for x in a[:, 1]], stress_mean=float(a[:, 1].mean()), update_kernel_ms=ms, snapshot_ms=aux[0], drain_ms=aux[1])), flush=True)
