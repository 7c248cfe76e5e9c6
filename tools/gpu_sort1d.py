#!/usr/bin/env python3
"""1D path-guided SGD (odgi sort -Y) at scale: throughput and quality on the 1M-node synthetic graph
whose node ranks were shuffled in blocks, and the CPU restatement (oracle, Hogwild threads) on a sample."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import odgi_amd as oa
from odgi_amd import sort as osort

def emit(**kw):
    print(json.dumps(kw), flush=True)

for n_nodes, n_paths in ((1_000_000, 50), (200_000, 20)):
    g = oa.Graph.synthetic(n_nodes, n_paths, seed=42)
    p = osort.sort_params_defaults(g, seed=9399220)
    t0 = time.time()
    X, st = osort.path_linear_sgd(g, p)
    wall = time.time() - t0
    order = osort.order_from_positions(X)
    # the synthetic graph is generated in its true order: rank correlation of the new order with it
    pos = np.empty(n_nodes, dtype=np.int64); pos[order.astype(np.int64)] = np.arange(n_nodes)
    rho = float(np.corrcoef(pos, np.arange(n_nodes))[0, 1])
    emit(exp="sort_1d", nodes=n_nodes, steps=int(g.n_steps), iter_max=int(p.iter_max), terms=int(st["term_updates"]),
         kernel_ms=st["kernel_ms"], wall_s=wall, terms_per_s=st["term_updates"] / (st["kernel_ms"] * 1e-3),
         n_streams=int(st["n_streams"]), stress=osort.sort_stress(g, X), stress_initial=osort.sort_stress(g, osort.sort_initial(g)),
         rank_correlation=rho)
