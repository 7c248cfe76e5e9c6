#!/usr/bin/env python3
"""The per-lane kernel, plain term loop (PGSGD_FLAG_NO_PIPELINE) against the software-pipelined one (the default): kernel time,
terms/s and layout stress on the reference's fixture graphs (lane count bounded by the busiest node), on a randomly
numbered 1M-node graph (no tile fits a window: full residency) and on BASELINE config 4 with the tile kernel switched off."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import odgi_amd as oa
from odgi_amd import _lib

G = os.path.join(ROOT, "tests", "golden")
cases = [(n, oa.Graph.from_gfa(os.path.join(G, n + ".gfa"))) for n in ("DRB1-3123", "DRB1-3123_unsorted", "LPA", "chr6.C4")]
g0 = oa.Graph.synthetic(1_000_000, 50, seed=42)
perm = np.random.RandomState(3).permutation(g0.n_nodes)
new_len = np.empty_like(g0.node_len); new_len[perm] = g0.node_len
h = g0.step_handle
cases.append(("synthetic-1M-shuffled", oa.Graph.from_arrays(new_len, g0.path_first, (perm[h >> 1].astype(np.uint32) << 1) | (h & 1))))
cases.append(("synthetic-1M-sorted (tile kernel off)", g0))
for name, g in cases:
    for mode, flags in (("piped", _lib.FLAG_NO_TILES), ("plain", _lib.FLAG_NO_TILES | _lib.FLAG_NO_PIPELINE)):
        res = []
        for rep in range(3 if g.n_nodes < 100_000 else 1):
            p = oa.LayoutParams.defaults(g, device=0, flags=flags, seed=9399220 + 7919 * rep)
            X, Y = oa.initial_layout(g, "d", seed=7 + rep)
            st = oa.path_linear_sgd_layout_gpu(g, p, X, Y)
            res.append((st["kernel_ms"], st["term_updates"] / (st["kernel_ms"] * 1e-3), oa.path_stress(g, X, Y, 1_000_000, seed=1), st["n_streams"]))
        print(json.dumps(dict(exp="per_lane_ab", graph=name, mode=mode, lanes=res[0][3], kernel_ms=float(np.mean([r[0] for r in res])),
                              terms_per_s=float(np.mean([r[1] for r in res])), stress=[round(r[2], 4) for r in res])), flush=True)
