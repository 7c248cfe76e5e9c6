#!/usr/bin/env python3
"""pgsgd_layout_run end to end on BASELINE config 4's graph with its nodes numbered at random: with the run's renaming of the
nodes by path position (default) and without (PGSGD_FLAG_NO_RELABEL: per-lane kernel)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PGSGD_TIMING"] = "1"
import numpy as np
import odgi_amd as oa
from odgi_amd import _lib
g = oa.Graph.synthetic(1_000_000, 50, seed=42)
perm = np.random.RandomState(3).permutation(g.n_nodes)
new_len = np.empty_like(g.node_len); new_len[perm] = g.node_len
h = g.step_handle
gr = oa.Graph.from_arrays(new_len, g.path_first, (perm[h >> 1].astype(np.uint32) << 1) | (h & 1))
X0, Y0 = oa.initial_layout(gr, "d", seed=42)
for name, graph, flags in (("sorted", g, 0), ("random numbering, renamed by the run", gr, 0), ("random numbering, NO_RELABEL", gr, _lib.FLAG_NO_RELABEL)):
    p = oa.LayoutParams.defaults(graph, device=0, flags=flags)
    Xi, Yi = (oa.initial_layout(g, "d", seed=42) if graph is g else (X0, Y0))
    for rep in range(2):
        X, Y = Xi.astype(np.float32), Yi.astype(np.float32)
        t = time.perf_counter()
        st = oa.path_linear_sgd_layout_gpu(graph, p, X, Y)
        wall = time.perf_counter() - t
    print(json.dumps({"case": name, "wall_s": wall, "kernel_ms": st["kernel_ms"], "relabeled": st["relabeled"], "tiled": st["tiled"],
                      "terms_per_s_end_to_end": st["term_updates"] / wall, "stress": oa.path_stress(graph, X, Y, 1_000_000, seed=1)}), flush=True)
