#!/usr/bin/env python3
"""Debug driver: the one-lane tile kernel against the oracle's mirror for different wave-queue thresholds (PGSGD_TILE_WQ)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
name, wq = sys.argv[1], sys.argv[2]
block = sys.argv[3] if len(sys.argv) > 3 else "64"
lanes = sys.argv[4] if len(sys.argv) > 4 else "1"
os.environ.update({"PGSGD_DEBUG": "1", "PGSGD_TILE_FORCE": "1", "PGSGD_TILE_REGION": "64", "PGSGD_TILE_BLOCK": block, "PGSGD_TILE_GRID": "1",
                   "PGSGD_TILE_LANES": lanes, "PGSGD_TILE_WQ": wq})
import odgi_amd as oa
from odgi_amd import _lib
from oracle import oracle as orc
if name == "synthetic":
    g = oa.Graph.synthetic(3000, 4, seed=3)
elif name == "ragged":
    import test_gpu_parity as t
    g = t._ragged_graph(oa)
else:
    g = oa.Graph.from_gfa(os.path.join(ROOT, "tests", "golden", name + ".gfa"))
og = orc.Graph.from_product(g)
X0, Y0 = oa.initial_layout(g, "d", seed=5)
p = oa.LayoutParams.defaults(g, device=0, iter_max=6, min_term_updates=(20 if name == "ragged" else 2) * g.n_steps, flags=_lib.FLAG_EXACT_MATH)
etas = oa.path_linear_sgd_layout_schedule(p)
with oa.LayoutSession(g, p) as s:
    info, tiles, items = s.tile_info(), s.tile_table(), s.tile_items()
    s.upload(X0, Y0)
    fixed, x_off, y_off, q = s.coord_format()
    w0 = s.download_words()
    for it in range(p.iter_max):
        s.iteration(etas[it], it >= p.first_cooling_iteration(), p.min_term_updates)
        s.sync()
        print("iteration", it, "done", flush=True)
    Xg, Yg = s.download()
    w1 = s.download_words()
    ov = s.outbox_overflow()
Xo, Yo, dmax_o, ck, far = orc.tile_layout_q32(og, orc.params_from(p), p.seed, tiles, items, info["region_nodes"], X0, Y0, x_off, y_off, q)
sums = lambda w: (int((w & np.uint64(0xffffffff)).sum()), int((w >> np.uint64(32)).sum()))
print(name, "wq", wq, "block", block, "lanes", lanes, "equal", bool(np.array_equal(Xg, Xo) and np.array_equal(Yg, Yo)), "mismatches", int((Xg != Xo).sum()), "of", len(Xo),
      "checksums", sums(w0) == sums(w1), "overflow", ov, "max dev", float(np.abs(Xg - Xo).max()))
