#!/usr/bin/env python3
"""A graph whose node ranks are in random order (no tile fits a window) or shuffled in blocks: which kernel the
library picks, and how the tile kernel and the per-lane kernel compare there."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import odgi_amd as oa
from odgi_amd import _lib

def relabel(g0, perm):
    new_len = np.empty_like(g0.node_len); new_len[perm] = g0.node_len
    h = g0.step_handle
    return oa.Graph.from_arrays(new_len, g0.path_first, (perm[h >> 1].astype(np.uint32) << 1) | (h & 1))

g0 = oa.Graph.synthetic(300_000, 24, seed=7)
n = g0.n_nodes
rs = np.random.RandomState(3)
cases = {"random": rs.permutation(n)}
nb = n // 4096
blk = np.arange(n); order = rs.permutation(nb)
blk[:nb * 4096] = np.repeat(order, 4096) * 4096 + np.tile(np.arange(4096), nb)
cases["blocks4096"] = blk
frac = np.arange(n); sel = rs.rand(n // 512) < 0.3     # 30 % of the 512-node blocks get their nodes shuffled internally... and moved
idx = np.where(sel)[0]
for b in idx:
    frac[b * 512:(b + 1) * 512] = b * 512 + rs.permutation(512)
cases["30pct_blocks_scrambled"] = frac
for name, perm in cases.items():
    g = relabel(g0, perm)
    X0, Y0 = oa.initial_layout(g, "d", seed=7)
    for mode, flags in (("default", 0), ("per_lane", _lib.FLAG_NO_TILES)):
        p = oa.LayoutParams.defaults(g, device=0, flags=flags)
        with oa.LayoutSession(g, p) as s:
            info = s.tile_info()
        X, Y = X0.copy(), Y0.copy()
        st = oa.path_linear_sgd_layout_gpu(g, p, X, Y)
        print(json.dumps(dict(exp="shuffled", case=name, mode=mode, tiled=info["tiled"], tiles=info["n_tiles"], windowless=info["n_nonlocal_tiles"],
                              kernel_ms=st["kernel_ms"], terms_per_s=st["term_updates"] / (st["kernel_ms"] * 1e-3),
                              stress=oa.path_stress(g, X, Y, 2_000_000, seed=1))), flush=True)
