#!/usr/bin/env python3
"""Graph shapes the synthetic benchmark graph does not have: thousands of short paths; two components laid end to
end; a few very long nodes.  Tile kernel (default plan) vs per-lane kernel: which runs, speed, stress."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import odgi_amd as oa
from odgi_amd import _lib
rs = np.random.RandomState(21)
n = 300_000
base_len = np.minimum(np.maximum(rs.geometric(1 / 32.0, n), 1), 4096).astype(np.uint32)

def short_paths():
    handles, first = [], [0]
    for p in range(3000):
        cnt = int(rs.randint(500, 4000)); start = int(rs.randint(0, n - cnt))
        keep = rs.rand(cnt) > 0.03
        r = np.arange(start, start + cnt)[keep]
        handles.append((2 * r + (rs.rand(len(r)) < 0.01)).astype(np.uint32)); first.append(first[-1] + len(r))
    return oa.Graph.from_arrays(base_len, np.array(first, dtype=np.uint64), np.concatenate(handles))

def two_components():
    half = n // 2
    handles, first = [], [0]
    for p in range(24):
        lo, hi = (0, half) if p % 2 == 0 else (half, n)
        keep = rs.rand(hi - lo) > 0.03
        r = np.arange(lo, hi)[keep]
        handles.append((2 * r).astype(np.uint32)); first.append(first[-1] + len(r))
    return oa.Graph.from_arrays(base_len, np.array(first, dtype=np.uint64), np.concatenate(handles))

def long_nodes():
    ln = base_len.copy(); ln[rs.choice(n, 30, replace=False)] = 1_000_000
    g0 = oa.Graph.synthetic(n, 24, seed=7)
    return oa.Graph.from_arrays(ln, g0.path_first, g0.step_handle)

for name, make in (("short_paths", short_paths), ("two_components", two_components), ("long_nodes", long_nodes)):
    g = make()
    X0, Y0 = oa.initial_layout(g, "d", seed=7)
    for mode, flags in (("default", 0), ("per_lane", _lib.FLAG_NO_TILES)):
        p = oa.LayoutParams.defaults(g, device=0, flags=flags)
        with oa.LayoutSession(g, p) as s:
            s.upload(X0, Y0)
            info = s.tile_info()
        X, Y = X0.copy(), Y0.copy()
        st = oa.path_linear_sgd_layout_gpu(g, p, X, Y)
        print(json.dumps(dict(exp="shapes", case=name, mode=mode, steps=int(g.n_steps), paths=int(g.n_paths), tiled=info["tiled"], warm_per_lane=info["warm_per_lane"],
                              windowless=info["n_nonlocal_tiles"], tiles=info["n_tiles"], kernel_ms=st["kernel_ms"],
                              terms_per_s=st["term_updates"] / (st["kernel_ms"] * 1e-3), finite=bool(np.isfinite(X).all()),
                              stress=oa.path_stress(g, X, Y, 2_000_000, seed=1), path_distance=oa.path_distance(g, X, Y)[0])), flush=True)
