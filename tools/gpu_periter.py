#!/usr/bin/env python3
"""Kernel time of every iteration of the schedule at config 4 (tile kernel, and per-lane with --no-tiles)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import odgi_amd as oa
from odgi_amd import _lib
g = oa.Graph.synthetic(1_000_000, 50, seed=42)
X0, Y0 = oa.initial_layout(g, "d", seed=42)
for name, flags in (("tiled", 0), ("per_lane", _lib.FLAG_NO_TILES)):
    if name == "per_lane" and "--no-tiles" not in sys.argv:
        continue
    p = oa.LayoutParams.defaults(g, device=0, flags=flags)
    etas = oa.path_linear_sgd_layout_schedule(p)
    ms = []
    with oa.LayoutSession(g, p) as s:
        s.upload(X0, Y0)
        for it in range(p.iter_max):
            s.kernel_time(reset=True)
            s.iteration(etas[it], it >= p.first_cooling_iteration(), p.min_term_updates)
            s.sync()
            ms.append(round(s.kernel_time()[0], 2))
    print(json.dumps(dict(exp="per_iteration_ms", mode=name, eta=[float(f"{e:.3g}") for e in etas[:p.iter_max]], ms=ms, total_ms=sum(ms))), flush=True)
