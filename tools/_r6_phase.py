#!/usr/bin/env python3
"""Profiling instance 5 (make -C odgi_amd/csrc ../lib/libpgsgd_x5.so; PGSGD_DEBUG=1 PGSGD_LIB=libpgsgd_x5.so): where a tile's
time goes at config 4 — thread 0 of every workgroup sums, in 100 MHz ticks, tile start .. term loop start (the tile record, the
staging of its step records, the lanes' stream seeds) and the term loop; read through pgsgd_session_tile_conflicts."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import odgi_amd as oa
g = oa.Graph.synthetic(1_000_000, 50, seed=42)
p = oa.LayoutParams.defaults(g)
X0, Y0 = oa.initial_layout(g, "d", seed=42)
etas = oa.path_linear_sgd_layout_schedule(p)
with oa.LayoutSession(g, p) as s:
    s.upload(X0, Y0)
    prev = (0, 0)
    for it in range(p.iter_max):
        s.iteration(etas[it], it >= p.first_cooling_iteration(), p.min_term_updates)
        s.sync()
        a, b = s.tile_conflicts()
        if it in (6, 10, 14, 16, 20, 28):
            print(json.dumps({"iteration": it, "cooling": it >= p.first_cooling_iteration(), "prologue_ticks": a - prev[0], "loop_ticks": b - prev[1],
                              "prologue_share": (a - prev[0]) / max(1, (a - prev[0]) + (b - prev[1])), "tiles": s.tile_info()["n_tiles"]}))
        prev = (a, b)
