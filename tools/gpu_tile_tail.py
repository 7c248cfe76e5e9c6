#!/usr/bin/env python3
"""How much of a tile launch is tail: the share of (workgroups x launch duration) its persistent workgroups are alive for
(pgsgd_session_tile_tail), for the second colour's launch of every iteration of the schedule.  Usage: gpu_tile_tail.py [N]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PGSGD_DEBUG"] = "1"
os.environ["PGSGD_TILE_TAIL"] = "1"
import odgi_amd as oa

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
g = oa.Graph.synthetic(N, 50, seed=42)
X0, Y0 = oa.initial_layout(g, "d", seed=42)
p = oa.LayoutParams.defaults(g, device=0)
etas = oa.path_linear_sgd_layout_schedule(p)
rows = []
with oa.LayoutSession(g, p) as s:
    s.upload(X0, Y0)
    w0 = s.download_words()
    for it in range(p.iter_max):
        s.iteration(etas[it], it >= p.first_cooling_iteration(), p.min_term_updates)
        alive, ms, n = s.tile_tail()
        rows.append({"iteration": it, "cooling": it >= p.first_cooling_iteration(), "alive": round(alive, 4), "launch_ms": round(ms, 3), "workgroups": n})
    info = s.tile_info()
    w1 = s.download_words()
    X, Y = s.download_f64()
import numpy as np
sums = lambda w: (int((w & np.uint64(0xffffffff)).sum()), int((w >> np.uint64(32)).sum()))
conserved = sums(w0) == sums(w1)   # every term moves its two ends by opposite steps: the word sums never change (a stale or lost window would)
stress = oa.path_stress(g, X, Y, 2_000_000, seed=1)
for r in rows:
    print(json.dumps(r))
warm = [r["alive"] for r in rows if not r["cooling"] and r["iteration"] >= 2]
cool = [r["alive"] for r in rows if r["cooling"]]
print(json.dumps({"nodes": N, "mean_alive_warm": sum(warm) / max(1, len(warm)), "mean_alive_cooling": sum(cool) / max(1, len(cool)),
                  "region": info.get("region_nodes"), "tile_steps": info.get("tile_steps"), "items": info.get("n_work_items"),
                  "split": os.environ.get("PGSGD_TILE_SPLIT", "1"), "word_sums_conserved": conserved, "stress": stress}))
