#!/usr/bin/env python3
"""Merge-rule experiment with G = 8 virtual ranks (tile shard): from iteration T on the exchange adds the ranks'
moves (f = 1) instead of applying the coherence factor clamp(Q/|S|^2, 1/G, 1) — small coherent gradient steps of
the late iterations should add up, not average."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import odgi_amd as oa
from odgi_amd.distributed import HipEngine
g = oa.Graph.synthetic(1_000_000, 50, seed=42)
G = 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for rep in range(reps):
    X0, Y0 = oa.initial_layout(g, "d", seed=42 + rep)
    for T in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "99,23,20,17,15").split(",")]:
        p = oa.LayoutParams.defaults(g, device=0)
        etas = oa.path_linear_sgd_layout_schedule(p)
        engines = []
        for r in range(G):
            e = HipEngine(g, oa.LayoutParams.defaults(g, device=0, stream_offset=r * (1 << 20), seed=9399220 + 7919 * rep), X0, Y0)
            e.exchange_mark(); e.set_shard(r, G, by_region=False); engines.append(e)
        bufs = [e.new_exchange_buffer() for e in engines]
        for it in range(p.iter_max):
            for e in engines:
                e.iteration_part(etas[it], it >= p.first_cooling_iteration(), p.min_term_updates, 0, 1)
            for e, b in zip(engines, bufs): e.exchange_begin(b)
            torch.cuda.synchronize()
            total = torch.stack(bufs).sum(0)
            for e in engines: e.exchange_end(total, G if it < T else 1)
            for e in engines: e.sync()
        X, Y = engines[0].result()
        print(json.dumps(dict(exp="merge_rule", rep=rep, G=G, plain_sum_from_iteration=T, stress=oa.path_stress(g, X, Y, 2_000_000, seed=1),
                              path_distance=oa.path_distance(g, X, Y)[0])), flush=True)
        for e in engines: e.close()
