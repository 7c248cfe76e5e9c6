#!/usr/bin/env python3
"""Debug: region shard with the exact exchange, G virtual ranks, against one GPU with a snapshot pass per iteration: where do they part?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.update({"PGSGD_DEBUG": "1", "PGSGD_TILE_FORCE": "1", "PGSGD_TILE_LANES": "1", "PGSGD_TILE_BLOCK": "64"})
import numpy as np, torch
import odgi_amd as oa
from odgi_amd.distributed import HipEngine
G = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = oa.Graph.synthetic(120_000, 10, seed=11)
kw = dict(min_term_updates=3 * g.n_steps, iter_max=20)
p = oa.LayoutParams.defaults(g, device=0, **kw)
etas = oa.path_linear_sgd_layout_schedule(p)
X0, Y0 = oa.initial_layout(g, "d", seed=4)
os.environ["PGSGD_TILE_SNAPSHOT_PASS"] = "1"
ref = []
with oa.LayoutSession(g, p) as s:
    s.upload(X0, Y0)
    for it in range(iters):
        s.iteration(etas[it], it >= p.first_cooling_iteration(), p.min_term_updates)
        s.sync()
        ref.append(s.download_words())   # flushes: far pulls of the last launch delivered
del os.environ["PGSGD_TILE_SNAPSHOT_PASS"]
engines = [HipEngine(g, oa.LayoutParams.defaults(g, device=0, stream_offset=r * (1 << 20), **kw), X0, Y0) for r in range(G)]
for r, e in enumerate(engines):
    e.exchange_mark()
    print("rank", r, "set_shard", e.set_shard(r, G, by_region="exact"), e.shard_mode)
bufs = [e.new_exact_exchange_buffer(G) for e in engines]
for it in range(iters):
    for colour in range(2):
        for e in engines:
            e.iteration_part(etas[it], it >= p.first_cooling_iteration(), p.min_term_updates, colour, 2)
        for r, (e, b) in enumerate(zip(engines, bufs)):
            e.exchange_exact_begin(b, r, G)
        torch.cuda.synchronize()
        total = torch.stack(bufs).sum(0)
        nz = [int((b[: 2 * g.n_nodes] != 0).sum()) for b in bufs]
        for e in engines:
            e.exchange_exact_end(total, G)
        tail = total[-3 * G:].cpu().numpy()
        print("it", it, "colour", colour, "nonzero deltas per rank", nz, "far counts", tail[:G], "sum", int(tail[:G].sum()))
    for e in engines:
        e.sync()
    w = engines[0].session.download_words()
    print("iteration", it, "mismatches vs one GPU", int((w != ref[it]).sum()), "of", len(w))
