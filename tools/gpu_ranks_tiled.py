#!/usr/bin/env python3
"""G virtual ranks on one GPU with the tile kernel: stress vs exchanges per iteration."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import odgi_amd as oa
from odgi_amd.distributed import HipEngine, shard_terms, split_blocks
g = oa.Graph.synthetic(1_000_000, 50, seed=42)
X0, Y0 = oa.initial_layout(g, "d", seed=42)
# argv[1]: "terms" (every rank runs every tile with 1/G of its terms), "tiles" (default: every rank runs every
# G-th tile with its whole share) or "regions" (every rank runs every G-th work item = node region with all its tiles)
# argv[2]: comma-separated G:exchanges list (default 1:1,8:1,8:2,4:1,2:1); argv[3]: replicates (initial layout / sampler seeds)
mode = sys.argv[1] if len(sys.argv) > 1 else "tiles"
cfgs = [tuple(int(v) for v in c.split(":")) for c in (sys.argv[2] if len(sys.argv) > 2 else "1:1,8:1,8:2,4:1,2:1").split(",")]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
for rep in range(reps):
    X0, Y0 = oa.initial_layout(g, "d", seed=42 + rep)
    for G, blocks in cfgs:
        p = oa.LayoutParams.defaults(g, device=0)
        etas = oa.path_linear_sgd_layout_schedule(p)
        engines = []
        for r in range(G):
            pr = oa.LayoutParams.defaults(g, device=0, stream_offset=r * (1 << 20), seed=9399220 + 7919 * rep)
            e = HipEngine(g, pr, X0, Y0); e.exchange_mark(); engines.append(e)
            sharded = (mode == "tiles" and e.set_shard(r, G, by_region=False)) or (mode == "regions" and e.set_shard(r, G, by_region=True))
        bufs = [e.new_exchange_buffer() for e in engines]
        kms = 0.0
        for it in range(p.iter_max):
            for b in range(blocks):
                for r, e in enumerate(engines):
                    e.iteration_part(etas[it], it >= p.first_cooling_iteration(),
                                     p.min_term_updates if sharded else shard_terms(p.min_term_updates, G, r), b, blocks)
                if G > 1:
                    for e, b in zip(engines, bufs): e.exchange_begin(b)
                    torch.cuda.synchronize()
                    total = torch.stack(bufs).sum(0)
                    for e in engines: e.exchange_end(total, G)
                for e in engines: e.sync()
        # the far pulls of every rank's last launch: delivered, then merged like any other move (DistributedLayout.finish)
        for e in engines: e.flush()
        if G > 1:
            for e, b in zip(engines, bufs): e.exchange_begin(b)
            torch.cuda.synchronize()
            total = torch.stack(bufs).sum(0)
            for e in engines: e.exchange_end(total, G)
        for e in engines: e.sync()
        ms, n = engines[0].session.kernel_time()
        X, Y = engines[0].result()
        aux = engines[0].session.aux_time()
        print(json.dumps(dict(exp="ranks_tiled", shard=mode, rep=rep, G=G, exchanges_per_iteration=blocks, stress=oa.path_stress(g, X, Y, 2_000_000, seed=1),
                              near_exact=oa.path_stress_near(g, X, Y, zmax=4)["near"], rank0_snapshot_ms=aux[0], rank0_drain_ms=aux[1],
                              path_distance=oa.path_distance(g, X, Y)[0], rank0_kernel_ms=ms, rank0_launches=n,
                              rank0_terms_per_s=1e3 * p.min_term_updates * p.iter_max / G / ms)), flush=True)
        for e in engines: e.close()
