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
for G, blocks_list in ((1, (1,)), (8, (1, 2, 4)), (4, (1, 2)), (2, (1, 2))):
    for blocks in blocks_list:
        p = oa.LayoutParams.defaults(g, device=0)
        etas = oa.path_linear_sgd_layout_schedule(p)
        engines = []
        for r in range(G):
            pr = oa.LayoutParams.defaults(g, device=0, stream_offset=r * (1 << 20))
            e = HipEngine(g, pr, X0, Y0); e.exchange_mark(); sharded = False; engines.append(e)
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
        ms, n = engines[0].session.kernel_time()
        X, Y = engines[0].result()
        print(json.dumps(dict(exp="ranks_tiled", G=G, exchanges_per_iteration=blocks, stress=oa.path_stress(g, X, Y, 2_000_000, seed=1),
                              path_distance=oa.path_distance(g, X, Y)[0], rank0_kernel_ms=ms, rank0_launches=n,
                              rank0_terms_per_s=1e3 * p.min_term_updates * p.iter_max / G / ms)), flush=True)
        for e in engines: e.close()
