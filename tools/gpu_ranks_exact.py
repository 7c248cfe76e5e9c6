#!/usr/bin/env python3
"""G virtual ranks on one GPU, region shard with the EXACT exchange (every rank owns every G-th window of a colour; after each
colour's launch the ranks sum what they changed as 64-bit integers) at a size of choice — also where the product's rule would
not pick it (config 4 at G = 8: 244 windows per rank and launch on 1280 workgroup slots): what window ownership costs a rank
when the windows no longer fill its device.   gpu_ranks_exact.py NODES G[,G...] [reps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import odgi_amd as oa
from odgi_amd.distributed import HipEngine
N = int(float(sys.argv[1]))
Gs = [int(v) for v in sys.argv[2].split(",")]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
g = oa.Graph.synthetic(N, 50, seed=42)
for rep in range(reps):
    X0, Y0 = oa.initial_layout(g, "d", seed=42 + rep)
    for G in Gs:
        p = oa.LayoutParams.defaults(g, device=0)
        etas = oa.path_linear_sgd_layout_schedule(p)
        engines = []
        for r in range(G):
            e = HipEngine(g, oa.LayoutParams.defaults(g, device=0, stream_offset=r * (1 << 20), seed=9399220 + 7919 * rep), X0, Y0)
            e.exchange_mark()
            assert e.set_shard(r, G, by_region="exact")
            engines.append(e)
        bufs = [e.new_exact_exchange_buffer(G) for e in engines]

        def exchange():
            for r, (e, b) in enumerate(zip(engines, bufs)):
                e.exchange_exact_begin(b, r, G)
            torch.cuda.synchronize()
            total = torch.stack(bufs).sum(0)
            for e in engines:
                e.exchange_exact_end(total, G)

        for it in range(p.iter_max):
            for colour in range(2):
                for e in engines:
                    e.iteration_part(etas[it], it >= p.first_cooling_iteration(), p.min_term_updates, colour, 2)
                exchange()
            for e in engines:
                e.sync()
        for e in engines:
            e.flush()
        ms, n = engines[0].session.kernel_time()
        aux = engines[0].session.aux_time()
        X, Y = engines[0].result()
        print(json.dumps(dict(exp="ranks_exact", nodes=N, rep=rep, G=G, near_exact=oa.path_stress_near(g, X, Y, zmax=4)["near"], stress=oa.path_stress(g, X, Y, 2_000_000, seed=1),
                              rank0_kernel_ms=ms, rank0_launches=n, rank0_snapshot_ms=aux[0], rank0_drain_ms=aux[1], info=engines[0].session.tile_info())), flush=True)
        for e in engines:
            e.close()
