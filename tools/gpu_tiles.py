#!/usr/bin/env python3
"""Region-exclusive tile mode vs the default kernel: speed, stress, invariants, robustness."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import odgi_amd as oa
from odgi_amd import _lib
from odgi_amd.distributed import HipEngine, shard_terms, split_blocks

def emit(**kw):
    print(json.dumps(kw), flush=True)

def run(g, X0, Y0, flags, iters=30, seed=9399220):
    p = oa.LayoutParams.defaults(g, device=0, flags=flags, iter_max=iters, seed=seed)
    etas = oa.path_linear_sgd_layout_schedule(p)
    with oa.LayoutSession(g, p) as s:
        info = s.tile_info()
        s.upload(X0, Y0)
        w0 = s.download_words()
        per_iter = []
        for it in range(p.iter_max):
            s.kernel_time(reset=True)
            s.iteration(etas[it], it >= p.first_cooling_iteration(), p.min_term_updates)
            s.sync()
            per_iter.append(s.kernel_time()[0])
        X, Y = s.download()
        w1 = s.download_words()
        lo, hi = np.uint64(0xffffffff), np.uint64(32)
        conserved = int((w0 & lo).sum()) == int((w1 & lo).sum()) and int((w0 >> hi).sum()) == int((w1 >> hi).sum())
        return dict(info=info, n_streams=s.n_streams, ms_warm=per_iter[min(1, iters - 1)], ms_cool=per_iter[-1],
                    terms_per_s=1e3 * p.min_term_updates * p.iter_max / sum(per_iter), conserved=conserved,
                    finite=bool(np.isfinite(X).all() and np.isfinite(Y).all()), stress=oa.path_stress(g, X, Y, 2_000_000),
                    path_distance=oa.path_distance(g, X, Y)[0])

exps = sys.argv[1:] or ["replicates", "regions", "unsorted", "ranks", "big"]
g = oa.Graph.synthetic(1_000_000, 50, seed=42)
if "replicates" in exps:
    for rep in range(3):
        X0, Y0 = oa.initial_layout(g, "d", seed=42 + rep)
        for name, flags in (("per_lane", _lib.FLAG_NO_TILES), ("tiled", 0)):
            emit(exp="tiles_replicates", rep=rep, mode=name, **run(g, X0, Y0, flags, seed=9399220 + 7919 * rep))
X0, Y0 = oa.initial_layout(g, "d", seed=42)
if "regions" in exps:
    for R, B in ((256, 64), (256, 128), (512, 64), (512, 128), (512, 192), (512, 256), (1024, 128), (1024, 256), (2048, 256)):
        os.environ["PGSGD_TILE_REGION"] = str(R)
        os.environ["PGSGD_TILE_BLOCK"] = str(B)
        for rep in range(2):
            Xr, Yr = oa.initial_layout(g, "d", seed=42 + rep)
            emit(exp="tiles_region", region=R, block=B, rep=rep, **run(g, Xr, Yr, 0, seed=9399220 + 7919 * rep))
    os.environ.pop("PGSGD_TILE_REGION"); os.environ.pop("PGSGD_TILE_BLOCK")
if "substeps" in exps:
    g3 = oa.Graph.synthetic(300_000, 24, seed=7)
    X3, Y3 = oa.initial_layout(g3, "d", seed=7)
    for graph_name, gg, XX, YY, mult in (("synthetic300k_3S", g3, X3, Y3, 3), ("synthetic1M_10S", g, X0, Y0, 10)):
        def run2(flags):
            p = oa.LayoutParams.defaults(gg, device=0, flags=flags, min_term_updates=mult * gg.n_steps)
            etas = oa.path_linear_sgd_layout_schedule(p)
            with oa.LayoutSession(gg, p) as s:
                s.upload(XX, YY)
                for it in range(p.iter_max):
                    s.iteration(etas[it], it >= p.first_cooling_iteration(), p.min_term_updates)
                s.sync()
                ms, n = s.kernel_time()
                X, Y = s.download()
            return dict(terms_per_s=1e3 * p.min_term_updates * p.iter_max / ms, launches=n, stress=oa.path_stress(gg, X, Y, 2_000_000), path_distance=oa.path_distance(gg, X, Y)[0])
        emit(exp="tiles_substeps", graph=graph_name, mode="per_lane", **run2(_lib.FLAG_NO_TILES))
        for K in (1, 2, 4, 8, 16):
            os.environ["PGSGD_TILE_SUBSTEPS"] = str(K)
            emit(exp="tiles_substeps", graph=graph_name, mode="tiled", substeps=K, **run2(0))
        os.environ.pop("PGSGD_TILE_SUBSTEPS")
if "unsorted" in exps:
    # relabel the nodes of three stretches at random: their tiles no longer fit a window
    rs = np.random.RandomState(5)
    perm = np.arange(g.n_nodes)
    for a, b in ((100_000, 130_000), (500_000, 505_000), (900_000, 960_000)):
        perm[a:b] = a + rs.permutation(b - a)
    inv = np.empty_like(perm); inv[perm] = np.arange(g.n_nodes)   # old rank -> new rank
    new_len = np.empty_like(g.node_len); new_len[inv] = g.node_len
    h = g.step_handle
    new_h = (inv[h >> 1].astype(np.uint32) << 1) | (h & 1)
    g2 = oa.Graph.from_arrays(new_len, g.path_first, new_h, step_pos=g.step_pos, step_path=g.step_path)
    X2, Y2 = oa.initial_layout(g2, "d", seed=42)
    for name, flags in (("per_lane", _lib.FLAG_NO_TILES), ("tiled", 0)):
        emit(exp="tiles_unsorted", mode=name, **run(g2, X2, Y2, flags))
if "ranks" in exps:
    for G in (1, 4):
        for name, flags in (("per_lane", _lib.FLAG_NO_TILES), ("tiled", 0)):
            p = oa.LayoutParams.defaults(g, device=0, flags=flags)
            etas = oa.path_linear_sgd_layout_schedule(p)
            engines = []
            for r in range(G):
                pr = oa.LayoutParams.defaults(g, device=0, flags=flags, stream_offset=r * (1 << 20))
                e = HipEngine(g, pr, X0, Y0); e.exchange_mark(); engines.append(e)
            bufs = [e.new_exchange_buffer() for e in engines]
            t0 = time.time()
            for it in range(p.iter_max):
                for bt in split_blocks(p.min_term_updates, 4 if G > 1 else 1):
                    for r, e in enumerate(engines):
                        e.iteration(etas[it], it >= p.first_cooling_iteration(), shard_terms(bt, G, r))
                    if G > 1:
                        for e, b in zip(engines, bufs): e.exchange_begin(b)
                        torch.cuda.synchronize()
                        total = torch.stack(bufs).sum(0)
                        for e in engines: e.exchange_end(total, G)
                    for e in engines: e.sync()
            X, Y = engines[0].result()
            emit(exp="tiles_ranks", G=G, mode=name, stress=oa.path_stress(g, X, Y, 2_000_000), wall_s=time.time() - t0)
            for e in engines: e.close()
if "big" in exps:
    del g
    t0 = time.time()
    gb = oa.Graph.synthetic(10_000_000, 50, seed=42)
    Xb, Yb = oa.initial_layout(gb, "d", seed=42)
    emit(exp="tiles_big", what="graph", N=gb.n_nodes, S=gb.n_steps, build_s=time.time() - t0)
    for name, flags in (("per_lane", _lib.FLAG_NO_TILES), ("tiled", 0)):
        emit(exp="tiles_big", mode=name, **run(gb, Xb, Yb, flags, iters=4))
