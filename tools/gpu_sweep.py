#!/usr/bin/env python3
"""GPU experiments behind the design decisions in DESIGN.md (run on the MI355X box via gpurun).

  streams : layout quality and speed vs the number of concurrent sampler streams, fixture graphs
  synth   : terms/s vs streams and coordinate-load flavour on the 1M-node synthetic graph
  ranks   : quality of the multi-GPU exchange with G virtual ranks on ONE GPU (G sessions in turn)
Writes JSON lines to gpurun_out/sweep.jsonl.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import odgi_amd as oa  # noqa: E402
from odgi_amd import _lib  # noqa: E402
from odgi_amd.distributed import HipEngine, shard_terms, split_blocks  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
LOG = open(os.path.join(OUT, "sweep.jsonl"), "a")


def emit(**kw):
    line = json.dumps(kw)
    print(line, flush=True)
    LOG.write(line + "\n")
    LOG.flush()


def run_layout(g, p, X0, Y0):
    X, Y = X0.copy(), Y0.copy()
    st = oa.path_linear_sgd_layout_gpu(g, p, X, Y)
    return X, Y, st


def exp_streams(args):
    from oracle import oracle as orc
    for name in ["DRB1-3123_unsorted", "DRB1-3123", "chr6.C4", "LPA"]:
        g = oa.Graph.from_gfa(os.path.join(GOLDEN, name + ".gfa"))
        og = orc.Graph.from_product(g)
        p0 = oa.LayoutParams.defaults(g, device=0)
        X0, Y0 = oa.initial_layout(g, "d", seed=11)
        Xo, Yo, hst = orc.layout_hogwild(og, orc.params_from(p0), min(8, os.cpu_count()), X0, Y0)
        s_cpu = orc.path_stress_sampled(og, Xo, Yo, 1_000_000)
        emit(exp="streams", graph=name, engine="cpu_oracle_hogwild", stress=s_cpu, terms_per_s=hst["terms"] / hst["seconds"])
        for flags in (0, _lib.FLAG_HOGWILD_STORES, _lib.FLAG_HOGWILD_STORES | _lib.FLAG_FP32_ATOMICS):
            for ns in (256, 1024, 4096, 16384, 65536, 262144, 0):
                p = oa.LayoutParams.defaults(g, device=0, n_streams=ns, flags=flags)
                X, Y, st = run_layout(g, p, X0, Y0)
                ok = bool(np.isfinite(X).all() and np.isfinite(Y).all())
                s = orc.path_stress_sampled(og, X, Y, 1_000_000) if ok else float("nan")
                emit(exp="streams", graph=name, flags=flags, n_streams=st["n_streams"], auto=(ns == 0), stress=s,
                     stress_cpu=s_cpu, kernel_ms=st["kernel_ms"], terms_per_s=1e3 * st["term_updates"] / st["kernel_ms"])


def exp_synth(args):
    t0 = time.time()
    g = oa.Graph.synthetic(args.nodes, 50, seed=42)
    X0, Y0 = oa.initial_layout(g, "d", seed=42)
    s_init = oa.path_stress(g, X0, Y0, 1_000_000)
    emit(exp="synth", what="graph", N=g.n_nodes, S=g.n_steps, build_s=time.time() - t0, stress_init=s_init)
    p0 = oa.LayoutParams.defaults(g, device=0)
    etas = oa.path_linear_sgd_layout_schedule(p0)
    for flags in (0, _lib.FLAG_COORD_LOAD_PLAIN):
        for ns in (32768, 65536, 131072, 262144, 524288, 0):
            p = oa.LayoutParams.defaults(g, device=0, n_streams=ns, flags=flags)
            with oa.LayoutSession(g, p) as s:
                s.upload(X0, Y0)
                res = {}
                for tag, it in (("warm", 0), ("warm2", 1), ("mid", 10), ("cool", 20)):
                    s.kernel_time(reset=True)
                    s.iteration(etas[it], it >= 15, p.min_term_updates)
                    s.sync()
                    ms, _ = s.kernel_time()
                    res[tag] = 1e3 * p.min_term_updates / ms
                X, Y = s.download()
                emit(exp="synth", what="speed", plain_loads=bool(flags), n_streams=s.n_streams, auto=(ns == 0), terms_per_s=res,
                     finite=bool(np.isfinite(X).all()), stress_after_4=oa.path_stress(g, X, Y, 500_000))
    # full default schedule at a few stream counts: quality vs concurrency
    for ns in (65536, 262144, 524288, 0):
        p = oa.LayoutParams.defaults(g, device=0, n_streams=ns)
        X, Y, st = run_layout(g, p, X0, Y0)
        emit(exp="synth", what="full30", n_streams=st["n_streams"], auto=(ns == 0), kernel_ms=st["kernel_ms"],
             terms_per_s=1e3 * st["term_updates"] / st["kernel_ms"], wall_ms=st["wall_ms"],
             stress=oa.path_stress(g, X, Y, 2_000_000), path_distance=oa.path_distance(g, X, Y), stress_init=s_init)


def exp_ranks(args):
    """G virtual ranks on one GPU: every rank is a session with its own coordinates and streams."""
    from oracle import oracle as orc
    cases = [("LPA", oa.Graph.from_gfa(os.path.join(GOLDEN, "LPA.gfa")), 30),
             ("synthetic200k", oa.Graph.synthetic(200_000, 20, seed=42), 30)]
    for name, g, iters in cases:
        X0, Y0 = oa.initial_layout(g, "d", seed=11)
        for G in (1, 2, 4, 8):
            for blocks in ((1,) if G == 1 else (1, 4)):
                p = oa.LayoutParams.defaults(g, device=0, iter_max=iters)
                etas = oa.path_linear_sgd_layout_schedule(p)
                engines = []
                for r in range(G):
                    pr = oa.LayoutParams.defaults(g, device=0, iter_max=iters)
                    e = HipEngine(g, pr, X0, Y0)
                    if r == 0:
                        L = e.session.n_streams
                    e.close()
                    pr.n_streams, pr.stream_offset = L, r * L
                    e = HipEngine(g, pr, X0, Y0)
                    e.exchange_mark()
                    engines.append(e)
                bufs = [e.new_exchange_buffer() for e in engines]
                for it in range(iters):
                    for bt in split_blocks(p.min_term_updates, blocks):
                        for r, e in enumerate(engines):
                            e.iteration(etas[it], it >= p.first_cooling_iteration(), shard_terms(bt, G, r))
                        if G > 1:
                            for e, b in zip(engines, bufs):
                                e.exchange_begin(b)
                            torch.cuda.synchronize()
                            total = torch.stack(bufs).sum(0)
                            for e in engines:
                                e.exchange_end(total, G)
                        for e in engines:
                            e.sync()
                X, Y = engines[0].result()
                emit(exp="ranks", graph=name, G=G, exchanges_per_iteration=blocks, stress=oa.path_stress(g, X, Y, 1_000_000),
                     finite=bool(np.isfinite(X).all()))
                for e in engines:
                    e.close()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("exps", nargs="+", choices=["streams", "synth", "ranks"])
    ap.add_argument("--nodes", type=int, default=1_000_000)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    for e in a.exps:
        {"streams": exp_streams, "synth": exp_synth, "ranks": exp_ranks}[e](a)
