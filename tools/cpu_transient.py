#!/usr/bin/env python3
"""CPU experiment (no GPU): the transient of the tile pipeline under different launch/drain orders, run on the oracle's
sequential mirror of the tile kernel, against the reference rule (CPU restatement of the Hogwild loop).
Prints sampled path stress after selected iterations.  tools/cpu_transient.py [nodes] [paths]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import odgi_amd as oa
from oracle import oracle as orc
import pyref

def frame(g, X, Y, factor=8.0):
    """choose_xform of pgsgd_session.hip"""
    ext = max(float(X.max() - X.min()), float(Y.max() - Y.min()), float(g.max_path_bp()), 1.0)
    span_log2 = int(np.ceil(np.log2(factor * ext)))
    span = 2.0 ** span_log2
    q = float(np.float32(2.0 ** (32 - span_log2)))
    return 0.5 * (float(X.min()) + float(X.max())) - 0.5 * span, 0.5 * (float(Y.min()) + float(Y.max())) - 0.5 * span, q

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    paths = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    g = oa.Graph.synthetic(n, paths, seed=7)
    og = orc.Graph.from_product(g)
    p = oa.LayoutParams.defaults(g)
    X0, Y0 = oa.initial_layout(g, "d", seed=5)
    R, T = 256, 224
    tiles, items = pyref.build_tiles_py(g.path_first, g.step_handle, R, T)
    tiles["lanes"] = np.full(len(tiles["t0"]), 256, dtype=np.uint32)
    mpb = 0
    for i in range(g.n_paths):
        last = int(g.path_first[i + 1]) - 1
        mpb = max(mpb, int(g.step_pos[last]) + int(g.node_len[int(g.step_handle[last]) >> 1]))
    g.max_path_bp = lambda: mpb
    x_off, y_off, q = frame(g, X0, Y0)
    its = [1, 2, 3, 5, 10, 15, 20, 30]
    op = orc.params_from(p)
    print(json.dumps(dict(nodes=g.n_nodes, steps=g.n_steps, terms_per_iter=p.min_term_updates, initial=orc.path_stress_sampled(og, X0, Y0, 400000))), flush=True)
    t = time.time()
    _, _, st, sx, sy = orc.layout_hogwild_curve(og, op, 8, X0, Y0, its, fast=True)
    print(json.dumps(dict(policy="reference rule (CPU restatement, 8 threads)", seconds=round(time.time() - t, 1),
                          stress={k: orc.path_stress_sampled(og, sx[i], sy[i], 400000) for i, k in enumerate(its)})), flush=True)
    NF = orc.TILE_NO_FLUSH   # what a snapshot after the iteration sees
    CR = orc.TILE_CONSTANT_RELAX
    for name, pol in (("round 2: drain after its launch, two snapshots per warm iteration, far pulls = half a projection", orc.TILE_ROUND2),
                      ("drain before the next launch, two snapshots, half a projection", orc.TILE_TWO_SNAPSHOTS | CR | NF),
                      ("drain before the next launch, one snapshot, half a projection", CR | NF),
                      ("shipped: drain before the next launch, one snapshot, gentle first iterations", NF)):
        out = {}
        t = time.time()
        for k in its:
            X, Y, _, _, _ = orc.tile_layout_q32(og, op, p.seed, tiles, items, R, X0, Y0, x_off, y_off, q, policy=pol, stop_after=k)
            out[k] = orc.path_stress_sampled(og, X, Y, 400000)
        print(json.dumps(dict(policy=name, seconds=round(time.time() - t, 1), stress=out)), flush=True)

if __name__ == "__main__":
    main()
