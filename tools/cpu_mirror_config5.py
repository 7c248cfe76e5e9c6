#!/usr/bin/env python3
"""The 1e7-node gap, without a GPU: the oracle's SEQUENTIAL mirror of the tile kernel (orc_tile_layout_q32: the same tiles,
windows, term streams, far-pull rule and launch order, one term after the other — no concurrency inside a window) on the
truncated schedule of tests/golden/config5_cpu_point.json (`-x 15 -G 2`, 1.4e10 terms) from the same initial layout, scored
with the same evaluator.  If the mirror ends where the CPU restatement's Hogwild loop ends (0.115), what the GPU's tile kernel
loses at this size (0.132) is lost to the concurrency of its 256 lanes inside a window; if it ends where the GPU ends, the
tile scheme itself (exclusive windows, capped far pulls from snapshots) loses it.

    python tools/cpu_mirror_config5.py [--nodes N] [--lanes L]     (hours of one CPU core at 1e7 nodes; prints one JSON line)"""
import argparse, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--lanes", type=int, default=256, help="term streams per tile (the GPU's workgroup: 256)")
    ap.add_argument("--iters", type=int, default=15)
    ap.add_argument("--terms-per-step", type=int, default=2)
    ap.add_argument("--order", default="size", help="work items of a launch by decreasing size (sessions of whole windows) or in node order (\"region\")")
    ap.add_argument("--policy", type=lambda v: int(v, 0), default=0, help="orc_tile_layout_q32_ex policy bits (0 = the product's run)")
    ap.add_argument("--hogwild", type=int, default=0, help="also run the CPU restatement's Hogwild loop with this many threads on the same schedule")
    args = ap.parse_args()
    import odgi_amd as oa
    from oracle import oracle as orc
    import pyref
    t0 = time.time()
    g = oa.Graph.synthetic(args.nodes, 50, seed=42)
    og = orc.Graph.from_product(g)
    p = oa.LayoutParams.defaults(g, iter_max=args.iters, min_term_updates=args.terms_per_step * g.n_steps)
    X0, Y0 = oa.initial_layout(g, "d", seed=42)
    R, T = 256, 224
    tiles, items = pyref.build_tiles_py(g.path_first, g.step_handle, R, T, order=args.order)
    tiles["lanes"] = np.minimum(np.full(len(tiles["t0"]), args.lanes, dtype=np.uint32), np.maximum(tiles["n"], 1)).astype(np.uint32)
    # the device's fixed-point frame (pgsgd_session.hip: choose_xform): 2^32 quanta span 8x the larger of extent and longest path
    Xf, Yf = X0.astype(np.float32), Y0.astype(np.float32)
    ends = g.path_first[1:].astype(np.int64) - 1
    max_path_bp = int(max(int(g.step_pos[e]) + int(g.node_len[g.step_handle[e] >> 1]) for e in ends))
    minx, maxx, miny, maxy = float(Xf.min()), float(Xf.max()), float(Yf.min()), float(Yf.max())
    extent = max(maxx - minx, maxy - miny, float(max_path_bp), 1.0)
    span_log2 = int(math.ceil(math.log2(8.0 * extent)))
    span = math.ldexp(1.0, span_log2)
    q = math.ldexp(1.0, 32 - span_log2)
    x_off, y_off = 0.5 * (minx + maxx) - 0.5 * span, 0.5 * (miny + maxy) - 0.5 * span
    print(f"graph + tiles in {time.time() - t0:.0f} s: {len(tiles['t0'])} tiles, {len(items['local'])} items, {int((items['local'] == 0).sum())} window-less; "
          f"{q} quanta per bp; {args.iters} x {p.min_term_updates} terms", file=sys.stderr, flush=True)
    t1 = time.time()
    X, Y, dmax, ck, far = orc.tile_layout_q32(og, orc.params_from(p), p.seed, tiles, items, R, X0, Y0, x_off, y_off, q, policy=args.policy)
    secs = time.time() - t1
    hog = None
    if args.hogwild:
        Xh, Yh, _ = orc.layout_hogwild(og, orc.params_from(p), args.hogwild, X0, Y0, fast=True)
        hog = orc.path_stress_sampled(og, Xh, Yh, 2_000_000, 1)
    out = dict(exp="cpu_mirror_config5", order=args.order, policy=args.policy, hogwild_stress=hog, nodes=g.n_nodes, steps=g.n_steps, iter_max=p.iter_max, min_term_updates=p.min_term_updates, lanes_per_tile=args.lanes,
               quanta_per_bp=q, stress_initial=orc.path_stress_sampled(og, X0, Y0, 2_000_000, 1), stress_final=orc.path_stress_sampled(og, X, Y, 2_000_000, 1),
               far_terms=int(far), checksum_conserved=bool(ck[0] == ck[2] and ck[1] == ck[3]), seconds=secs)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
