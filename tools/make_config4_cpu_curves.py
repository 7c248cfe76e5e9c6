#!/usr/bin/env python3
"""Generates tests/golden/config4_cpu_curves.json: stress-vs-iteration curves of the CPU oracle's restatement of the
reference's Hogwild loop (path_sgd_layout.cpp:165-377, fp64) on BASELINE config 4 (synthetic 1M nodes / 50 paths,
seed 42, reference defaults: 30 iterations x 10*S terms, theta 0.99), one run per initial-layout seed.

The whole schedule is 1.4e10 terms per run (592 s on 256 threads of the GPU box, hours on a small host), far too slow
for a test: run once, commit the numbers.  The GPU parity test (tests/test_gpu_parity.py::
test_tile_kernel_against_the_reference_rule_at_config4) evaluates GPU layouts with the same evaluator
(orc.path_stress_sampled, same pair count and evaluator seed) after the same iterations.

    python tools/make_config4_cpu_curves.py [--threads T] [--runs R] [--nodes N] [--out FILE]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SNAP_ITERS = [1, 5, 10, 15, 20, 30]
EVAL_PAIRS = 2_000_000
EVAL_SEED = 0x5eed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=max(1, (os.cpu_count() or 2) - 1))
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--paths", type=int, default=50)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "config4_cpu_curves.json"))
    args = ap.parse_args()
    import odgi_amd as oa
    from oracle import oracle as orc
    g = oa.Graph.synthetic(args.nodes, args.paths, seed=42)
    og = orc.Graph.from_product(g)
    p = oa.LayoutParams.defaults(g)
    out = {"graph": {"nodes": g.n_nodes, "paths": g.n_paths, "steps": g.n_steps, "seed": 42},
           "params": {"iter_max": p.iter_max, "min_term_updates": p.min_term_updates, "theta": p.theta},
           "oracle_hogwild_source_id": __import__("cpu_reference").hogwild_source_id(), "snap_iters": SNAP_ITERS, "eval_pairs": EVAL_PAIRS, "eval_seed": EVAL_SEED, "threads": args.threads, "runs": []}
    if os.path.exists(args.out):
        with open(args.out) as f:
            old = json.load(f)
        if old.get("graph") == out["graph"]:
            out["runs"] = old["runs"]
    for r in range(len(out["runs"]), args.runs):
        init_seed = 42 + r
        X0, Y0 = oa.initial_layout(g, "d", seed=init_seed)
        t = time.time()
        X, Y, st, sx, sy = orc.layout_hogwild_curve(og, orc.params_from(p), args.threads, X0, Y0, SNAP_ITERS, fast=True)
        curve = [orc.path_stress_sampled(og, sx[k], sy[k], EVAL_PAIRS, EVAL_SEED) for k in range(len(SNAP_ITERS))]
        rec = {"init_seed": init_seed, "stress_initial": orc.path_stress_sampled(og, X0, Y0, EVAL_PAIRS, EVAL_SEED),
               "stress_at": curve, "stress_final": orc.path_stress_sampled(og, X, Y, EVAL_PAIRS, EVAL_SEED),
               "terms": st["terms"], "iterations": st["iterations"], "seconds": st["seconds"], "wall": time.time() - t}
        ne = orc.path_stress_near(og, X, Y, zmax=4, threads=args.threads)   # the final layout's exact near-pair figure (round 6)
        rec["near_exact"] = {"near": ne["near"], "by_z": ne["num"].sum(axis=(1, 2)).tolist(), "zero_mass": ne["zero_mass"], "zmax": 4}
        print(json.dumps(rec), flush=True)
        out["runs"].append(rec)
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
