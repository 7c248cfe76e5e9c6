#!/usr/bin/env python3
"""Generates tests/golden/config5_cpu_point.json: the CPU oracle's restatement of the reference's Hogwild loop
(path_sgd_layout.cpp:120-377) at BASELINE config 5's SIZE — synthetic 1e7 nodes / 50 paths / ~4.7e8 steps, seed 42 — on a
TRUNCATED schedule the CPU can finish: `-x 15 -G 2` (15 iterations of 2*S terms = 1.4e10 terms; the whole default schedule is
1.4e11 terms, hours on 256 threads).  15 iterations is the shortest schedule the product runs the tile kernel on.
The GPU test (tests/test_gpu_parity.py::test_config5_size_schedules_against_the_committed_cpu_points) runs the
same schedule from the same initial layout with both kernels and scores with the same evaluator (2e6 pairs, seed 1).

    python tools/make_config5_cpu_point.py [--threads T] [--runs R] [--nodes N]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SNAP_ITERS = [5, 10, 15]
EVAL_PAIRS = 2_000_000
EVAL_SEED = 1
ITER_MAX = 15
TERMS_PER_STEP = 2


def _hogwild_id():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_reference as cr
    return cr.hogwild_source_id()


def main():
    global ITER_MAX, TERMS_PER_STEP, SNAP_ITERS
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--runs", type=int, default=2)
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "config5_cpu_point.json"))
    ap.add_argument("--iter-max", type=int, default=ITER_MAX)           # round 6: a second, longer point (-x 30 -G 2) in config5_cpu_point_x30.json
    ap.add_argument("--terms-per-step", type=int, default=TERMS_PER_STEP)
    ap.add_argument("--save-layout", default="")                        # .npz of the final X, Y (not committed: 320 MB)
    ap.add_argument("--classes", action="store_true")                   # tools/stress_classes.py of the final layout into the record
    args = ap.parse_args()
    ITER_MAX, TERMS_PER_STEP = args.iter_max, args.terms_per_step
    SNAP_ITERS = [ITER_MAX // 3, 2 * ITER_MAX // 3, ITER_MAX]
    import odgi_amd as oa
    from oracle import oracle as orc
    g = oa.Graph.synthetic(args.nodes, 50, seed=42)
    og = orc.Graph.from_product(g)
    p = oa.LayoutParams.defaults(g, iter_max=ITER_MAX, min_term_updates=TERMS_PER_STEP * g.n_steps)
    out = {"generator": "tools/make_config5_cpu_point.py", "graph": {"nodes": g.n_nodes, "paths": g.n_paths, "steps": g.n_steps, "seed": 42},
           "params": {"iter_max": p.iter_max, "min_term_updates": p.min_term_updates, "theta": p.theta, "cooling_start": p.cooling_start},
           "oracle_hogwild_source_id": _hogwild_id(), "snap_iters": SNAP_ITERS, "eval_pairs": EVAL_PAIRS, "eval_seed": EVAL_SEED, "threads": args.threads, "runs": []}
    if os.path.exists(args.out):
        with open(args.out) as f:
            old = json.load(f)
        if old.get("graph") == out["graph"] and old.get("params") == out["params"]:
            out["runs"] = old["runs"]
            for run in out["runs"]:   # (runs rolled before the thread count was kept per run)
                run.setdefault("threads", old.get("threads"))
    for r in range(len(out["runs"]), args.runs):
        init_seed = 42 + r
        X0, Y0 = oa.initial_layout(g, "d", seed=init_seed)
        t = time.time()
        X, Y, st, sx, sy = orc.layout_hogwild_curve(og, orc.params_from(p), args.threads, X0, Y0, SNAP_ITERS, fast=True)
        rec = {"init_seed": init_seed, "stress_initial": orc.path_stress_sampled(og, X0, Y0, EVAL_PAIRS, EVAL_SEED),
               "stress_at": [orc.path_stress_sampled(og, sx[k], sy[k], EVAL_PAIRS, EVAL_SEED) for k in range(len(SNAP_ITERS))],
               "stress_final": orc.path_stress_sampled(og, X, Y, EVAL_PAIRS, EVAL_SEED),
               "terms": st["terms"], "iterations": st["iterations"], "seconds": st["seconds"], "wall": time.time() - t, "threads": args.threads}
        ne = orc.path_stress_near(og, X, Y, zmax=4, threads=args.threads)   # no sampling error (oracle/pgsgd_oracle.c: orc_path_stress_near)
        rec["near_exact"] = {"near": ne["near"], "by_z": ne["num"].sum(axis=(1, 2)).tolist(), "zero_mass": ne["zero_mass"], "zmax": 4}
        if args.classes:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import stress_classes
            rec["classes"] = stress_classes.classes(g, X, Y)
        if args.save_layout:
            import numpy as np
            np.savez(args.save_layout + f".run{r}.npz", X=X, Y=Y)
        print(json.dumps(rec), flush=True)
        out["runs"].append(rec)
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
