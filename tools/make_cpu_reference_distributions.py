#!/usr/bin/env python3
"""Generates tests/golden/cpu_reference_distributions.json: for every configuration a statistical GPU test compares
against, the distribution (>= 8 runs: every run, mean, sigma, min, max, median, robust sigma) of the CPU oracle's
restatement of the reference's Hogwild loop (oracle/pgsgd_oracle.c: orc_layout_hogwild = path_sgd_layout.cpp:120-377),
scored with the evaluators the tests score GPU layouts with (orc.path_stress_sampled with 1e6 pairs and its default
seed; orc.path_distance = `odgi stats -s`).

The loop is non-deterministic by construction (thread timing), so the GPU box must not re-roll it: roll it here, once,
commit the numbers (tests/cpu_reference.py reads them).  Configurations:
  * BASELINE configs 1-3 with the reference's defaults (DRB1-3123, LPA, chr6.C4; initial layouts -N d, seeds 11/12/13),
    4 worker threads (what the tests used when they rolled it live);
  * the 5000-short-paths graph of test_many_paths_use_the_global_path_table;
  * config 3's theta x -K sweep on chr6.C4 from the deterministic Hilbert initial layout (-N h);
  * the 300k-node synthetic pangenome of test_tiled_kernel_matches_per_lane_kernel_and_oracle (3*S terms per
    iteration, seeds 7/8/9), all cores;
  * 1D PG-SGD (`odgi sort -Y`) on the fixture graphs and on a shuffled 20000-node linear graph (tests/test_sort_1d.py).

    python tools/make_cpu_reference_distributions.py [--runs 9] [--only SUBSTRING] [--threads-large T]
Existing entries with enough runs are kept (delete the file to start over)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=9)
    ap.add_argument("--only", default="")
    ap.add_argument("--threads-large", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--exact", default="", help="entries whose key contains this get runs scored with the exact near-pair evaluator (entry['near_exact'])")
    args = ap.parse_args()
    import numpy as np
    import odgi_amd as oa
    from oracle import oracle as orc
    import cpu_reference as cr
    orc.build()

    def fixture(name):
        return oa.Graph.from_gfa(os.path.join(cr.GOLDEN, name + ".gfa"), threads=4)

    configs = []   # (name, graph factory, params factory, init, init seeds, threads, fast)
    for name in ("DRB1-3123", "LPA", "chr6.C4"):
        configs.append((name, (lambda n=name: fixture(n)), (lambda g: oa.LayoutParams.defaults(g)), "d", cr.INIT_SEEDS, 4, False))
    configs.append(("5000-paths", (lambda: cr.many_paths_graph(oa)), (lambda g: oa.LayoutParams.defaults(g)), "d", cr.INIT_SEEDS, 4, False))
    for theta in (0.5, 0.9, 0.99, 0.999):
        for K in (0.25, 0.5, 0.75):
            configs.append(("chr6.C4", (lambda: fixture("chr6.C4")), (lambda g, t=theta, k=K: oa.LayoutParams.defaults(g, theta=t, cooling_start=k)),
                            "h", cr.INIT_SEEDS, 4, False))
    configs.append(("synthetic-300k", (lambda: cr.synthetic_300k(oa)), (lambda g: oa.LayoutParams.defaults(g, min_term_updates=3 * g.n_steps)),
                    "d", (7, 8, 9), args.threads_large, True))

    if os.path.exists(cr.PATH):
        with open(cr.PATH) as f:
            db = json.load(f)
    else:
        db = {"generator": "tools/make_cpu_reference_distributions.py",
              "what": "sampled path stress (1e6 pairs, evaluator seed 0x5eed) and `odgi stats -s` path distance per node of "
                      "the CPU restatement's Hogwild layouts (oracle/pgsgd_oracle.c: orc_layout_hogwild)",
              "eval_pairs": cr.EVAL_PAIRS, "host": {"cpus": os.cpu_count()}, "entries": {}}
    # entries rolled with other oracle sources are stale: a file that mixes ids is refused rather than stamped over
    if db.get("oracle_hogwild_source_id", cr.hogwild_source_id()) != cr.hogwild_source_id():
        raise SystemExit(f"{cr.PATH} was rolled with oracle sources {db['oracle_hogwild_source_id']}, today's are {cr.hogwild_source_id()}: delete it and roll every entry again")
    db["oracle_hogwild_source_id"] = cr.hogwild_source_id()
    graphs = {}
    for name, gf, pf, init, seeds, threads, fast in configs:
        if name not in graphs:
            g = gf()
            graphs[name] = (g, orc.Graph.from_product(g))
        g, og = graphs[name]
        p = pf(g)
        k = cr.key(name, p, init)
        if args.only and args.only not in k:
            continue
        e = db["entries"].get(k)
        if e and e["stress"]["n"] >= args.runs:
            continue
        stress = list(e["stress"]["runs"]) if e else []
        dist = list(e["path_distance"]["runs"]) if e else []
        used_seeds = list(e["init_seeds"]) if e else []
        secs = list(e["seconds"]) if e else []
        while len(stress) < args.runs:
            seed = seeds[len(stress) % len(seeds)]
            X0, Y0 = oa.initial_layout(g, init, seed=seed)
            t = time.time()
            Xo, Yo, st = orc.layout_hogwild(og, orc.params_from(p), threads, X0, Y0, fast=fast)
            assert st["iterations"] == p.iter_max and st["terms"] >= p.iter_max * p.min_term_updates
            stress.append(orc.path_stress_sampled(og, Xo, Yo, cr.EVAL_PAIRS))
            dist.append(orc.path_distance(og, Xo, Yo)[0])
            used_seeds.append(seed)
            secs.append(round(time.time() - t, 2))
            print(f"{k}: run {len(stress)} seed {seed} stress {stress[-1]:.4f} path distance {dist[-1]:.3f} ({secs[-1]} s)", flush=True)
            db["entries"][k] = {"graph": name, "nodes": g.n_nodes, "steps": g.n_steps, "paths": g.n_paths, "init": init, "init_seeds": used_seeds,
                                "threads": threads, "fast_build": fast, "seconds": secs,
                                "stress": cr.summarize(stress), "path_distance": cr.summarize(dist)}
            with open(cr.PATH + ".tmp", "w") as f:
                json.dump(db, f, indent=1)
            os.replace(cr.PATH + ".tmp", cr.PATH)
    # --exact SUBSTRING (round 6): for the matching 2D entries, `runs` MORE runs scored with the evaluator that has no sampling error
    # (orc.path_stress_near, zmax 4) -> entry["near_exact"]; the sampled distributions above are left as they are
    if args.exact:
        for name, gf, pf, init, seeds, threads, fast in configs:
            if name not in graphs:
                g = gf()
                graphs[name] = (g, orc.Graph.from_product(g))
            g, og = graphs[name]
            p = pf(g)
            k = cr.key(name, p, init)
            e = db["entries"].get(k)
            if args.exact not in k or not e or e.get("near_exact", {}).get("n", 0) >= args.runs:
                continue
            vals = list(e.get("near_exact", {}).get("runs", []))
            while len(vals) < args.runs:
                seed = seeds[len(vals) % len(seeds)]
                X0, Y0 = oa.initial_layout(g, init, seed=seed)
                Xo, Yo, st = orc.layout_hogwild(og, orc.params_from(p), threads, X0, Y0, fast=fast)
                vals.append(orc.path_stress_near(og, Xo, Yo, zmax=4, threads=threads)["near"])
                print(f"{k}: exact run {len(vals)} seed {seed} near-pair figure {vals[-1]:.5f}", flush=True)
                e["near_exact"] = cr.summarize(vals)
                with open(cr.PATH + ".tmp", "w") as f:
                    json.dump(db, f, indent=1)
                os.replace(cr.PATH + ".tmp", cr.PATH)
    # 1D PG-SGD (`odgi sort -Y`, path_sgd.cpp): orc.sort_hogwild, scored with orc.sort_stress
    from odgi_amd.sort import sort_params_defaults
    sort_configs = [("1d:" + n, (lambda n=n: (fixture(n), None)), (lambda g: sort_params_defaults(g)), 300_000) for n in ("DRB1-3123", "DRB1-3123_unsorted", "chr6.C4")]
    sort_configs.append(("1d:shuffled-20000", (lambda: cr.shuffled_linear_graph(oa, n_nodes=20000, n_paths=8)),
                         (lambda g: sort_params_defaults(g, iter_max=30, min_term_updates=10 * g.n_steps)), 500_000))
    for name, gf, pf, pairs in sort_configs:
        g, true_order = gf()
        og = orc.Graph.from_product(g)
        p = pf(g)
        k = cr.key(name, p, "1d")
        if args.only and args.only not in k:
            continue
        e = db["entries"].get(k)
        if e and e["stress"]["n"] >= args.runs:
            continue
        stress = list(e["stress"]["runs"]) if e else []
        quality = list(e["order_quality"]["runs"]) if e and "order_quality" in e else []
        while len(stress) < args.runs:
            t = time.time()
            Xo, st = orc.sort_hogwild(og, orc.params_from(p), 4, orc.sort_initial(og))
            stress.append(orc.sort_stress(og, Xo, pairs))
            if true_order is not None:
                quality.append(cr.order_quality(np.argsort(Xo, kind="stable"), true_order))
            print(f"{k}: run {len(stress)} 1D stress {stress[-1]:.4f} ({time.time() - t:.1f} s)", flush=True)
            db["entries"][k] = {"graph": name, "nodes": g.n_nodes, "steps": g.n_steps, "paths": g.n_paths, "init": "1d", "threads": 4, "eval_pairs": pairs,
                                "stress": cr.summarize(stress)}
            if quality:
                db["entries"][k]["order_quality"] = cr.summarize(quality)
            with open(cr.PATH + ".tmp", "w") as f:
                json.dump(db, f, indent=1)
            os.replace(cr.PATH + ".tmp", cr.PATH)
    for k, e in db["entries"].items():
        s = e["stress"]
        print(f"{k}\n    stress n {s['n']} median {s['median']:.4f} mean {s['mean']:.4f} sigma {s['sigma']:.4f} robust {s['sigma_robust']:.4f} "
              f"min {s['min']:.4f} max {s['max']:.4f}")


if __name__ == "__main__":
    main()
