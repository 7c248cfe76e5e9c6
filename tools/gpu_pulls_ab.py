#!/usr/bin/env python3
"""When do a launch's far pulls have to arrive?  PGSGD_PULLS=sync (before the very next launch, rounds 3-4; the drain in front of
that launch), async (before the same colour's next launch; the drain beside the other colour's launch) and the shipped rule
(warm launches sync, cooling launches async): final sampled stress (2e6 pairs, seed 1) of the tile kernel on the whole default
schedule and on a truncated one (-x 15 -G 2), against the per-lane kernel (the reference's rule).  Usage: gpu_pulls_ab.py N [seeds]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PGSGD_DEBUG"] = "1"
import odgi_amd as oa
from odgi_amd import _lib
N = int(float(sys.argv[1]))
seeds = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "9399220").split(",")]
g = oa.Graph.synthetic(N, 50, seed=42)
X0, Y0 = oa.initial_layout(g, "d", seed=42)
for sched, kw in (("default", {}), ("x15_G2", dict(iter_max=15, min_term_updates=2 * g.n_steps))):
    for variant in ("sync", "async", "rule", "per_lane"):
        for seed in seeds:
            os.environ.pop("PGSGD_PULLS", None)
            if variant in ("sync", "async"):
                os.environ["PGSGD_PULLS"] = variant
            p = oa.LayoutParams.defaults(g, device=0, flags=_lib.FLAG_NO_TILES if variant == "per_lane" else 0, **kw)
            p.seed = seed
            etas = oa.path_linear_sgd_layout_schedule(p)
            t0 = time.time()
            with oa.LayoutSession(g, p) as s:
                s.upload(X0, Y0)
                for it in range(p.iter_max):
                    s.iteration(etas[it], it >= p.first_cooling_iteration(), p.min_term_updates)
                    s.sync()
                X, Y = s.download_f64()
                ms = s.kernel_time()[0]
            print(json.dumps(dict(exp="pulls_ab", nodes=N, schedule=sched, variant=variant, seed=seed, stress=oa.path_stress(g, X, Y, 2_000_000, seed=1),
                                  kernel_ms=ms, wall_s=time.time() - t0)), flush=True)
