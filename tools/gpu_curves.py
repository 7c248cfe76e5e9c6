#!/usr/bin/env python3
"""Stress-vs-iteration curve of one kernel configuration at BASELINE config 4 (same evaluator as the committed CPU curves).
usage: gpu_curves.py <label> [--no-tiles] [--seeds N] [--iters 30]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import odgi_amd as oa
from odgi_amd import _lib
from oracle import oracle as orc

label = sys.argv[1]
flags = _lib.FLAG_NO_TILES if "--no-tiles" in sys.argv else 0
seeds = int(sys.argv[sys.argv.index("--seeds") + 1]) if "--seeds" in sys.argv else 1
iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 30
ref = json.load(open(os.path.join(ROOT, "tests", "golden", "config4_cpu_curves.json")))
g = oa.Graph.synthetic(1_000_000, 50, seed=42)
og = orc.Graph.from_product(g)
snap = [1, 2, 3, 5, 10, 15, 20, 25, 30] if iters == 30 else list(range(1, iters + 1))
out = []
for i in range(seeds):
    X0, Y0 = oa.initial_layout(g, "d", seed=42 + i)
    p = oa.LayoutParams.defaults(g, device=0, flags=flags, seed=9399220 + 7919 * i, iter_max=iters)
    etas = oa.path_linear_sgd_layout_schedule(p)
    cur = []
    with oa.LayoutSession(g, p) as s:
        s.upload(X0, Y0)
        for it in range(p.iter_max):
            s.iteration(etas[it], it >= p.first_cooling_iteration(), p.min_term_updates)
            s.sync()
            if it + 1 in snap:
                X, Y = s.download_f64()
                cur.append(orc.path_stress_sampled(og, X, Y, 1_000_000, ref["eval_seed"]))
        ms, n = s.kernel_time()
        aux = s.aux_time()
    out.append(cur)
print(json.dumps({"label": label, "env": {k: v for k, v in os.environ.items() if k.startswith("PGSGD_")}, "iters": snap,
                  "stress": [[float("%.5g" % v) for v in c] for c in out], "kernel_ms": ms, "aux_ms": aux}))
