#!/usr/bin/env python3
"""Small graphs run the per-lane kernel with as many lanes as their busiest node allows: L = c * S / (most steps on one node),
c = 2 concurrent terms on that node (pgsgd_session.hip: auto_streams; round 1 measured layouts diverging from c = 4-16).
A whole layout of a fixture graph is then 5-50 ms of a mostly idle GPU.  How far can c rise before the layout leaves the band
of the committed CPU distribution (tests/golden/cpu_reference_distributions.json)?  Three layouts per graph and c,
reference defaults.  Prints JSON lines."""
import dataclasses, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["PGSGD_DEBUG"] = "1"
import numpy as np
import odgi_amd as oa
import cpu_reference as cr
from oracle import oracle as orc
caps = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2,4,8,16,32").split(",")]
for name in ("DRB1-3123", "LPA", "chr6.C4"):
    g = oa.Graph.from_gfa(os.path.join(cr.GOLDEN, name + ".gfa"), threads=4)
    og = orc.Graph.from_product(g)
    p0 = oa.LayoutParams.defaults(g, device=0)
    dist = cr.entry(name, p0)["stress"]
    lo, hi = cr.band(dist)
    for c in caps:
        os.environ["PGSGD_LANE_CAP"] = str(c)
        for no_split in (0, 1):
            vals, ms, streams = [], [], 0
            for i, seed in enumerate(cr.INIT_SEEDS):
                X, Y = oa.initial_layout(g, "d", seed=seed)
                st = oa.path_linear_sgd_layout_gpu(g, dataclasses.replace(p0, seed=p0.seed + 7919 * i, flags=0x1000 if no_split else 0), X, Y)
                vals.append(orc.path_stress_sampled(og, X, Y, cr.EVAL_PAIRS)); ms.append(st["kernel_ms"]); streams = st["n_streams"]
            m = float(np.mean(vals))
            print(json.dumps(dict(exp="lane_cap", graph=name, concurrent_terms_on_busiest_node=c, single_pass=bool(no_split), streams=streams, kernel_ms=float(np.mean(ms)),
                                  stress=[round(v, 4) for v in vals], mean=m, cpu_median=dist["median"], band=[lo, hi], inside=bool(lo <= m <= hi))), flush=True)
