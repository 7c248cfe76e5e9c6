#!/usr/bin/env python3
"""Run-to-run spread of layout quality vs terms_per_anchor (and update mode) on the fixture graphs.
Each cell: 6 runs with different initial-layout and sampler seeds; sampled path stress by the oracle."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import odgi_amd as oa
from oracle import oracle as orc
for name in ("DRB1-3123_unsorted", "DRB1-3123", "chr6.C4", "LPA"):
    g = oa.Graph.from_gfa(os.path.join(ROOT, "tests", "golden", name + ".gfa"))
    og = orc.Graph.from_product(g)
    rows = {}
    cpu = []
    for rep in range(6):
        X0, Y0 = oa.initial_layout(g, "d", seed=100 + rep)
        p = oa.LayoutParams.defaults(g, device=0)
        Xo, Yo, _ = orc.layout_hogwild(og, orc.params_from(p), 8, X0, Y0)
        cpu.append(orc.path_stress_sampled(og, Xo, Yo, 1_000_000))
        for flags in (0, 4):
            for m in (1, 2, 4, 8):
                p = oa.LayoutParams.defaults(g, device=0, terms_per_anchor=m, seed=9399220 + 1000 * rep, flags=flags)
                X, Y = X0.copy(), Y0.copy()
                oa.path_linear_sgd_layout_gpu(g, p, X, Y)
                rows.setdefault(f"flags{flags}_m{m}", []).append(orc.path_stress_sampled(og, X, Y, 1_000_000))
    out = {"exp": "replicates", "graph": name, "cpu_oracle": [float(np.mean(cpu)), float(np.std(cpu))]}
    for k, v in rows.items():
        out[k] = [float(np.mean(v)), float(np.std(v))]
    print(json.dumps(out), flush=True)
