#!/usr/bin/env python3
"""Per-lane kernel on the fixture graphs: the hot-node learning-rate cap experiment (PGSGD_FLAG_HOT_NODE_CAP) against
the default lane rule and the CPU restatement.  Three seeds each."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import odgi_amd as oa
from odgi_amd import _lib
from oracle import oracle as orc
import dataclasses

for name in ("DRB1-3123", "LPA", "chr6.C4", "DRB1-3123_unsorted"):
    g = oa.Graph.from_gfa(os.path.join(ROOT, "tests", "golden", name + ".gfa"))
    og = orc.Graph.from_product(g)
    for label, flags in (("hot-node cap (PGSGD_FLAG_HOT_NODE_CAP)", _lib.FLAG_HOT_NODE_CAP), ("lane rule (default)", 0)):
        res, ms, streams = [], [], 0
        for i in range(3):
            X0, Y0 = oa.initial_layout(g, "d", seed=11 + i)
            p = oa.LayoutParams.defaults(g, device=0, flags=flags, seed=9399220 + 7919 * i)
            X, Y = X0.copy(), Y0.copy()
            st = oa.path_linear_sgd_layout_gpu(g, p, X, Y)
            res.append(orc.path_stress_sampled(og, X, Y, 1_000_000))
            ms.append(st["kernel_ms"]); streams = st["n_streams"]
        print(json.dumps({"graph": name, "mode": label, "streams": streams, "stress": [round(v, 4) for v in res], "kernel_ms": [round(v, 2) for v in ms],
                          "terms_per_s": 30 * p.min_term_updates / (np.mean(ms) / 1e3)}), flush=True)
    cpu = []
    for i in range(3):
        X0, Y0 = oa.initial_layout(g, "d", seed=11 + i)
        p = oa.LayoutParams.defaults(g)
        Xo, Yo, _ = orc.layout_hogwild(og, orc.params_from(p), 4, X0, Y0)
        cpu.append(orc.path_stress_sampled(og, Xo, Yo, 1_000_000))
    print(json.dumps({"graph": name, "mode": "CPU restatement, 4 threads", "stress": [round(v, 4) for v in cpu]}), flush=True)
