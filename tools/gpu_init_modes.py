#!/usr/bin/env python3
"""Initial layouts other than the default: -N d (default), u, g, r, h on a sorted 300k-node graph, tile kernel vs
per-lane kernel.  The tile kernel moves a node end over long distances only twice per iteration (capped far
pulls), so does it still form the global structure when the initial layout has none?"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import odgi_amd as oa
from odgi_amd import _lib
g = oa.Graph.synthetic(300_000, 24, seed=7)
for mode in "dugrh":
    X0, Y0 = oa.initial_layout(g, mode, seed=7)
    for name, flags in (("default", 0), ("per_lane", _lib.FLAG_NO_TILES)):
        p = oa.LayoutParams.defaults(g, device=0, flags=flags)
        X, Y = X0.copy(), Y0.copy()
        st = oa.path_linear_sgd_layout_gpu(g, p, X, Y)
        print(json.dumps(dict(exp="init_modes", init=mode, mode=name, kernel_ms=st["kernel_ms"], stress_initial=oa.path_stress(g, X0, Y0, 1_000_000, seed=1),
                              stress=oa.path_stress(g, X, Y, 2_000_000, seed=1), path_distance=oa.path_distance(g, X, Y)[0])), flush=True)
