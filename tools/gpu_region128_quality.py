#!/usr/bin/env python3
"""128-node regions (PGSGD_FLAG_REGION_128, what multi-GPU sessions of mid-sized graphs are created with) against 256-node ones on ONE
GPU: the exact near-pair figure and the sampled stress of the final layout, three seeds.  gpu_region128_quality.py NODES PATHS TERMS_PER_STEP"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import odgi_amd as oa
from odgi_amd import _lib
N, P, K = int(float(sys.argv[1])), int(sys.argv[2]), int(sys.argv[3])
g = oa.Graph.synthetic(N, P, seed=7)
for rep in range(3):
    X0, Y0 = oa.initial_layout(g, "d", seed=7 + rep)
    for flags in (0, _lib.FLAG_REGION_128):
        p = oa.LayoutParams.defaults(g, device=0, seed=9399220 + 7919 * rep, min_term_updates=K * g.n_steps, flags=flags)
        X, Y = X0.copy(), Y0.copy()
        st = oa.path_linear_sgd_layout_gpu(g, p, X, Y)
        print(json.dumps(dict(exp="region128_quality", nodes=N, paths=P, terms_per_step=K, rep=rep, region=128 if flags else 256,
                              near_exact=oa.path_stress_near(g, X, Y, zmax=4)["near"], stress=oa.path_stress(g, X, Y, 1_000_000, seed=1),
                              stress_seed2=oa.path_stress(g, X, Y, 1_000_000, seed=2), kernel_ms=st.get("kernel_ms"))), flush=True)
