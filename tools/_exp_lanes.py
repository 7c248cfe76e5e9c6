import json, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import odgi_amd as oa
from odgi_amd import _lib
from odgi_amd.sort import path_linear_sgd, sort_params_defaults, sort_stress
G = "/root/repo/tests/golden"
for name in ("LPA", "chr6.C4", "DRB1-3123_unsorted"):
    g = oa.Graph.from_gfa(os.path.join(G, name + ".gfa"))
    for lanes in ("single", 1024, 512, 256, 128):
        if lanes == "single":
            flags = _lib.FLAG_NO_SPLIT
            os.environ.pop("PGSGD_SPLIT_APPLY_LANES", None)
        else:
            flags = 0
            os.environ["PGSGD_SPLIT_APPLY_LANES"] = str(lanes)
        s2, k2, s1, k1 = [], [], [], []
        for rep in range(5):
            p = oa.LayoutParams.defaults(g, device=0, flags=flags | _lib.FLAG_NO_TILES, seed=9399220 + 7919 * rep)
            X, Y = oa.initial_layout(g, "d", seed=7 + rep)
            st = oa.path_linear_sgd_layout_gpu(g, p, X, Y)
            s2.append(oa.path_stress(g, X, Y, 1_000_000, seed=1)); k2.append(st["kernel_ms"])
            al = st["apply_lanes"]
            p1 = sort_params_defaults(g, device=0, flags=flags, seed=9399220 + 7919 * rep)
            X1, st1 = path_linear_sgd(g, p1)
            s1.append(sort_stress(g, X1, 300000, seed=5)); k1.append(st1["kernel_ms"])
        print(json.dumps(dict(exp="split_apply_lanes", graph=name, lanes=lanes, apply_lanes=al, layout_ms=round(float(np.mean(k2)), 2), layout_stress=[round(v, 4) for v in s2],
                              sort_ms=round(float(np.mean(k1)), 2), sort_stress=[round(v, 3) for v in s1])), flush=True)
