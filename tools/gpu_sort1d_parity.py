#!/usr/bin/env python3
"""1D path-guided SGD at scale against the CPU oracle: a 200k-node linear pangenome whose node ranks are shuffled
in blocks of 64 (so the sorter has real work), `odgi sort -Y` defaults (100 iterations x 1*S terms)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import odgi_amd as oa
from odgi_amd import sort as osort
from oracle import oracle as orc

g0 = oa.Graph.synthetic(200_000, 20, seed=11)
rs = np.random.RandomState(3)
n = g0.n_nodes
blocks = np.arange(n).reshape(-1, 64) if n % 64 == 0 else None
perm = np.arange(n)                                    # new rank of old rank
nb = n // 64
order = rs.permutation(nb)
perm[:nb * 64] = (np.repeat(order, 64) * 64 + np.tile(np.arange(64), nb))   # old block i goes to block order[i]
new_len = np.empty_like(g0.node_len); new_len[perm] = g0.node_len
h = g0.step_handle
g = oa.Graph.from_arrays(new_len, g0.path_first, (perm[h >> 1].astype(np.uint32) << 1) | (h & 1))
og = orc.Graph.from_product(g)
true_pos = np.empty(n); true_pos[perm] = np.arange(n)  # node rank -> true position

def quality(X):
    o = np.argsort(X, kind="stable")
    pos = np.empty(n); pos[o] = np.arange(n)
    return float(abs(np.corrcoef(pos, true_pos)[0, 1]))

p = osort.sort_params_defaults(g, seed=9399220)
X0 = osort.sort_initial(g)
print(json.dumps(dict(what="initial", stress=osort.sort_stress(g, X0), order_quality=quality(X0))), flush=True)
t0 = time.time(); Xg, st = osort.path_linear_sgd(g, p); wg = time.time() - t0
print(json.dumps(dict(what="gpu", kernel_ms=st["kernel_ms"], wall_s=wg, terms=int(st["term_updates"]), stress=osort.sort_stress(g, Xg), order_quality=quality(Xg))), flush=True)
Xo, so = orc.sort_hogwild(og, orc.params_from(p), os.cpu_count() or 1, X0, fast=True)
print(json.dumps(dict(what="cpu_oracle_hogwild", threads=os.cpu_count(), seconds=so["seconds"], terms=int(so["terms"]), stress=osort.sort_stress(g, Xo), order_quality=quality(Xo))), flush=True)
