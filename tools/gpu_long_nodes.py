#!/usr/bin/env python3
"""A graph with 30 nodes of 1 Mbp among 300k ordinary ones: per-lane kernel in both coordinate formats and the default
plan, stress binned by path distance, next to the CPU oracle (8+ threads, fast build)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import odgi_amd as oa
from odgi_amd import _lib
from oracle import oracle as orc
rs = np.random.RandomState(21)
n = 300_000
ln = np.minimum(np.maximum(rs.geometric(1 / 32.0, n), 1), 4096).astype(np.uint32)
ln[rs.choice(n, 30, replace=False)] = 1_000_000
g0 = oa.Graph.synthetic(n, 24, seed=7)
g = oa.Graph.from_arrays(ln, g0.path_first, g0.step_handle)
og = orc.Graph.from_product(g)
X0, Y0 = oa.initial_layout(g, "d", seed=7)
# fixed sample of same-path step pairs: half near (within 50 steps), half uniform
pf = g.path_first.astype(np.int64)
M = 2_000_000
ka = rs.randint(0, g.n_steps, M)
path = g.step_path[ka].astype(np.int64)
lo, hi = pf[path], pf[path + 1]
near = rs.rand(M) < 0.5
kb = np.where(near, np.clip(ka + rs.randint(-50, 51, M), lo, hi - 1), lo + (rs.rand(M) * (hi - lo)).astype(np.int64))
ea, eb = g.step_handle[ka].astype(np.int64), g.step_handle[kb].astype(np.int64)
d = np.abs(g.step_pos[ka].astype(np.float64) - g.step_pos[kb].astype(np.float64))
ok = d > 0
bins = [(0, 1e2), (1e2, 1e4), (1e4, 1e6), (1e6, 1e9)]

def binned(X, Y):
    mag = np.hypot(X[ea] - X[eb], Y[ea] - Y[eb])
    e = ((mag - d) / np.where(ok, d, 1)) ** 2
    return {f"d<{hi:g}": float(e[ok & (d >= lo_) & (d < hi)].mean()) for lo_, hi in bins}

print(json.dumps(dict(what="initial", **binned(X0, Y0))), flush=True)
streams = [int(x) for x in os.environ.get("STREAMS", "0").split(",")]
variants = [("per_lane_q32", _lib.FLAG_NO_TILES, 0), ("per_lane_f32", _lib.FLAG_NO_TILES | _lib.FLAG_FP32_ATOMICS, 0), ("default_plan", 0, 0)]
if streams != [0]:
    variants = [(f"per_lane_q32_streams{n_}", _lib.FLAG_NO_TILES, n_) for n_ in streams] + [(f"per_lane_stores_streams{n_}", _lib.FLAG_NO_TILES | _lib.FLAG_HOGWILD_STORES, n_) for n_ in streams[:2]]
for name, flags, ns in variants:
    p = oa.LayoutParams.defaults(g, device=0, flags=flags, n_streams=ns)
    X, Y = X0.copy(), Y0.copy()
    st = oa.path_linear_sgd_layout_gpu(g, p, X, Y)
    print(json.dumps(dict(what=name, kernel_ms=st["kernel_ms"], stress=oa.path_stress(g, X, Y, 2_000_000, seed=1), **binned(X, Y))), flush=True)
if os.environ.get("SKIP_ORACLE"):
    sys.exit(0)
p = oa.LayoutParams.defaults(g)
Xo, Yo, st = orc.layout_hogwild(og, orc.params_from(p), os.cpu_count() or 1, X0, Y0, fast=True)
print(json.dumps(dict(what="cpu_oracle", seconds=st["seconds"], terms=st["terms"], stress=oa.path_stress(g, Xo, Yo, 2_000_000, seed=1), **binned(Xo, Yo))), flush=True)
