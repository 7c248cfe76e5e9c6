#!/usr/bin/env python3
"""BASELINE config 4 in full on the CPU oracle (the restatement of the reference's Hogwild loop, fp64, all host
threads) next to the GPU kernels from the same initial layout, one stress evaluator for all three: the
statistical parity number at the benchmark configuration.  ~10 minutes of host time."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import odgi_amd as oa
from odgi_amd import _lib
from oracle import oracle as orc

def emit(**kw):
    print(json.dumps(kw), flush=True)

g = oa.Graph.synthetic(1_000_000, 50, seed=42)
og = orc.Graph.from_product(g)
X0, Y0 = oa.initial_layout(g, "d", seed=42)
ev = lambda X, Y: dict(stress=oa.path_stress(g, X, Y, 4_000_000, seed=1), stress_seed2=oa.path_stress(g, X, Y, 4_000_000, seed=2),
                       path_distance=oa.path_distance(g, X, Y)[0])
emit(what="initial", **ev(X0, Y0))
for name, flags in (("gpu_tiled", 0), ("gpu_per_lane", _lib.FLAG_NO_TILES)):
    for rep in range(2):
        p = oa.LayoutParams.defaults(g, device=0, flags=flags, seed=9399220 + 7919 * rep)
        X, Y = X0.copy(), Y0.copy()
        st = oa.path_linear_sgd_layout_gpu(g, p, X, Y)
        emit(what=name, rep=rep, kernel_ms=st["kernel_ms"], **ev(X, Y))
p = oa.LayoutParams.defaults(g)
cores = os.cpu_count() or 1
t0 = time.time()
Xo, Yo, st = orc.layout_hogwild(og, orc.params_from(p), cores, X0, Y0, max_seconds=float(os.environ.get("ORACLE_MAX_SECONDS", "0")), fast=True)
emit(what="cpu_oracle_hogwild", threads=cores, terms=st["terms"], iterations=st["iterations"], seconds=st["seconds"],
     terms_per_s=st["terms"] / max(st["seconds"], 1e-9), wall=time.time() - t0, **ev(Xo, Yo))
