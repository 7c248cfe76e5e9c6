#!/usr/bin/env python3
"""Regenerates tests/golden/golden_vectors.npz from the CPU oracle (oracle/pgsgd_oracle.c).

The upstream reference cannot run here (its deps/ are empty) and has no golden vectors for the
layout path, so these vectors are the oracle's own output, committed so that (a) any later change
to the oracle is caught and (b) the GPU sampler can be checked on a box without re-deriving them.
Contents: learning-rate schedules and zeta tables for the three fixture graphs'
default parameters, and the first 1000 sampled terms (ka, kb, off_a, off_b) of stream seed 9399220
on DRB1-3123 in non-cooling and cooling mode; final words of the tile-kernel mirror on DRB1-3123.
Usage:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import parse_gfa_py  # noqa: E402
from oracle import oracle as orc  # noqa: E402

GOLDEN = os.path.dirname(os.path.abspath(__file__))
out = {}
for name in ("DRB1-3123", "LPA", "chr6.C4"):
    d = parse_gfa_py(os.path.join(GOLDEN, name + ".gfa"))
    counts = np.diff(d["path_first"].astype(np.int64))
    max_steps = int(counts.max())
    p = orc.params(iter_max=30, iter_with_max_learning_rate=0, min_term_updates=10 * len(d["step_handle"]),
                   delta=0.0, eps=0.01, eta_max=float(max_steps) ** 2, theta=0.99, space=max_steps,
                   space_max=1000, space_quantization_step=100, cooling_start=0.5)
    out[f"etas/{name}"] = orc.schedule(p)
    out[f"zetas/{name}"] = orc.zetas(0.99, max_steps, 1000, 100)
    if name == "DRB1-3123":
        g = orc.Graph(d["node_len"], d["path_first"], d["step_path"], d["step_handle"], d["step_pos"])
        out["terms/DRB1-3123/warm"] = orc.trace_terms(g, p, 9399220, 1, 0, False, 1000)[:, 0, :]
        out["terms/DRB1-3123/cooling"] = orc.trace_terms(g, p, 9399220, 1, 0, True, 1000)[:, 0, :]
out["zetas/theta0.5_space2932"] = orc.zetas(0.5, 2932, 1000, 100)

# The sequential mirror of the tile kernel (one workgroup, one lane per tile) on DRB1-3123, whose unsorted
# stretches give window-less tiles: tile table from the Python restatement (tests/pyref.py), region 64,
# 6 iterations of 2*S terms, frame 16 quanta per bp.  The GPU test checks the kernel against the same
# mirror with the product's own tile table and frame; this pins the mirror itself.
import pyref  # noqa: E402
from test_oracle_pins import tile_mirror_case  # noqa: E402
out.update({f"tile_mirror/{k}": v for k, v in tile_mirror_case(orc, pyref).items()})
# the same case in round 2's launch order (far pulls delivered right after their launch, two snapshots per warm
# iteration): the vectors committed in round 2 as tile_mirror/*, unchanged
out.update({f"tile_mirror_r2/{k}": v for k, v in tile_mirror_case(orc, pyref, orc.TILE_ROUND2).items()})
# ... and with round 3's pipeline (today's launch order, the Zipf/uniform coin of a warm term per lane instead of per
# wave and trip): the vectors committed in round 3 as tile_mirror/*, unchanged
out.update({f"tile_mirror_r3/{k}": v for k, v in tile_mirror_case(orc, pyref, orc.TILE_ROUND3).items()})
# ... and with rounds 4-5's (wave coin, partner pairs, far pulls ramping 0.1 .. to HALF a projection; round 6 ramps 0.2 .. to
# one): the vectors committed in round 5 as tile_mirror/*, unchanged
out.update({f"tile_mirror_r5/{k}": v for k, v in tile_mirror_case(orc, pyref, orc.TILE_ROUND5).items()})
# a regeneration may add arrays and replace tile_mirror/* when the shipped pipeline changes; everything else must come
# out as committed
path = os.path.join(GOLDEN, "golden_vectors.npz")
if os.path.exists(path):
    old = np.load(path)
    for k in old.files:
        if k.startswith("tile_mirror/"):
            continue
        assert k in out and np.array_equal(old[k], out[k]), f"{k} changed"
np.savez_compressed(path, **out)
print("wrote", len(out), "arrays")
