#!/usr/bin/env python3
"""Small lane-bound graphs: iterations in two passes (every stream the GPU holds samples; one workgroup with the lanes the
busiest node allows moves the ends with the coordinates in LDS) against the pipelined single-pass kernel
(PGSGD_FLAG_NO_SPLIT) on the reference's fixture graphs and small synthetic graphs: kernel time, terms/s, stress (three
seeds); and one stream + one lane, which must agree bit for bit with the single-pass kernel.
Run with PGSGD_DEBUG=1.  PGSGD_SPLIT_MAX_LANES moves the lane count up to which the two passes are taken."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import odgi_amd as oa
from odgi_amd import _lib

G = os.path.join(ROOT, "tests", "golden")
cases = [(n, oa.Graph.from_gfa(os.path.join(G, n + ".gfa"))) for n in ("DRB1-3123", "DRB1-3123_unsorted", "LPA", "chr6.C4")]
for n_nodes, n_paths in ((2000, 8), (8000, 12), (30000, 12)):
    cases.append((f"synthetic-{n_nodes}", oa.Graph.synthetic(n_nodes, n_paths, seed=5)))
label = os.environ.get("SPLIT_AB_LABEL", "")
keep = {k: os.environ.get(k) for k in ("PGSGD_SPLIT_FORCE",)}


def setenv(**kw):
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)


for name, g in cases:
    if not label:
        outs = {}
        for form, flags, env in (("single", _lib.FLAG_NO_TILES | _lib.FLAG_NO_SPLIT, {}), ("split", _lib.FLAG_NO_TILES, dict(PGSGD_SPLIT_FORCE=1))):
            setenv(**env)
            p = oa.LayoutParams.defaults(g, device=0, flags=flags, seed=77)
            p.n_streams = 1
            p.iter_max = 4
            p.min_term_updates = 3000
            X, Y = oa.initial_layout(g, "d", seed=3)
            with oa.LayoutSession(g, p) as sess:
                info = sess.split_info()
            oa.path_linear_sgd_layout_gpu(g, p, X, Y)
            outs[form] = (X.copy(), Y.copy(), info)
            setenv(**keep)
        for form in ("split",):
            same = bool(np.array_equal(outs["single"][0], outs[form][0]) and np.array_equal(outs["single"][1], outs[form][1]))
            print(json.dumps(dict(exp="split_one_stream", graph=name, form=form, info=outs[form][2], identical_to_single_pass=same)), flush=True)
    for mode, flags in (("split", _lib.FLAG_NO_TILES), ("piped", _lib.FLAG_NO_TILES | _lib.FLAG_NO_SPLIT)):
        res = []
        for rep in range(3):
            p = oa.LayoutParams.defaults(g, device=0, flags=flags, seed=9399220 + 7919 * rep)
            X, Y = oa.initial_layout(g, "d", seed=7 + rep)
            with oa.LayoutSession(g, p) as sess:
                info = sess.split_info()
            st = oa.path_linear_sgd_layout_gpu(g, p, X, Y)
            res.append((st["kernel_ms"], st["term_updates"] / (st["kernel_ms"] * 1e-3), oa.path_stress(g, X, Y, 1_000_000, seed=1), st["n_streams"], st["wall_ms"]))
        print(json.dumps(dict(exp="split_ab", label=label, graph=name, nodes=int(g.n_nodes), steps=int(g.n_steps), mode=mode, info=info, streams=res[0][3],
                              kernel_ms=round(float(np.mean([r[0] for r in res])), 3), wall_ms=round(float(np.mean([r[4] for r in res])), 1),
                              terms_per_s=float(np.mean([r[1] for r in res])), stress=[round(r[2], 4) for r in res])), flush=True)
