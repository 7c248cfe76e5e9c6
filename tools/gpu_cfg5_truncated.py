#!/usr/bin/env python3
"""The truncated schedule of tests/golden/config5_cpu_point.json (`-x 15 -G 2`) on the GPU for variants of the tile kernel's
work order and snapshot policy (debug knobs), against the per-lane kernel: where does a SHORT schedule on a large graph lose its
layout?  Usage: gpu_cfg5_truncated.py N variant ...   variant = name[+env:KNOB=VALUE...][@sampler-seed]; prints JSON lines."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PGSGD_DEBUG"] = "1"
import odgi_amd as oa
from odgi_amd import _lib
N = int(float(sys.argv[1]))
g = oa.Graph.synthetic(N, 50, seed=42)
X0, Y0 = oa.initial_layout(g, "d", seed=42)
set_before = []
for v_in in sys.argv[2:]:
    for k in set_before:
        os.environ.pop(k, None)
    set_before = []
    v, seed = v_in, 9399220
    if "@" in v:
        v, sd = v.rsplit("@", 1)
        seed = int(sd)
    parts = v.split("+env:")
    for kv in parts[1:]:
        k, val = kv.split("=", 1)
        os.environ[k] = val
        set_before.append(k)
    flags = _lib.FLAG_NO_TILES if parts[0] == "per_lane" else 0
    p = oa.LayoutParams.defaults(g, device=0, flags=flags, iter_max=15, min_term_updates=2 * g.n_steps)
    p.seed = seed
    etas = oa.path_linear_sgd_layout_schedule(p)
    out = {}
    with oa.LayoutSession(g, p) as s:
        s.upload(X0, Y0)
        info = s.tile_info()
        for it in range(p.iter_max):
            s.iteration(etas[it], it >= p.first_cooling_iteration(), p.min_term_updates)
            s.sync()
            if it + 1 in (5, 10, 15):
                X, Y = s.download_f64(flush=it + 1 == 15)
                out[it + 1] = oa.path_stress(g, X, Y, 2_000_000, seed=1)
        ms = s.kernel_time()[0] + sum(s.aux_time())
    print(json.dumps(dict(exp="cfg5_truncated", nodes=N, variant=v_in, tiled=bool(info["tiled"]), parts=info.get("parts"), xcd_runs=info.get("xcd_runs"),
                          stress_after_5_10_15=[out[5], out[10], out[15]], kernel_ms=ms)), flush=True)
