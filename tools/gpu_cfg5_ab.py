#!/usr/bin/env python3
"""Where the tile kernel's final stress differs from the per-lane kernel's (the reference's rule) at 1e6 and 1e7 nodes: the
whole 30-iteration schedule from the same initial layout, one evaluator, for variants of the far pulls' treatment —
default; constant under-relaxation r (PGSGD_TILE_FAR_RELAX); no learning-rate cap of far terms (PGSGD_FLAG_NO_FAR_CAP); a
snapshot pass per iteration instead of the tiles' own record rewrites (PGSGD_TILE_SNAPSHOT_PASS); exact math / no pairs.
Usage: gpu_cfg5_ab.py N [variant ...]   prints one JSON line per variant."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PGSGD_DEBUG"] = "1"
import numpy as np
import odgi_amd as oa
from odgi_amd import _lib

N = int(float(sys.argv[1]))
variants = sys.argv[2:] or ["tile", "per_lane"]
g = oa.Graph.synthetic(N, 50, seed=42)
X0, Y0 = oa.initial_layout(g, "d", seed=42)
KNOBS = ("PGSGD_TILE_FAR_RELAX", "PGSGD_TILE_SNAPSHOT_PASS", "PGSGD_TILE_LANES", "PGSGD_FRAME_SPAN", "PGSGD_TILE_LOCK_MU", "PGSGD_TILE_SUBSTEPS", "PGSGD_TILE_LANE_COIN")
EXTRA = []
for v_in in variants:
    v = v_in
    for k in KNOBS:
        os.environ.pop(k, None)
    flags = 0
    seed = 9399220
    for k in list(EXTRA):
        os.environ.pop(k, None)
    EXTRA.clear()
    while "+env:" in v:   # e.g. tile+env:PGSGD_TILE_SPLIT=13@11: any debug knob of the library
        v, rest = v.split("+env:", 1)
        kv, tail = (rest.split("@", 1) + [""])[:2] if "@" in rest and "+env:" not in rest else (rest, "")
        if "+env:" in kv:
            kv, more = kv.split("+env:", 1)
            v = v + "+env:" + more
        name, val = kv.split("=", 1)
        os.environ[name] = val
        EXTRA.append(name)
        if tail:
            v = v + "@" + tail
    if "+span" in v:   # e.g. tile+span128@11: a frame 128x the extent instead of 8x (16x coarser quanta)
        v, rest = v.split("+span")
        os.environ["PGSGD_FRAME_SPAN"] = rest.split("@")[0]
        v = v + ("@" + rest.split("@")[1] if "@" in rest else "")
    if "@" in v:       # variant@sampler-seed
        v, sd = v.split("@")
        seed = int(sd)
    if v == "per_lane": flags = _lib.FLAG_NO_TILES
    elif v.startswith("relax"): os.environ["PGSGD_TILE_FAR_RELAX"] = v[5:]
    elif v == "nocap": flags = _lib.FLAG_NO_FAR_CAP
    elif v == "pass": os.environ["PGSGD_TILE_SNAPSHOT_PASS"] = "1"
    elif v == "exact": flags = _lib.FLAG_EXACT_MATH
    elif v == "nopairs": flags = _lib.FLAG_NO_PARTNER_PAIRS
    elif v == "lanecoin": os.environ["PGSGD_TILE_LANE_COIN"] = "1"      # the Zipf/uniform coin per lane, as in round 3 (implies no partner pairs)
    elif v.startswith("lanes"): os.environ["PGSGD_TILE_LANES"] = v[5:]
    elif v.startswith("lock"): os.environ["PGSGD_TILE_LOCK_MU"] = v[4:]
    elif v.startswith("sub"): os.environ["PGSGD_TILE_SUBSTEPS"] = v[3:]
    n_streams = 0
    if v.startswith("pl"):   # per-lane kernel with this many streams (lanes), e.g. pl45875@11
        flags = _lib.FLAG_NO_TILES
        n_streams = int(v[2:])
    p = oa.LayoutParams.defaults(g, device=0, flags=flags, n_streams=n_streams)
    p.seed = seed
    etas = oa.path_linear_sgd_layout_schedule(p)
    t0 = time.time()
    out = {}
    with oa.LayoutSession(g, p) as s:
        s.upload(X0, Y0)
        for it in range(p.iter_max):
            s.iteration(etas[it], it >= p.first_cooling_iteration(), p.min_term_updates)
            s.sync()
            if it + 1 in (10, 20, 30):
                X, Y = s.download_f64(flush=it + 1 == 30)
                out[it + 1] = oa.path_stress(g, X, Y, 2_000_000, seed=1)
        ms = s.kernel_time()[0] + sum(s.aux_time())
        tiled = s.tile_info()["tiled"]
        q_per_bp, doublings = s.coord_format()[3], s.frame_status()[1]
        locked, lost = s.tile_conflicts()
    # where the stress sits: pairs of steps at fixed step distances along a path (layout distance of the steps' first node
    # ends against their path distance), 400k pairs per class, one sampler for every variant
    rs = np.random.RandomState(5)
    first = g.path_first.astype(np.int64)
    cnt = np.diff(first)
    by_class = {}
    for name_c, lo, hi in (("adjacent", 1, 1), ("2-30", 2, 30), ("31-1000", 31, 1000), ("1e3-1e5", 1000, 100000), ("uniform", 0, 0)):
        pth = rs.choice(len(cnt), 400000, p=cnt / cnt.sum())
        ka = first[pth] + (rs.rand(400000) * cnt[pth]).astype(np.int64)
        if name_c == "uniform":
            kb = first[pth] + (rs.rand(400000) * cnt[pth]).astype(np.int64)
        else:
            kb = ka + rs.randint(lo, hi + 1, 400000)
        ok = (kb < first[pth] + cnt[pth]) & (kb != ka)
        ka, kb = ka[ok], kb[ok]
        ha, hb = g.step_handle[ka].astype(np.int64), g.step_handle[kb].astype(np.int64)
        d = np.abs(g.step_pos[kb].astype(np.float64) - g.step_pos[ka].astype(np.float64))
        okd = d > 0
        mag = np.hypot(X[ha] - X[hb], Y[ha] - Y[hb])[okd]
        by_class[name_c] = float(np.mean(((mag - d[okd]) / d[okd]) ** 2))
    print(json.dumps({"nodes": N, "stress_by_step_distance": by_class, "variant": v_in, "tiled": bool(tiled), "stress_after_10_20_30": [out[10], out[20], out[30]], "kernel_ms": ms, "wall_s": time.time() - t0,
                      "quanta_per_bp_at_end": q_per_bp, "frame_doublings": doublings, "frame_span": os.environ.get("PGSGD_FRAME_SPAN"), "locked_terms": locked, "lost_terms": lost}), flush=True)
