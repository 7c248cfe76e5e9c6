#!/usr/bin/env python3
"""Round 6: WHERE the tile kernel's final stress differs from the per-lane kernel's at 1e7 nodes.  Every variant runs the given
schedule from one initial layout; its final layout is scored by the tests' evaluator (2e6 pairs, seed 1) and decomposed by
tools/stress_classes.py on pairs that depend on the graph alone — the same pairs for every variant, for the CPU oracle's
layout (tools/make_config5_cpu_point.py --classes) and on every machine.

    gpu_gap_classes.py N ITER_MAX TERMS_PER_STEP variant ...      variant = name[+env:KNOB=VALUE...][@sampler-seed]
      name: tile | per_lane | until<k> (per-lane kernel from iteration k on) | lanes<k>
    prints one JSON line per variant (stress curve over the last iterations, classes) and a table against the first variant."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["PGSGD_DEBUG"] = "1"
import numpy as np
import odgi_amd as oa
from odgi_amd import _lib
import stress_classes

N, ITER_MAX, TPS = int(float(sys.argv[1])), int(sys.argv[2]), int(sys.argv[3])
CURVE_FROM = int(os.environ.get("GAP_CURVE_FROM", str(ITER_MAX - 10)))
g = oa.Graph.synthetic(N, 50, seed=42)
X0, Y0 = oa.initial_layout(g, "d", seed=42)
results = []
ms_of = {}
set_before = []
for v_in in sys.argv[4:]:
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
    name = parts[0]
    flags = 0
    if name == "per_lane":
        flags = _lib.FLAG_NO_TILES
    elif name.startswith("until"):
        os.environ["PGSGD_TILE_UNTIL"] = name[5:]
        set_before.append("PGSGD_TILE_UNTIL")
    elif name == "sync":      # tile kernel, every launch's far pulls delivered in front of the very next launch (PGSGD_FLAG_SYNC_DRAIN)
        flags = _lib.FLAG_SYNC_DRAIN
    elif name == "nocap":
        flags = _lib.FLAG_NO_FAR_CAP
    elif name.startswith("lanes"):
        os.environ["PGSGD_TILE_LANES"] = name[5:]
        set_before.append("PGSGD_TILE_LANES")
    p = oa.LayoutParams.defaults(g, device=0, flags=flags, iter_max=ITER_MAX, min_term_updates=TPS * g.n_steps)
    p.seed = seed
    etas = oa.path_linear_sgd_layout_schedule(p)
    curve = {}
    t0 = time.time()
    with oa.LayoutSession(g, p) as s:
        s.upload(X0, Y0)
        info = s.tile_info()
        info["drain_beside"] = s.drain_beside()[0]
        for it in range(p.iter_max):
            s.iteration(etas[it], it >= p.first_cooling_iteration(), p.min_term_updates)
            s.sync()
            if it + 1 >= CURVE_FROM and ((it + 1 - CURVE_FROM) % 2 == 0 or it + 1 == p.iter_max):
                X, Y = s.download_f64(flush=True)
                curve[it + 1] = oa.path_stress(g, X, Y, 2_000_000, seed=1)
        ms = s.kernel_time()[0] + sum(s.aux_time())
    cl = stress_classes.classes(g, X, Y) if not os.environ.get("GAP_NO_CLASSES") else {"total": 0.0, "pairs": 0, "by": {}, "tail": {}, "node_segments": {}}
    nr = oa.path_stress_near(g, X, Y, zmax=4, mod_step=224, mod_rank=256)   # no sampling error: what the layouts really differ by
    cl["near_exact"] = dict(near=nr["near"], by_z=nr["num"].sum(axis=(1, 2)).tolist(), by_z_flips=nr["num"].tolist(), zero_mass=nr["zero_mass"],
                            hist_step=nr["hist_step"].tolist(), hist_rank=nr["hist_rank"].tolist())
    # how much the SAMPLED evaluator moves with its own seed on one layout (GAP_EVAL_SEEDS=k: seeds 1..k, 2e6 pairs; and 2e7 pairs, seeds 1..3)
    ev = {}
    for es in range(1, int(os.environ.get("GAP_EVAL_SEEDS", "0")) + 1):
        ev[f"2e6@{es}"] = oa.path_stress(g, X, Y, 2_000_000, seed=es)
        if es <= 3:
            ev[f"2e7@{es}"] = oa.path_stress(g, X, Y, 20_000_000, seed=es)
    cl["sampled_evaluator_by_seed"] = ev
    rec = dict(exp="gap_classes", nodes=N, iter_max=ITER_MAX, terms_per_step=TPS, variant=v_in, tiled=bool(info["tiled"]), parts=info.get("parts"), drain_beside=info["drain_beside"],
               stress_curve=curve, stress_final=curve[p.iter_max], near_exact=cl["near_exact"]["near"], kernel_ms=ms, wall_s=time.time() - t0, classes=cl)
    print(json.dumps(rec), flush=True)
    results.append((v_in, cl))
    ms_of[v_in] = ms
for name, cl in results:
    ne = cl["near_exact"]
    hs, hr = np.array(ne["hist_step"]), np.array(ne["hist_rank"])
    if ne and cl.get("sampled_evaluator_by_seed"):
        print(f"{name:40s} sampled evaluator by seed: " + " ".join(f"{k}={v:.4f}" for k, v in cl["sampled_evaluator_by_seed"].items()), flush=True)
    print(f"{name:40s} kernel_ms {ms_of[name]:.0f} near_exact {ne['near']:.5f} by z {np.round(ne['by_z'], 5).tolist()}  step%224: first8 {hs[:8].sum() / hs.sum():.4f} last8 {hs[-8:].sum() / hs.sum():.4f} (uniform {8 / 224:.4f})"
          f"  rank%256: first8 {hr[:8].sum() / hr.sum():.4f} last8 {hr[-8:].sum() / hr.sum():.4f} (uniform {8 / 256:.4f})", flush=True)
for name, cl in results[1:]:
    if os.environ.get("GAP_NO_CLASSES"):
        break
    print(f"--- {results[0][0]} vs {name}", flush=True)
    print(stress_classes.diff_table(results[0][1], cl, (results[0][0][:12], name[:12])), flush=True)
