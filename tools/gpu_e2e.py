#!/usr/bin/env python3
"""pgsgd_layout_run end to end at BASELINE config 4 (host buffers in, host buffers out): wall time by phase
(PGSGD_TIMING=1 prints them on stderr) and the PCIe-inclusive rate."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PGSGD_TIMING"] = "1"
import numpy as np
import odgi_amd as oa
g = oa.Graph.synthetic(1_000_000, 50, seed=42)
p = oa.LayoutParams.defaults(g, device=0)
X0, Y0 = oa.initial_layout(g, "d", seed=42)
for rep in range(2):
    X, Y = X0.astype(np.float32), Y0.astype(np.float32)
    t = time.perf_counter()
    st = oa.path_linear_sgd_layout_gpu(g, p, X, Y)
    wall = time.perf_counter() - t
    print(json.dumps({"rep": rep, "wall_s": wall, "kernel_ms": st["kernel_ms"], "terms": st["term_updates"], "terms_per_s_end_to_end": st["term_updates"] / wall,
                      "stress": oa.path_stress(g, X, Y, 1_000_000)}), flush=True)
