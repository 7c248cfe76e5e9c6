#!/usr/bin/env python3
"""End to end `odgi layout -i <GFA>` at BASELINE config 5's size: writes the synthetic 1e7-node / 50-path pangenome as GFA v1, then
times the product's CLI on it — parse, lower (handles only: positions are built on the device), upload, the whole default
schedule, .lay — phase by phase (PGSGD_DEBUG=1 PGSGD_TIMING=1 prints the library's laps).   gpu_e2e_gfa.py [NODES] [PATHS]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import odgi_amd as oa
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 50
tmp = os.environ.get("TMPDIR", "/tmp")
gfa = os.path.join(tmp, f"synthetic_{N}_{P}.gfa")
t0 = time.time()
g = oa.Graph.synthetic(N, P, seed=42)
t_synth = time.time() - t0
t0 = time.time()
with open(gfa, "wb") as f:   # S lines with sequences of the nodes' lengths, L lines of the backbone, P lines
    f.write(b"H\tVN:Z:1.0\n")
    ln = g.node_len
    seq = b"A" * int(ln.max())
    chunk = []
    for i in range(N):
        chunk.append(b"S\t%d\t%s\n" % (i + 1, seq[:ln[i]]))
        if len(chunk) >= 1 << 16:
            f.write(b"".join(chunk)); chunk = []
    f.write(b"".join(chunk))
    e = g.edges
    for lo in range(0, len(e), 1 << 16):
        part = e[lo:lo + (1 << 16)]
        f.write(b"".join(b"L\t%d\t%s\t%d\t%s\t0M\n" % ((a >> 1) + 1, b"-" if a & 1 else b"+", (b >> 1) + 1, b"-" if b & 1 else b"+") for a, b in part.tolist()))
    first, h = g.path_first, g.step_handle
    for p in range(P):
        hp = h[int(first[p]):int(first[p + 1])]
        ids = (hp >> 1) + 1
        toks = np.char.add(ids.astype("U"), np.where(hp & 1, "-", "+"))
        f.write(b"P\thap%d\t" % p + ",".join(toks.tolist()).encode() + b"\t*\n")
t_write = time.time() - t0
size = os.path.getsize(gfa)
del g
lay = os.path.join(tmp, "synthetic.lay")
env = dict(os.environ, PGSGD_DEBUG="1", PGSGD_TIMING="1")
odgi = os.path.join(ROOT, "odgi_amd", "lib", "odgi")
runs = []
for rep in range(2):
    t0 = time.time()
    r = subprocess.run([odgi, "layout", "-i", gfa, "-o", lay, "-t", "64", "-P", "--seed", "42"], env=env, capture_output=True, text=True)
    wall = time.time() - t0
    laps = [l for l in r.stderr.splitlines() if "pgsgd timing" in l or "timing" in l.lower()]
    runs.append(dict(rc=r.returncode, wall_s=wall, lay_bytes=os.path.getsize(lay) if os.path.exists(lay) else 0, laps=laps, stderr_tail=r.stderr.splitlines()[-12:]))
# the same in one process, with and without the host-built step index (what the CLI did until round 6): load + whole layout run
ab = {}
if os.environ.get("E2E_AB", "1") == "1":
    for name, loader in (("full index (12 more bytes per step built on the host and uploaded)", oa.Graph.load), ("handles only (positions built on the device)", oa.Graph.load_lean)):
        t0 = time.time()
        gg = loader(gfa, threads=64)
        t_load = time.time() - t0
        X0, Y0 = oa.initial_layout(gg, "d", seed=42)
        p = oa.LayoutParams.defaults(gg, device=0)
        X, Y = X0.copy(), Y0.copy()
        t0 = time.time()
        st = oa.path_linear_sgd_layout_gpu(gg, p, X, Y)
        t_run = time.time() - t0
        X, Y = X0.copy(), Y0.copy()
        t0 = time.time()
        st = oa.path_linear_sgd_layout_gpu(gg, p, X, Y)
        t_run2 = time.time() - t0
        ab[name] = dict(load_s=t_load, layout_run_s_first=t_run, layout_run_s_second=t_run2, kernel_ms=st["kernel_ms"], terms=st["term_updates"],
                        near_exact=oa.path_stress_near(gg, X.astype(np.float64), Y.astype(np.float64), zmax=4)["near"])
        del gg
print(json.dumps(dict(exp="e2e_gfa", nodes=N, paths=P, gfa_bytes=size, synth_s=t_synth, gfa_write_s=t_write, runs=runs, in_process=ab), indent=1))
os.remove(gfa)
