#!/usr/bin/env python3
"""Static VALU / SALU / LDS / VMEM instruction counts per source line of one kernel, from a `hipcc -g -S
--cuda-device-only` listing (.loc directives).  Usage: isa_lines.py dev.s kernel-substring [file-substring]"""
import re, sys, collections
path, want = sys.argv[1], sys.argv[2]
files = {}
counts = collections.defaultdict(lambda: collections.Counter())
inside = False
cur = None
def kind(op):
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_sleep"): return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "br"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "vmem"
    return "other"
with open(path, errors="replace") as f:
    for l in f:
        if l.startswith("\t.file"):
            m = re.match(r'\t\.file\t(\d+) "([^"]*)"(?: "([^"]*)")?', l)
            if m: files[int(m.group(1))] = (m.group(3) or m.group(2))
            continue
        if not inside:
            if l.startswith("_Z") and want in l and ":" in l:
                inside = True
            continue
        if l.startswith(".Lfunc_end"):
            break
        s = l.strip()
        if s.startswith(".loc"):
            p = s.split()
            cur = (int(p[1]), int(p[2]))
            continue
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        op = s.split()[0]
        counts[cur][kind(op)] += 1
tot = collections.Counter()
rows = []
for k, c in counts.items():
    tot.update(c)
    rows.append((k, c))
rows.sort(key=lambda r: (files.get(r[0][0], "?") if r[0] else "?", r[0][1] if r[0] else 0))
for k, c in rows:
    fn = files.get(k[0], "?").split("/")[-1] if k else "?"
    if len(sys.argv) > 3 and sys.argv[3] not in fn: continue
    print("%-22s %5d  valu %4d salu %4d lds %3d vmem %3d" % (fn, k[1] if k else 0, c["valu"], c["salu"], c["lds"], c["vmem"]))
print("total", dict(tot))
