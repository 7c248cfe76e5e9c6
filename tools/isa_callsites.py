#!/usr/bin/env python3
"""Static instruction counts of one kernel grouped by the outermost call site of the inline chain (the `@[ file:line ]`
comments of a `hipcc -g -S --cuda-device-only` listing).  Usage: isa_callsites.py dev.s kernel-substring file.hpp lo hi
groups every instruction by the line in [lo, hi] of file.hpp that appears in its inline chain (the term loop's stages)."""
import re, sys, collections
path, want, fname, lo, hi = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
def kind(op):
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_sleep"): return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "br"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "vmem"
    return "other"
counts = collections.defaultdict(collections.Counter)
sub = collections.defaultdict(lambda: collections.defaultdict(collections.Counter))
inside = False
cur = None
cur_leaf = None
pat = re.compile(re.escape(fname) + r":(\d+):\d+")
with open(path, errors="replace") as f:
    for l in f:
        if not inside:
            if l.startswith("_Z") and want in l and ":" in l: inside = True
            continue
        if l.startswith(".Lfunc_end"): break
        s = l.strip()
        if s.startswith(".loc"):
            c = s.split(";", 1)[1] if ";" in s else ""
            ls = [int(x) for x in pat.findall(c)]
            site = [x for x in ls if lo <= x <= hi]
            cur = site[-1] if site else None
            m = re.search(r"([\w\.]+):(\d+):\d+", c)
            cur_leaf = (m.group(1), int(m.group(2))) if m else None
            continue
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"): continue
        k = kind(s.split()[0])
        counts[cur][k] += 1
        sub[cur][cur_leaf][k] += 1
for site in sorted(counts, key=lambda x: (x is None, x)):
    c = counts[site]
    print("site %s: valu %d salu %d lds %d vmem %d br %d" % (site, c["valu"], c["salu"], c["lds"], c["vmem"], c["br"]))
    if "-v" in sys.argv:
        for leaf, cc in sorted(sub[site].items(), key=lambda x: -x[1]["valu"])[:40]:
            if cc["valu"] >= 3: print("      %-28s valu %3d salu %3d lds %2d vmem %2d" % ("%s:%d" % leaf if leaf else "?", cc["valu"], cc["salu"], cc["lds"], cc["vmem"]))
