#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc -save-temps .s file: per basic block, VALU / SALU / LDS / VMEM / other
counts, with backward branches marked (loops).  Usage: isa_blocks.py file.s [kernel-substring]"""
import re, sys
path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else None
lines = open(path).read().split("\n")
start = 0
if want:
    for i, l in enumerate(lines):
        if l.startswith("_Z") and want in l and l.rstrip().endswith(("E", "E:")) or (l.startswith("_Z") and want in l and ":" in l):
            start = i
            break
blocks = []
cur = ["entry", []]
labels = {}
for i in range(start + 1, len(lines)):
    l = lines[i]
    if l.startswith(".Lfunc_end"):
        break
    m = re.match(r"^(\.LBB[0-9_]+):", l)
    if m:
        blocks.append(cur)
        cur = [m.group(1), []]
        continue
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        continue
    cur[1].append(s.split(";")[0].strip())
blocks.append(cur)
order = {b[0]: k for k, b in enumerate(blocks)}
def kind(op):
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_sleep"): return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "br"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "vmem"
    return "other"
tot = {}
for k, (name, ins) in enumerate(blocks):
    c = {}
    back = []
    for s in ins:
        op = s.split()[0]
        kd = kind(op)
        c[kd] = c.get(kd, 0) + 1
        tot[kd] = tot.get(kd, 0) + 1
        if kd == "br":
            tgt = s.split()[-1]
            if tgt in order and order[tgt] <= k:
                back.append(tgt)
    print(f"{k:4d} {name:12s} n={len(ins):4d} valu={c.get('valu',0):4d} salu={c.get('salu',0):4d} lds={c.get('lds',0):3d} vmem={c.get('vmem',0):3d} wait={c.get('wait',0):3d} br={c.get('br',0):2d}" + (f"  BACK->{','.join(back)}" if back else ""))
print("total", tot)
