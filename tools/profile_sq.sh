#!/bin/bash
# rocprofv3 PMC passes of the default bench command: where the wave cycles of the update kernel go (SQ) and how the
# L2 behaves (TCC).  Counters only, no tracing.  Usage: tools/profile_sq.sh <tag>; outputs gpurun_out/prof_sq_<tag>/.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${1:-run}
OUT=$PWD/gpurun_out/prof_sq_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python $PWD/bench.py --cpu-seconds 0 --steps 20 --warmup 5"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d "$OUT/sq1" -o bench -- $CMD > "$OUT/sq1.json" 2> "$OUT/sq1.err")
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU -d "$OUT/sq2" -o bench -- $CMD > "$OUT/sq2.json" 2> "$OUT/sq2.err")
if [ -z "$SKIP_TCC" ]; then
(cd /tmp && timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/tcc" -o bench -- $CMD > "$OUT/tcc.json" 2> "$OUT/tcc.err")
fi
python3 - "$OUT" <<'PY'
import sqlite3, sys, os, json, glob
out = sys.argv[1]
res = {}
for d in ("sq1", "sq2", "tcc"):
    dbs = glob.glob(os.path.join(out, d, "**", "*.db"), recursive=True)
    if not dbs:
        print("missing db for", d); continue
    con = sqlite3.connect(dbs[0])
    try:
        rows = con.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                           "group by kernel_name, counter_name").fetchall()
    except Exception as e:
        print(d, "query failed", e); rows = []
    for k, c, n, v, dur in rows:
        short = "tile" if "sgd_tile_kernel" in k else "snapshot" if "snapshot_kernel" in k else "drain" if "far_drain" in k else None
        if short:
            res.setdefault(short, {})[c] = {"dispatches": n, "mean": v, "mean_duration_ns": dur}
json.dump(res, open(os.path.join(out, "sq_tcc_summary.json"), "w"), indent=1)
t = res.get("tile", {})
def g(k): return t.get(k, {}).get("mean", float("nan"))
print(json.dumps({k: round(g(k)) for k in t}, indent=0))
if t:
    wc = g("SQ_WAVE_CYCLES")
    print("tile kernel: wait %.3f issue-stall %.3f active %.3f | VALU-active %.3f LDS-active %.3f | VALU wave-instr %.3g LDS %.3g VMEM rd %.3g wr %.3g SALU %.3g | LDS conflict/active %.3f"
          % (g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc, g("SQ_ACTIVE_INST_ANY") / wc, g("SQ_ACTIVE_INST_VALU") / wc, g("SQ_ACTIVE_INST_LDS") / wc,
             g("SQ_INSTS_VALU"), g("SQ_INSTS_LDS"), g("SQ_INSTS_VMEM_RD"), g("SQ_INSTS_VMEM_WR"), g("SQ_INSTS_SALU"),
             g("SQ_LDS_BANK_CONFLICT") / max(1.0, g("SQ_LDS_IDX_ACTIVE"))))
PY
for f in "$OUT"/*.err; do tail -n 1 "$f"; done
find "$OUT" -type f -size +4M -delete
