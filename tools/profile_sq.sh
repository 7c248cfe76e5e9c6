#!/bin/bash
# Extra rocprofv3 PMC passes of the default bench command: where the wave cycles of the update kernel go (SQ)
# and how the L2 behaves (TCC).  Counters only, no tracing.  Outputs under gpurun_out/prof_sq/.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_sq
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python $PWD/bench.py --cpu-seconds 0"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES -d "$OUT/sq1" -o bench -- $CMD > "$OUT/sq1.json" 2> "$OUT/sq1.err")
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU -d "$OUT/sq2" -o bench -- $CMD > "$OUT/sq2.json" 2> "$OUT/sq2.err")
(cd /tmp && timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/tcc" -o bench -- $CMD > "$OUT/tcc.json" 2> "$OUT/tcc.err")
(cd /tmp && timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d "$OUT/grbm" -o bench -- $CMD > "$OUT/grbm.json" 2> "$OUT/grbm.err")
python3 - "$OUT" <<'PY'
import sqlite3, sys, os, json
out = sys.argv[1]
res = {}
for d in ("sq1", "sq2", "tcc", "grbm"):
    db = os.path.join(out, d, "bench_results.db")
    if not os.path.exists(db):
        print("missing", db); continue
    con = sqlite3.connect(db)
    try:
        rows = con.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                           "where kernel_name like '%sgd_tile_kernel%' group by kernel_name, counter_name").fetchall()
    except Exception as e:
        print(d, "query failed", e); rows = []
    for k, c, n, v, dur in rows:
        res[c] = {"dispatches": n, "mean": v, "mean_duration_ns": dur}
json.dump(res, open(os.path.join(out, "sq_tcc_summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
for f in "$OUT"/*.err; do tail -n 1 "$f"; done
find "$OUT" -type f -size +8M -delete
