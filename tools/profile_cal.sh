#!/bin/bash
# Calibration of rocprofv3's memory-side counters on kernels of known traffic (tools/microbench.hip, MICROBENCH_CAL=1) and
# the same counters on the bench command's tile kernel.  Counters only, one small group per pass.
# Usage: tools/profile_cal.sh <tag>; outputs gpurun_out/prof_cal_<tag>/summary.json
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${1:-run}
OUT=$PWD/gpurun_out/prof_cal_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
(cd /tmp && timeout 120 rocprofv3 -L > "$OUT/counters_avail.txt" 2>&1)
grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_HIT[_a-z]*\|TCC_MISS[_a-z]*\|TCC_REQ[_a-z]*\|FETCH_SIZE\|WRITE_SIZE" "$OUT/counters_avail.txt" | sort -u | tr '\n' ' ' > "$OUT/counters_tcc.txt"
MB="$PWD/odgi_amd/lib/microbench"
BENCH="python $PWD/bench.py --cpu-seconds 0 --steps 20 --warmup 5"
pass() {  # name, counters...
  local name=$1; shift
  (cd /tmp && MICROBENCH_CAL=1 timeout 200 rocprofv3 --pmc "$@" -d "$OUT/mb_$name" -o mb -- $MB > "$OUT/mb_$name.txt" 2> "$OUT/mb_$name.err")
  if [ -z "$SKIP_BENCH" ]; then
  (cd /tmp && timeout 300 rocprofv3 --pmc "$@" -d "$OUT/bench_$name" -o bench -- $BENCH > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err")
  fi
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass wrreq TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pass hit TCC_HIT_sum TCC_MISS_sum
pass rdsize TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass rddram TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum
python3 - "$OUT" <<'PY'
import sqlite3, sys, os, json, glob
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(os.path.join(out, "*_*"))):
    if not os.path.isdir(d): continue
    tag = os.path.basename(d)
    for db in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        con = sqlite3.connect(db)
        try:
            rows = con.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name").fetchall()
        except Exception as e:
            print(tag, "query failed", e); rows = []
        for k, c, n, v, dur in rows:
            short = None
            for key in ("cal_stream_read", "cal_stream_write", "cal_gather16", "cal_gather32", "cal_linewrite64", "cal_gather_halves", "sgd_tile_kernel", "far_drain_kernel", "snapshot_kernel"):
                if key in k: short = key
            if short and "policy" in k: short = "cal_gather16_policy" + k[k.index("policy") + 6:][:8]
            if short == "sgd_tile_kernel":   # sgd_tile_kernel<COORD_LOAD, FAR, COOLING, LOCAL, MATH, ABL>
                import re
                m = re.search(r"sgd_tile_kernel<\s*\d+,\s*\d+,\s*(\w+),\s*(\w+)", k)
                if m: short += ("_cooling" if m.group(1) in ("true", "1") else "_warm") + ("" if m.group(2) in ("true", "1") else "_windowless")
            if short:
                res.setdefault(("mb:" if tag.startswith("mb_") else "bench:") + short, {})[c] = {"dispatches": n, "mean": v, "mean_duration_ns": dur}
known = {}
for f in glob.glob(os.path.join(out, "mb_*.txt")):
    for l in open(f):
        if l.startswith('{"cal"'):
            d = json.loads(l); known[d["cal"]] = d
json.dump({"known": known, "counters": res}, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k in sorted(res):
    print(k, {c: (round(v["mean"]), v["dispatches"]) for c, v in res[k].items()})
for k, d in known.items():
    print("known", k, d)
PY
for f in "$OUT"/*.err; do echo "$f: $(tail -n 1 $f | cut -c1-150)"; done
find "$OUT" -type f -size +4M -delete
