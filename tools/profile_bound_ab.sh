#!/bin/bash
# What bounds the tile kernel — memory lines or latency?  The same rocprofv3 --pmc passes (counters only, one small group
# per pass) of the bench command for three partner rules: the shipped one (partner pairs: two lanes share a 64-byte
# half line), partner quads (the shipped rule since round 6; PGSGD_TILE_PAIRS=1: the pairs of rounds 4-6) — four lanes share a 128-byte line; halves the uniform partners' lines
# again) and no sharing (PGSGD_FLAG_NO_PARTNER_PAIRS).  If the read requests fall with the sharing and the time does
# not, the kernel is not bound by lines.  Usage: tools/profile_bound_ab.sh <tag> -> gpurun_out/bound_ab_<tag>/bound_ab.json
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG=${1:-run}
OUT=$PWD/gpurun_out/bound_ab_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
BENCH="python $PWD/bench.py --cpu-seconds 0 --steps 20 --warmup 5"
variant() {  # name, env assignments..., then "--", then extra bench flags
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  (cd /tmp && env "${envs[@]}" timeout 300 $BENCH "$@" > "$OUT/${name}_plain.json" 2> "$OUT/${name}_plain.err")
  (cd /tmp && env "${envs[@]}" timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d "$OUT/${name}_req" -o bench -- $BENCH "$@" > "$OUT/${name}_req.json" 2> "$OUT/${name}_req.err")
  (cd /tmp && env "${envs[@]}" timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/${name}_hit" -o bench -- $BENCH "$@" > "$OUT/${name}_hit.json" 2> "$OUT/${name}_hit.err")
  (cd /tmp && env "${envs[@]}" timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD -d "$OUT/${name}_sq" -o bench -- $BENCH "$@" > "$OUT/${name}_sq.json" 2> "$OUT/${name}_sq.err")
}
variant pairs PGSGD_AB=pairs --
variant pairs PGSGD_DEBUG=1 PGSGD_TILE_PAIRS=1 --   # (quads are what sessions run since round 6; rounds 4-5 ran this A/B the other way round)
variant nopairs PGSGD_AB=nopairs -- --flags 0x4000
python3 - "$OUT" <<'PY'
import sqlite3, sys, os, json, glob, re
out = sys.argv[1]
res = {}
for variant in ("pairs", "quads", "nopairs"):
    v = res.setdefault(variant, {})
    try:
        d = json.loads(open(os.path.join(out, variant + "_plain.json")).read().strip().splitlines()[-1])
        r = d["roofline"]
        v["unprofiled"] = {"value": d["value"], "frac": r["frac"], "frac_all_kernels": r["frac_all_kernels"],
                           "warm_frac": r["phases"].get("warm", {}).get("frac"), "cooling_frac": r["phases"].get("cooling", {}).get("frac"),
                           "warm_ms_per_step": r["phases"].get("warm", {}).get("update_kernel_ms_per_step"),
                           "cooling_ms_per_step": r["phases"].get("cooling", {}).get("update_kernel_ms_per_step"), "terms_per_launch": r["terms_per_launch"]}
    except Exception as e:
        v["unprofiled"] = {"error": str(e)}
    for p in ("req", "hit", "sq"):
        for db in glob.glob(os.path.join(out, f"{variant}_{p}", "**", "*.db"), recursive=True):
            con = sqlite3.connect(db)
            try:
                rows = con.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name").fetchall()
            except Exception as e:
                print(variant, p, "query failed", e); rows = []
            for k, c, n, val, dur in rows:
                m = re.search(r"sgd_tile_kernel<\s*\d+,\s*\d+,\s*(\w+),\s*(\w+)", k)
                if m:
                    phase = ("cooling" if m.group(1) in ("true", "1") else "warm")
                elif "far_drain_kernel" in k:
                    phase = "drain"
                else:
                    continue
                v.setdefault(phase, {})[c] = {"dispatches": n, "mean": val, "mean_duration_ns": dur}
# derived
for variant, v in res.items():
    for phase in ("warm", "cooling"):
        c = v.get(phase, {})
        g = lambda k: c.get(k, {}).get("mean")
        d = {}
        if g("TCC_EA0_RDREQ_sum") is not None:
            t = v.get("unprofiled", {}).get("terms_per_launch") or 0
            d["read_requests_per_term"] = g("TCC_EA0_RDREQ_sum") / t if t else None
            d["write_requests_per_term"] = g("TCC_EA0_WRREQ_sum") / t if t else None
        if g("TCC_HIT_sum") is not None:
            d["l2_hit_rate"] = g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum"))
        if g("SQ_WAVE_CYCLES"):
            wc = g("SQ_WAVE_CYCLES")
            d["wait_any_over_wave_cycles"] = g("SQ_WAIT_ANY") / wc if g("SQ_WAIT_ANY") is not None else None
            d["wait_inst_any_over_wave_cycles"] = g("SQ_WAIT_INST_ANY") / wc if g("SQ_WAIT_INST_ANY") is not None else None
            d["valu_active_over_wave_cycles"] = g("SQ_ACTIVE_INST_VALU") / wc if g("SQ_ACTIVE_INST_VALU") is not None else None
        c["derived"] = d
json.dump(res, open(os.path.join(out, "bound_ab.json"), "w"), indent=1)
for variant, v in res.items():
    print(variant, "unprofiled", v.get("unprofiled"))
    for phase in ("warm", "cooling"):
        c = v.get(phase, {})
        print("   ", phase, {k: (round(x["mean"]), round(x["mean_duration_ns"] / 1e3)) for k, x in c.items() if k != "derived"}, c.get("derived"))
PY
for f in "$OUT"/*.err; do echo "$(basename $f): $(tail -n 1 $f | cut -c1-160)"; done
find "$OUT" -type f -size +4M -delete
