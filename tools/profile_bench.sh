#!/bin/bash
# rocprofv3 evidence for bench.py's roofline: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in
# separate PMC passes (MI355X_MICROARCH.md: TCC slots cannot hold both).  Outputs under gpurun_out/prof/.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python $PWD/bench.py --cpu-seconds 0 --steps 20 --warmup 5"   # the driver's bench command minus the CPU baseline leg
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o bench -- $CMD > "$OUT/kt.json" 2> "$OUT/kt.err")
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o bench -- $CMD > "$OUT/fetch.json" 2> "$OUT/fetch.err")
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o bench -- $CMD > "$OUT/write.json" 2> "$OUT/write.err")
find "$OUT" -type f | head -50
# keep only the small summaries (kernel trace CSV of 30 launches is small; drop anything huge)
find "$OUT" -type f -size +8M -delete
for f in $(find "$OUT" -name "*stats*.csv" | head -5); do echo "== $f"; head -12 "$f"; done
for f in $(find "$OUT" -name "*counter_collection*.csv" | head -2); do echo "== $f"; head -5 "$f"; done
for f in "$OUT"/*.err; do tail -n 2 "$f"; done
