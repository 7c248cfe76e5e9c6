#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
: > $O/curves_far_policy.jsonl
python tools/gpu_curves.py "snapshot per iteration, relax 1" >> $O/curves_far_policy.jsonl 2>/dev/null
PGSGD_FAR_RELAX=0.5 python tools/gpu_curves.py "snapshot per iteration, relax 0.5" >> $O/curves_far_policy.jsonl 2>/dev/null
PGSGD_SNAPSHOT_PER_LAUNCH=1 python tools/gpu_curves.py "snapshot per launch, relax 1" >> $O/curves_far_policy.jsonl 2>/dev/null
PGSGD_SNAPSHOT_PER_LAUNCH=1 PGSGD_FAR_RELAX=0.5 python tools/gpu_curves.py "snapshot per launch, relax 0.5" >> $O/curves_far_policy.jsonl 2>/dev/null
PGSGD_SNAPSHOT_PER_LAUNCH=1 PGSGD_FAR_RELAX=0.25 python tools/gpu_curves.py "snapshot per launch, relax 0.25" >> $O/curves_far_policy.jsonl 2>/dev/null
python tools/gpu_curves.py "per-lane kernel" --no-tiles >> $O/curves_far_policy.jsonl 2>/dev/null
cat $O/curves_far_policy.jsonl | cut -c1-420
