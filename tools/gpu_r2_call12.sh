#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout 600 python tools/gpu_hotcap.py > $O/hotcap_per_lane.jsonl 2> $O/hotcap.err; cat $O/hotcap_per_lane.jsonl; tail -3 $O/hotcap.err
