#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02; mkdir -p $O
timeout 600 python -m pytest tests -x -q -m gpu -k "one_workgroup or terms_bit_exact or refuses or unsorted or tandem" > $O/pytest_13.log 2>&1; echo "pytest rc $?" >> $O/pytest_13.log; tail -4 $O/pytest_13.log
timeout 300 python tools/gpu_e2e.py > $O/e2e_config4.jsonl 2> $O/e2e_config4_phases.txt; cat $O/e2e_config4.jsonl; grep "pgsgd timing" $O/e2e_config4_phases.txt | tail -14
timeout 600 python tools/gpu_hotcap.py > $O/hotcap_per_lane_v2.jsonl 2> $O/hotcap.err; cut -c1-260 $O/hotcap_per_lane_v2.jsonl
