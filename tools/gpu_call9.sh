#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu.ids\|2D path-guided SGD: iteration\|snapshot thread" | grep -E "5000 paths|passed|failed|Error|assert" | tail -10 | tee gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -1
echo "== bench"; timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; cut -c1-400 gpurun_out/bench.json
bash tools/profile_bench.sh > gpurun_out/profile.log 2>&1; tail -1 gpurun_out/profile.log
