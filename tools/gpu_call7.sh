#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu.ids\|2D path-guided SGD: iteration\|snapshot thread" | tail -30 | tee gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2
echo "== bench"; timeout 600 python bench.py --steps 10 --warmup 2 --stress > gpurun_out/bench.json 2> gpurun_out/bench.err; cat gpurun_out/bench.json; tail -2 gpurun_out/bench.err
echo "== bench per-lane"; timeout 600 python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --no-tiles > gpurun_out/bench_per_lane.json 2>> gpurun_out/bench.err; cat gpurun_out/bench_per_lane.json
bash tools/profile_bench.sh > gpurun_out/profile.log 2>&1; tail -3 gpurun_out/profile.log
