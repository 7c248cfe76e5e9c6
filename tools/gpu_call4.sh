#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v "amdgpu.ids\|2D path-guided SGD: iteration" | tail -8 | tee gpurun_out/pytest_gpu.log
echo "== bench"; timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err; cat gpurun_out/bench.json; tail -2 gpurun_out/bench.err
echo "== ranks"; timeout 900 python tools/gpu_sweep.py ranks 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ranks.jsonl
