#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "amdgpu.ids\|2D path-guided SGD: iteration" | tail -40 | tee gpurun_out/pytest_gpu.log
echo "== ablate"; timeout 600 python tools/gpu_ablate.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ablate2.jsonl

