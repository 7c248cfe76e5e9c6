#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu.ids\|2D path-guided SGD: iteration" | tail -40 | tee gpurun_out/pytest_gpu.log
echo "== replicates"; timeout 900 python tools/gpu_replicates.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/replicates.jsonl
echo "== anchor speed"; timeout 600 python tools/gpu_ablate.py 2>&1 | grep -v amdgpu.ids | grep "anchor_speed\|anchor_full30" | tee gpurun_out/anchor.jsonl
