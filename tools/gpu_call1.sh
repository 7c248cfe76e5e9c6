#!/bin/bash
# first GPU call: parity tests, smoke, design sweeps, bench, rocprof
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -5 > gpurun_out/gpu.txt; nproc >> gpurun_out/gpu.txt; lscpu | grep "Model name" >> gpurun_out/gpu.txt
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -60 > gpurun_out/pytest_gpu.log; tail -30 gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -5 gpurun_out/smoke.log
echo "== bench"; timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
echo "== sweep synth"; timeout 900 python tools/gpu_sweep.py synth > gpurun_out/sweep_synth.log 2>&1; tail -25 gpurun_out/sweep_synth.log
echo "== sweep streams"; timeout 900 python tools/gpu_sweep.py streams > gpurun_out/sweep_streams.log 2>&1; tail -60 gpurun_out/sweep_streams.log
