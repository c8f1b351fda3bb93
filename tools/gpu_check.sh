#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, smoke, short bench.  Logs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider 2>&1 | grep -v "^E  *+\|Warn" | tail -60 | tee gpurun_out/test_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -5 | tee gpurun_out/smoke.log
timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -3 | tee gpurun_out/bench.log
timeout 600 python bench.py --steps 10 --warmup 3 --fp32 --no-cpu-baseline 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -3 | tee gpurun_out/bench_fp32.log
