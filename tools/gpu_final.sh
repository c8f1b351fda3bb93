#!/bin/bash
# Round-end validation on the GPU box: full -m gpu suite, smoke, judged bench line, rocprofv3 trace + PMC summary.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 2>&1 | grep -v "Warn" | tail -25 | cut -c1-180 | tee gpurun_out/test_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/smoke.log
bash tools/gpu_round_profile.sh > /dev/null 2>&1
head -14 gpurun_out/round_profile.txt | cut -c1-150
cut -c1-260 gpurun_out/bench_line.json
