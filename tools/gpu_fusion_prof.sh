#!/bin/bash
# Runs on the GPU box: stage timing of the fusion encoder + rocprofv3 kernel trace summary.
mkdir -p gpurun_out
export TMPDIR=/tmp
B=${FUSION_BATCH:-4}
timeout 600 python tools/fusion_bench.py --batch $B 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -2 > gpurun_out/fusion_bench.log
cat gpurun_out/fusion_bench.log
rm -rf /tmp/fprof
timeout 900 rocprofv3 --kernel-trace -d /tmp/fprof -o fus -- python tools/fusion_bench.py --batch $B --steps 3 > /tmp/fprof.log 2>&1
db=$(find /tmp/fprof -name "*.db" | head -1)
python tools/rocpd_summary.py "$db" > gpurun_out/fusion_kernels.txt 2>&1
head -60 gpurun_out/fusion_kernels.txt
