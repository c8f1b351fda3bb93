#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of the end-to-end point-cloud path, summarised to text.
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/eprof
timeout 900 rocprofv3 --kernel-trace -d /tmp/eprof -o e2e -- python tools/e2e_bench.py --steps 5 --warmup 2 > /tmp/eprof.log 2>&1
tail -1 /tmp/eprof.log | cut -c1-300
db=$(find /tmp/eprof -name "*.db" | head -1)
python tools/rocpd_summary.py "$db" | cut -c1-170 > gpurun_out/e2e_kernels.txt 2>&1
head -45 gpurun_out/e2e_kernels.txt
