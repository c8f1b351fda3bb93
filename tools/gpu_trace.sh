#!/bin/bash
# kernel trace only; summary text copied back (databases stay on the box)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof/trace -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -v "Warn\|W2026" | tail -1 | cut -c1-200
cd $R && python tools/rocpd_summary.py /tmp/prof/trace/bench_results.db | cut -c1-200 > gpurun_out/trace_summary.txt
head -45 gpurun_out/trace_summary.txt
