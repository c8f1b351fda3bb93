#!/bin/bash
# Runs on the GPU box: the judged bench line + rocprofv3 kernel-trace stats + HBM traffic counters (separate passes),
# everything summarised to text under gpurun_out/ (the rocpd databases stay on the box).
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 > $R/gpurun_out/bench_line.json
CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline"
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof/trace -o bench -- $CMD > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof/fetch -o bench -- $CMD > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof/write -o bench -- $CMD > /dev/null 2>&1
cd $R
python tools/rocpd_summary.py /tmp/prof/trace/bench_results.db --pmc fetch=/tmp/prof/fetch/bench_results.db write=/tmp/prof/write/bench_results.db | cut -c1-260 > gpurun_out/round_profile.txt
head -50 gpurun_out/round_profile.txt
cut -c1-300 gpurun_out/bench_line.json
