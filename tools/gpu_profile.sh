#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command, then PMC passes (HBM traffic) in separate runs.
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline"
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace -o bench -- $CMD 2>&1 | grep -v Warn | tail -3
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof/pmc_fetch -o bench -- $CMD 2>&1 | grep -v Warn | tail -2
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof/pmc_write -o bench -- $CMD 2>&1 | grep -v Warn | tail -2
cd $R
find gpurun_out/prof -type f | head -30
du -sh gpurun_out/prof
