#!/bin/bash
# cfg3 kernel trace of one forward in launch order (tools/step_kernels.py) -> gpurun_out/<tag>/cfg3_step.txt
TAG=${1:-r06_cfg3_trace}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ISF_BENCH_FRAME_CACHE=/tmp/isf_frames MIOPEN_FIND_MODE=FAST
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof/trace3 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline --config 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/step_kernels.py /tmp/prof/trace3/bench_results.db vfe_prep_kernel 8 | cut -c1-160 > $OUT/cfg3_step.txt
python tools/timeline_gaps.py /tmp/prof/trace3/bench_results.db vfe_prep_kernel 4 4 | cut -c1-200 > $OUT/timeline_gaps_cfg3.txt
head -3 $OUT/cfg3_step.txt; head -4 $OUT/timeline_gaps_cfg3.txt
