#!/bin/bash
# kernel trace of one headline step in launch order (tools/step_kernels.py) -> gpurun_out/<tag>/headline_step.txt
TAG=${1:-r06_headline_trace}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ISF_BENCH_FRAME_CACHE=/tmp/isf_frames
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof/traceh -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-cfg3 --no-cfg4 --no-cfg5 --no-pipelined > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/step_kernels.py /tmp/prof/traceh/bench_results.db vfe_prep_kernel 8 | cut -c1-160 > $OUT/headline_step.txt
python tools/timeline_gaps.py /tmp/prof/traceh/bench_results.db vfe_prep_kernel 4 4 | cut -c1-200 > $OUT/timeline_gaps.txt
head -3 $OUT/headline_step.txt; head -4 $OUT/timeline_gaps.txt
