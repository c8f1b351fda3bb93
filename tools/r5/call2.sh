#!/bin/bash
# round 5, GPU call 2: phase trace of the 256 -> 256 (level 3) and 128 -> 128 (level 2) launches + the round's baseline line
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5c2
mkdir -p $OUT
cd $R
export ISF_BENCH_FRAME_CACHE=/tmp/isf_bench_frames
( timeout 60 rocprofv3 --att --kernel-trace -d /tmp/att_try -- /bin/true 2>&1 | tail -3 ) > $OUT/att_attempt.txt
timeout 600 python tools/conv_phase_trace.py --level 3 --dump $OUT/phase_l3.npz > $OUT/phase_256.txt 2>&1
timeout 600 python tools/conv_phase_trace.py --level 2 > $OUT/phase_128.txt 2>&1
timeout 900 python bench.py 2>/dev/null | tail -1 > $OUT/bench_base.json
cat $OUT/att_attempt.txt; cat $OUT/phase_256.txt; cat $OUT/phase_128.txt | tail -30; cut -c1-600 $OUT/bench_base.json
