#!/bin/bash
# round 5, GPU call 11: the phase trace with the issue phase split in three (index reads | gathers | weight DMA)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5c11
mkdir -p $OUT
cd $R
export ISF_BENCH_FRAME_CACHE=/tmp/isf_bench_frames
timeout 600 python tools/conv_phase_trace.py --level 3 > $OUT/phase_256.txt 2>&1
timeout 600 python tools/conv_phase_trace.py --level 2 > $OUT/phase_128.txt 2>&1
grep -v Warning $OUT/phase_256.txt | head -20; grep "three pieces" $OUT/phase_128.txt
