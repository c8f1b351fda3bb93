#!/bin/bash
# per-workgroup trace + cycle account of the narrow layers' kernel (tools/conv_trace.py --level 0 | 1)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06_dma_trace}; mkdir -p $OUT
export ISF_BENCH_FRAME_CACHE=/tmp/isf_frames
for lvl in 0 1; do
  timeout 600 python tools/conv_trace.py --level $lvl 2>&1 | tail -20
done | tee $OUT/dma_trace.txt
