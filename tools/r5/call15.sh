#!/bin/bash
# round 5, GPU call 15: narrow-layer kernel, weight run issued before the transit wait (production) vs after (side lib)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5c15
mkdir -p $OUT
cd $R
export ISF_BENCH_FRAME_CACHE=/tmp/isf_bench_frames
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "dma or tile_mix or one_column_block" 2>&1 | tail -3 ) > $OUT/pytest.txt
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-cfg3 --no-cfg4 --no-cfg5 --no-pipelined"
for rep in 1 2 3; do
  echo "late-B side lib rep $rep: $(timeout 300 $B --lib tools/probes/_build/libisf_hip_dma_late_b.so 2>/dev/null | tail -1 | python tools/r5/line_brief.py)"
  echo "production rep $rep: $(timeout 300 $B 2>/dev/null | tail -1 | python tools/r5/line_brief.py)"
done > $OUT/ab.txt 2>&1
cat $OUT/pytest.txt $OUT/ab.txt
