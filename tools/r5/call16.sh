#!/bin/bash
# round 5, GPU call 14: the A2 loop (gathered rows two steps ahead) -- bits, then A/B against the production loop
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5c16
mkdir -p $OUT
cd $R
export ISF_BENCH_FRAME_CACHE=/tmp/isf_bench_frames
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "one_column_block" 2>&1 | tail -3 ) > $OUT/pytest.txt
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-cfg3 --no-cfg4 --no-cfg5 --no-pipelined"
for rep in 1 2 3; do
  for d in 8388608 0; do
    echo "diag $d rep $rep: $(timeout 300 $B --conv-diag $d 2>/dev/null | tail -1 | python tools/r5/line_brief.py)"
  done
done > $OUT/ab.txt 2>&1
cat $OUT/pytest.txt $OUT/ab.txt
