#!/bin/bash
# round 5, GPU call 12: index prefetch + one-M0 weight runs in the deep kernel: bits, A/B against round 4's issue phase
# (diag 2097152), phase trace
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5c12
mkdir -p $OUT
cd $R
export ISF_BENCH_FRAME_CACHE=/tmp/isf_bench_frames
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "one_column_block or tile_mix or tile_order or deterministic or batch" 2>&1 | tail -3 ) > $OUT/pytest.txt
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-cfg3 --no-cfg4 --no-cfg5 --no-pipelined"
for rep in 1 2 3; do
  for d in 4194304 0; do
    echo "diag $d rep $rep: $(timeout 300 $B --conv-diag $d 2>/dev/null | tail -1 | python tools/r5/line_brief.py)"
  done
done > $OUT/ab.txt 2>&1

cat $OUT/pytest.txt $OUT/ab.txt
