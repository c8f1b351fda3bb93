#!/bin/bash
# round 5, GPU call 17: weight runs in the single-pass f16 modes -- f16-mode tests + the configs[4]-shaped line
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5c17
mkdir -p $OUT
cd $R
( timeout 1200 python -m pytest tests/test_gpu_widened.py tests/test_gpu_parity.py -x -q -k "f16 or half_mode or under_autocast or config4" 2>&1 | tail -3 ) > $OUT/pytest.txt
for rep in 1 2 3; do
  timeout 400 python bench.py --voxel 0.05 --points 500000 --f16 --steps 30 --warmup 8 --no-cpu-baseline --no-cfg3 --no-cfg4 --no-cfg5 --no-pipelined 2>/dev/null | tail -1 | python tools/r5/line_brief.py
done > $OUT/f16.txt 2>&1
cat $OUT/pytest.txt $OUT/f16.txt
