#!/bin/bash
# round 4: parity of the one-workgroup-per-CU conv kernel, then an interleaved same-box A/B of the headline bench
# (--conv-diag 512 = 256-column layers on isf_sparse_conv_forward_cu, + 1024 v = its variant v).  Usage (GPU box):
#   bash tools/gpu_r4_ab.sh <tag> [pytest -k expression]
set -u
TAG=${1:-r04_ab}
KEXPR=${2:-cu_unit}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export ISF_BENCH_FRAME_CACHE=/tmp/isf_frames
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$KEXPR" 2>&1 | tail -15 ) > $OUT/pytest.txt
cat $OUT/pytest.txt | tail -5
VARIANTS=${3:-"0 512"}
REPS=${4:-"1 2 3"}
for rep in $REPS; do
  for v in $VARIANTS; do
    timeout 300 python bench.py --steps 40 --warmup 8 --no-cfg3 --no-cfg4 --no-cfg5 --no-pipelined --no-cpu-baseline --conv-diag $v > $OUT/bench_diag${v}_$rep.json 2> $OUT/bench_diag${v}_$rep.err
    python tools/ab_summary.py $OUT/bench_diag${v}_$rep.json
  done
done 2>&1 | tee $OUT/ab_summary.txt
