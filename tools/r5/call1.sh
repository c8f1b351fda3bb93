#!/bin/bash
# round 5, GPU call 1: where the training step's time goes NOW (per-kernel, two traces differenced), the dW kernel's
# rate on four geometries (shipped vs the compacted-rows probe), and an untraced step at the BASELINE configs[3] size.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5c1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MIOPEN_FIND_MODE=FAST
T="python $R/tools/train_step.py --autocast"
( timeout 300 $T --steps 3 --points 60000 | tail -1 ) > $OUT/train_60k.json 2>&1
( timeout 400 $T --steps 3 --points 300000 | tail -1 ) > $OUT/train_300k.json 2>&1
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof/t3 -o t -- $T --steps 2 --points 60000 > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof/t9 -o t -- $T --steps 6 --points 60000 > /dev/null 2>&1
cd $R
python tools/train_profile.py /tmp/prof/t3/t_results.db /tmp/prof/t9/t_results.db 2 6 | cut -c1-200 > $OUT/train_step_kernels.txt 2>&1
LIB=tools/probes/_build/libisf_hip_wgrad_compact.so
for shape in "40000 256 256" "120000 128 128" "300000 32 32" "300000 64 64"; do
  set -- $shape
  for lib in "" "--lib $LIB"; do
    timeout 120 python tools/wgrad_bench.py --rows $1 --cin $2 --cout $3 $lib 2>&1 | tail -1
  done
done > $OUT/wgrad_bench.txt 2>&1
( timeout 500 python -m pytest tests -x -q -m gpu -k "backward or grad or train" --isf-lib $LIB 2>&1 | tail -5 ) > $OUT/pytest_compact.txt
cat $OUT/train_60k.json $OUT/train_300k.json; head -50 $OUT/train_step_kernels.txt; cat $OUT/wgrad_bench.txt; cat $OUT/pytest_compact.txt
