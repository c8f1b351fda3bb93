#!/bin/bash
# round 5, GPU call 4: the f16x3 dW kernel -- parity tests, its rate, the training step with it (timing + per-kernel profile),
# the two-rank rehearsals
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5c4
mkdir -p $OUT
cd $R
( timeout 900 python -m pytest tests/test_gpu_widened.py -x -q -k "pair_lists or wgrad_f16x3 or without_the_f16_wgrad or sparse_conv_backward or training_step" 2>&1 | tail -15 ) > $OUT/pytest_wgrad.txt
for shape in "40000 256 256" "120000 128 128" "300000 64 64" "300000 32 32"; do
  set -- $shape
  timeout 120 python tools/wgrad_bench.py --rows $1 --cin $2 --cout $3 2>&1 | tail -1
  timeout 120 python tools/wgrad_bench.py --rows $1 --cin $2 --cout $3 --f16x3 2>&1 | tail -1
done > $OUT/wgrad_bench.txt 2>&1
( timeout 900 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -15 ) > $OUT/pytest_train.txt
cd /tmp && export TMPDIR=/tmp
export MIOPEN_FIND_MODE=FAST
T="python $R/tools/train_step.py --autocast"
( timeout 300 $T --steps 3 --points 60000 2>/dev/null | tail -1 ) > $OUT/train_60k.json
( timeout 400 $T --steps 3 --points 300000 2>/dev/null | tail -1 ) > $OUT/train_300k.json
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof/t3 -o t -- $T --steps 2 --points 60000 > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof/t9 -o t -- $T --steps 6 --points 60000 > /dev/null 2>&1
cd $R
python tools/train_profile.py /tmp/prof/t3/t_results.db /tmp/prof/t9/t_results.db 2 6 | cut -c1-220 > $OUT/train_step_kernels.txt 2>&1
cat $OUT/pytest_wgrad.txt $OUT/wgrad_bench.txt $OUT/pytest_train.txt $OUT/train_60k.json $OUT/train_300k.json; head -70 $OUT/train_step_kernels.txt
