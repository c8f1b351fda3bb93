#!/bin/bash
# round 5, GPU call 5: wgrad16 v2 (blocks of a workgroup share pairs, groups pinned to an XCD), training-path diet
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5c5
mkdir -p $OUT
cd $R
( timeout 900 python -m pytest tests/test_gpu_widened.py -x -q -k "pair_lists or wgrad_f16x3 or without_the_f16_wgrad or sparse_conv_backward or training_step" 2>&1 | tail -6 ) > $OUT/pytest_wgrad.txt
for args in "--rows 40000 --cin 256 --cout 256" "--level 3 --cin 256 --cout 256" "--level 2 --cin 128 --cout 128" "--rows 300000 --cin 64 --cout 64" "--rows 300000 --cin 32 --cout 32" "--rows 120000 --cin 64 --cout 128"; do
  timeout 200 python tools/wgrad_bench.py $args --f16x3 2>&1 | tail -1
done > $OUT/wgrad_bench.txt 2>&1
( timeout 900 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6 ) > $OUT/pytest_train.txt
cd /tmp && export TMPDIR=/tmp
export MIOPEN_FIND_MODE=FAST
T="python $R/tools/train_step.py --autocast"
( timeout 300 $T --steps 5 --points 60000 2>/dev/null | tail -1 ) > $OUT/train_60k.json
( timeout 400 $T --steps 5 --points 300000 2>/dev/null | tail -1 ) > $OUT/train_300k.json
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof/t3 -o t -- $T --steps 2 --points 60000 > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof/t9 -o t -- $T --steps 6 --points 60000 > /dev/null 2>&1
cd $R
python tools/train_profile.py /tmp/prof/t3/t_results.db /tmp/prof/t9/t_results.db 2 6 | cut -c1-220 > $OUT/train_step_kernels.txt 2>&1
cat $OUT/pytest_wgrad.txt $OUT/wgrad_bench.txt $OUT/pytest_train.txt $OUT/train_60k.json $OUT/train_300k.json; head -40 $OUT/train_step_kernels.txt
