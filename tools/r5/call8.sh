#!/bin/bash
# round 5, GPU call 8: single-pass f16 (autocast) mode of the sparse-conv training kernels, chunked attention key gradient,
# staggered-issue bit test; training step timing + profile
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5c8
mkdir -p $OUT
cd $R
( timeout 1200 python -m pytest tests/test_gpu_widened.py -x -q -k "half_mode or under_autocast or attention_backward or wgrad_f16x3 or sparse_conv_backward or training_step or without_the_f16" 2>&1 | tail -8 ) > $OUT/pytest_a.txt
( timeout 1500 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8 ) > $OUT/pytest_train.txt
( timeout 300 python tools/wgrad_bench.py --level 3 --cin 256 --cout 256 --f16x3 2>&1 | tail -1 ) > $OUT/wgrad.txt
cd /tmp && export TMPDIR=/tmp
export MIOPEN_FIND_MODE=FAST
T="python $R/tools/train_step.py --autocast"
( timeout 300 $T --steps 6 --points 60000 2>/dev/null | grep ms_per_train_step ) > $OUT/train_60k.json
( timeout 400 $T --steps 6 --points 300000 2>/dev/null | grep ms_per_train_step ) > $OUT/train_300k.json
( timeout 400 python $R/tools/train_step.py --steps 6 --points 300000 2>/dev/null | grep ms_per_train_step ) > $OUT/train_300k_fp32.json
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof/t3 -o t -- $T --steps 2 --points 60000 > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof/t9 -o t -- $T --steps 6 --points 60000 > /dev/null 2>&1
cd $R
python tools/train_profile.py /tmp/prof/t3/t_results.db /tmp/prof/t9/t_results.db 2 6 | cut -c1-220 > $OUT/train_step_kernels.txt 2>&1
cat $OUT/pytest_a.txt $OUT/pytest_train.txt $OUT/wgrad.txt; grep -ho '"points": [0-9]*\|"autocast_bf16": [a-z]*\|"ms_per_train_step": [0-9.]*' $OUT/train_*.json; head -40 $OUT/train_step_kernels.txt
