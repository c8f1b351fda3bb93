#!/bin/bash
# round 5, GPU call 9: lane-per-row attention backward kernels; training step at both sizes with per-step clocks; the
# staggered / one-column bit tests in their final form
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5c9
mkdir -p $OUT
cd $R
( timeout 1200 python -m pytest tests/test_gpu_widened.py tests/test_gpu_parity.py -x -q -k "attention_backward or one_column_block" 2>&1 | tail -4 ) > $OUT/pytest_a.txt
( timeout 1500 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 ) > $OUT/pytest_train.txt
cd /tmp && export TMPDIR=/tmp
export MIOPEN_FIND_MODE=FAST
T="python $R/tools/train_step.py --autocast"
( timeout 300 $T --steps 8 --points 60000 2>/dev/null | grep ms_per_train_step ) > $OUT/train_60k.json
( timeout 400 $T --steps 8 --points 300000 2>/dev/null | grep ms_per_train_step ) > $OUT/train_300k.json
( timeout 400 python $R/tools/train_step.py --steps 8 --points 300000 2>/dev/null | grep ms_per_train_step ) > $OUT/train_300k_fp32.json
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof/t3 -o t -- $T --steps 2 --points 300000 > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof/t9 -o t -- $T --steps 6 --points 300000 > /dev/null 2>&1
cd $R
python tools/train_profile.py /tmp/prof/t3/t_results.db /tmp/prof/t9/t_results.db 2 6 | cut -c1-220 > $OUT/train_step_kernels_300k.txt 2>&1
cat $OUT/pytest_a.txt $OUT/pytest_train.txt; grep -ho '"points": [0-9]*\|"autocast_bf16": [a-z]*\|"ms_per_train_step": [0-9.]*\|"ms_each_step_gpu_clock": [^]]*\]' $OUT/train_*.json | paste - - - -; head -30 $OUT/train_step_kernels_300k.txt
