#!/bin/bash
# round 5, final GPU call: soak + late tests, the judged line with rocprofv3 stats / PMC passes (tools/gpu_round_profile.sh),
# the training step's per-kernel profile at the BASELINE configs[3] size, the two-rank real-step rehearsal line
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5final
mkdir -p $OUT
cd $R
( timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity.py tests/test_gpu_widened.py -x -q -k "soak or one_column_block or dynamic_scatter or scatter" 2>&1 | tail -3 ) > $OUT/pytest_late.txt
( timeout 600 python bench.py --gpus 2 --backend gloo --real-step --steps 6 --warmup 2 --points 300000 --batch 2 2>/dev/null | grep '^{' ) > $OUT/rehearsal_2ranks.json
bash tools/gpu_round_profile.sh > $OUT/round_profile_stdout.txt 2>&1
cd /tmp && export TMPDIR=/tmp
export MIOPEN_FIND_MODE=FAST
T="python $R/tools/train_step.py --autocast"
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof/t3 -o t -- $T --steps 2 --points 300000 > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof/t9 -o t -- $T --steps 6 --points 300000 > /dev/null 2>&1
cd $R
python tools/train_profile.py /tmp/prof/t3/t_results.db /tmp/prof/t9/t_results.db 2 6 | cut -c1-220 > $OUT/train_step_kernels_300k.txt 2>&1
cat $OUT/pytest_late.txt; cut -c1-400 $OUT/rehearsal_2ranks.json; cut -c1-1500 gpurun_out/bench_line.json; head -25 $OUT/train_step_kernels_300k.txt
