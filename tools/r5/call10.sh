#!/bin/bash
# round 5, GPU call 10: the whole -m gpu tier + smoke() as the driver runs them, then the training step at both sizes
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5c10
mkdir -p $OUT
cd $R
( timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -8 ) > $OUT/pytest_gpu.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $OUT/smoke.txt
cd /tmp && export TMPDIR=/tmp
export MIOPEN_FIND_MODE=FAST
T="python $R/tools/train_step.py --autocast"
( timeout 300 $T --steps 10 --points 60000 2>/dev/null | grep ms_per_train_step ) > $OUT/train_60k.json
( timeout 400 $T --steps 10 --points 300000 2>/dev/null | grep ms_per_train_step ) > $OUT/train_300k.json
cd $R
cat $OUT/pytest_gpu.txt $OUT/smoke.txt; grep -ho '"ms_per_train_step": [0-9.]*\|"ms_each_step_gpu_clock": [^]]*\]' $OUT/train_*.json
