#!/bin/bash
# steady-state kernel table of the training step at the cfg4 size: two rocprofv3 traces of tools/train_step.py that differ
# in the number of steps (tools/train_profile.py) -> gpurun_out/<tag>/train_step.txt ; plus the plain timing line
TAG=${1:-r06_train}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp MIOPEN_FIND_MODE=FAST
R=$GRAFT_REPO_ROOT
ARGS="--batch 2 --points 300000 --autocast"
timeout 600 python $R/tools/train_step.py $ARGS --steps 6 2>/dev/null | tail -1 > $OUT/train_line.json
timeout 900 rocprofv3 --kernel-trace -d /tmp/tp/a -o t -- python $R/tools/train_step.py $ARGS --steps 3 > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace -d /tmp/tp/b -o t -- python $R/tools/train_step.py $ARGS --steps 6 > /dev/null 2>&1
cd $R
python tools/train_profile.py /tmp/tp/a/t_results.db /tmp/tp/b/t_results.db 3 6 | cut -c1-180 > $OUT/train_step.txt
cut -c1-400 $OUT/train_line.json; head -25 $OUT/train_step.txt
