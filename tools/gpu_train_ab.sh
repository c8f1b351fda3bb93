#!/bin/bash
# A/B of an environment toggle on the training step (tools/train_step.py at the configs[3] size): bash tools/gpu_train_ab.sh TAG VAR
TAG=${1:-r06_train_ab}; VAR=${2:-ISF_TRAIN_CHANNELS_LAST}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export MIOPEN_FIND_MODE=FAST
for v in 0 1 0 1; do
  env $VAR=$v timeout 600 python tools/train_step.py --batch 2 --points 300000 --autocast --steps 8 2>$OUT/err_$v.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['ms_per_train_step'], d['ms_each_step_gpu_clock'][2:], d['losses'][-1])"
done | tee $OUT/ab.txt
