#!/bin/bash
# round 6: dense 3x3 conv + BN stacks of the training step on the sparse-conv kernels (dense_train.py): parity tests, then
# an alternating A/B of the training step at the configs[3] size against the stock modules (MIOpen)
TAG=${1:-r06_dense_train}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export MIOPEN_FIND_MODE=FAST
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "dense_conv_stack or parameter_gradients or whole_path" 2>&1 | tail -15 > $OUT/pytest.txt
cat $OUT/pytest.txt
for v in "" "--stock-dense" "" "--stock-dense"; do
  timeout 600 python tools/train_step.py --batch 2 --points 300000 --autocast --steps 8 $v 2>$OUT/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dense=${v:-hip}', d['ms_per_train_step'], d['ms_each_step_gpu_clock'][2:], d['losses'][-1])"
done | tee $OUT/ab.txt
tail -5 $OUT/err.txt
