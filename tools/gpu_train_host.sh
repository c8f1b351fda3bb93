#!/bin/bash
# host-side profile of the training step (cProfile over tools/train_step.py): where the Python / launch time of a step goes
TAG=${1:-r06_train_host}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export MIOPEN_FIND_MODE=FAST
timeout 900 python -m cProfile -o /tmp/train.prof tools/train_step.py --batch 2 --points 300000 --autocast --steps 10 > $OUT/line.txt 2>/dev/null
python - <<'P' > $OUT/host_profile.txt
import pstats
p = pstats.Stats('/tmp/train.prof')
p.sort_stats('tottime').print_stats(45)
p.sort_stats('cumulative').print_stats(70)
P
cut -c1-300 $OUT/line.txt; head -70 $OUT/host_profile.txt | cut -c1-160
