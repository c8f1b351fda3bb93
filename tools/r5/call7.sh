#!/bin/bash
# round 5, GPU call 7: staggered issue phases (conv mode 65536 / encoder diagnostic 1048576) A/B on the headline workload,
# bit-identity; training step after the BN second-level fix
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5c7
mkdir -p $OUT
cd $R
export ISF_BENCH_FRAME_CACHE=/tmp/isf_bench_frames
( timeout 600 python - <<'PY' 2>&1 | tail -3
import sys, torch
sys.path.insert(0, ".")
import isfusion_amd as m
from isfusion_amd import synthetic
dev = torch.device("cuda", 0)
lb = m.LidarBranch().randomize_weights_(0).randomize_bn_(1).eval().to(dev)
for n, frames in ((3000, 1), (60000, 2), (300000, 4)):
    pl = [torch.from_numpy(synthetic.lidar_sweeps(900 + i, n)).to(dev) for i in range(frames)]
    want = lb(pl)
    got = lb(pl, conv_diag=1048576)
    print(n, "staggered == production bits:", bool(torch.equal(got, want)))
PY
) > $OUT/stagger_bits.txt
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-cfg3 --no-cfg4 --no-cfg5 --no-pipelined"
for rep in 1 2 3; do
  for d in 0 1048576; do
    echo "diag $d rep $rep: $(timeout 300 $B --conv-diag $d 2>/dev/null | tail -1 | python tools/r5/line_brief.py)"
  done
done > $OUT/ab_stagger.txt 2>&1
( timeout 600 python -m pytest tests/test_gpu_train.py -x -q -k "fused_bn or sync_bn" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 ) > $OUT/pytest_bn.txt
cd /tmp && export TMPDIR=/tmp
export MIOPEN_FIND_MODE=FAST
T="python $R/tools/train_step.py --autocast"
( timeout 300 $T --steps 6 --points 60000 2>/dev/null | grep ms_per_train_step ) > $OUT/train_60k.json
( timeout 400 $T --steps 6 --points 300000 2>/dev/null | grep ms_per_train_step ) > $OUT/train_300k.json
cd $R
cat $OUT/stagger_bits.txt $OUT/ab_stagger.txt $OUT/pytest_bn.txt; cut -c1-330 $OUT/train_60k.json $OUT/train_300k.json
