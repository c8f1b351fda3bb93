#!/bin/bash
# VFE work: parity tests of the voxel feature encoder, then the headline step's kernel table (vfe_* / voxelize rows) and
# the frames/s of two headline runs.  Usage (GPU box): bash tools/gpu_vfe.sh <tag>
TAG=${1:-r06_vfe}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export ISF_BENCH_FRAME_CACHE=/tmp/isf_frames
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_widened.py -x -q -m gpu -k "vfe or lidar or config0 or config5 or encoder" 2>&1 | tail -8 ) > $OUT/pytest.txt
tail -3 $OUT/pytest.txt
for rep in 1 2; do
  timeout 300 python bench.py --steps 40 --warmup 8 --no-cfg3 --no-cfg4 --no-cfg5 --no-pipelined --no-cpu-baseline > $OUT/bench_$rep.json 2> $OUT/bench_$rep.err
  python tools/ab_summary.py $OUT/bench_$rep.json
done 2>&1 | tee $OUT/ab_summary.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof/vfe -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --no-cfg3 --no-cfg4 --no-cfg5 --no-pipelined --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py /tmp/prof/vfe/bench_results.db 2>/dev/null | grep -E "^kernel|vfe_|voxel|occ_|scan_u32|total GPU" | cut -c1-170 > $OUT/vfe_kernels.txt
cat $OUT/vfe_kernels.txt
