#!/bin/bash
# default (legacy NULL) stream vs a private launch stream for the callers of the library: headline and configs[2]
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06_stream_ab}; mkdir -p $OUT
export ISF_BENCH_FRAME_CACHE=/tmp/isf_frames
for rep in 1 2; do
  for f in "" "--launch-stream"; do
    timeout 600 python bench.py --config 3 --batch 2 --no-cpu-baseline $f 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config3  [$f]', d['ms_per_step'])"
    timeout 600 python bench.py --no-cfg3 --no-cfg4 --no-cfg5 --no-cpu-baseline --no-pipelined $f 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('headline [$f]', d['value'], d['ms_per_step'])"
  done
done | tee $OUT/ab.txt
