#!/bin/bash
# is the cfg3 leg of the default bench run slower than `--config 3` on its own?  same box, alternating
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06_cfg3_ctx}; mkdir -p $OUT
export ISF_BENCH_FRAME_CACHE=/tmp/isf_frames
for rep in 1 2; do
  timeout 600 python bench.py --config 3 --batch 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config3 alone      ', d['ms_per_step'], d['steps'], d['warmup'])"
  timeout 900 python bench.py --no-cfg4 --no-cfg5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default (no 4/5)   ', d['cfg3_ms'], d['value'], d.get('pipelined_fps'), d['cpu_baseline']['value'])"
done | tee $OUT/ctx.txt
timeout 1200 python bench.py > $OUT/default_full.txt 2>$OUT/default_full.err; tail -1 $OUT/default_full.txt | cut -c1-900
