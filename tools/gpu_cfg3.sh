#!/bin/bash
# cfg3 (full HSF + IGF forward, B = 2) check: parity tests named by $2, then bench.py --config 3 $3 times.
# Usage (GPU box): bash tools/gpu_cfg3.sh <tag> [pytest -k expression | NONE] [reps]
set -u
TAG=${1:-r06_cfg3}
KEXPR=${2:-NONE}
REPS=${3:-2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export ISF_BENCH_FRAME_CACHE=/tmp/isf_frames
if [ "$KEXPR" != "NONE" ]; then
  ( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py tests/test_gpu_fusion.py -x -q -m gpu -k "$KEXPR" 2>&1 | tail -15 ) > $OUT/pytest.txt
  tail -5 $OUT/pytest.txt
fi
for rep in $(seq 1 $REPS); do
  timeout 600 python bench.py --config 3 --steps 30 --warmup 6 --no-cpu-baseline > $OUT/cfg3_$rep.json 2> $OUT/cfg3_$rep.err
  python - <<PY
import json
try:
    l = [json.loads(x) for x in open("$OUT/cfg3_$rep.json").read().splitlines() if x.startswith("{")][-1]
    print("cfg3 rep $rep:", l["value"], "frames/s", l["ms_per_step"], "ms; launches", l.get("launches_per_forward"), l["roofline"].get("stages_ms"))
except Exception as e:
    print("cfg3 rep $rep FAILED", repr(e))
PY
done 2>&1 | tee $OUT/summary.txt
