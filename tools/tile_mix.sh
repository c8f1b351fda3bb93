#!/bin/bash
# A/B of the sparse-conv launch / kernel options on the bench (B=4 x 300 k points, 2 frame sets rotated):
#   --conv-diag 0 default, 32 uniform tiles (no full / half mix), 16 no neighbour sharing, 48 both off
#   gpurun --timeout 900 -- 'bash tools/tile_mix.sh "0 32 16 48"'
set -u
mkdir -p gpurun_out
export ISF_BENCH_FRAME_CACHE=/tmp/isf_bench_frames
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "tile_mix or sharing or sparse_conv or encoder or full_size" 2>&1 | tail -4 | tee gpurun_out/tile_mix_tests.log
for mode in ${1:-0 32 0 32}; do
  timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --conv-diag $mode \
      > gpurun_out/mix_$mode.json 2> gpurun_out/mix_$mode.err
  python - "$mode" <<'PY'
import json, sys
mode = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/mix_{mode}.json").read().strip().splitlines()[-1])
    print("conv-diag %-3s %s frames/s %s ms/step conv %s" % (mode, d["value"], d["ms_per_step"], d["roofline"]["conv_ms_per_step"]),
          {k.replace("spconv_mfma", ""): v["ms"] for k, v in d["roofline"]["per_kernel"].items()})
except Exception as e:   # noqa: BLE001
    print(mode, "unreadable:", e, open(f"gpurun_out/mix_{mode}.err").read()[-600:])
PY
done 2>&1 | tee gpurun_out/tile_mix_bench.txt
