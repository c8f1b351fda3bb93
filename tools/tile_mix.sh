#!/bin/bash
# Full / half-tile mix of the sparse-conv launches (isf_spconv16.h, conv16_plan) against uniform tiles (--conv-diag 32),
# plus the workgroup-placement probe the plan's "per CU" dealing rests on:
#   gpurun --timeout 900 -- 'bash tools/tile_mix.sh'
set -u
mkdir -p gpurun_out
export ISF_BENCH_FRAME_CACHE=/tmp/isf_bench_frames
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value tools/probes/wg_placement.hip -o /tmp/wg_placement \
  && timeout 60 /tmp/wg_placement 2>&1 | tee gpurun_out/wg_placement.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value tools/probes/masked_gather.hip -o /tmp/masked_gather \
  && timeout 60 /tmp/masked_gather 2>&1 | tee gpurun_out/masked_gather.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "tile_mix or sharing or sparse_conv or encoder" 2>&1 | tail -4 | tee gpurun_out/tile_mix_tests.log
for mode in 0 32 0 32; do
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
    print(mode, "unreadable:", e)
PY
done 2>&1 | tee gpurun_out/tile_mix_bench.txt
