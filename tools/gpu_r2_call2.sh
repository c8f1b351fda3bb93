#!/bin/bash
# Round 2, GPU call 2: ring conv kernel -- bit-equality tests, per-layer sweep, whole-bench comparison
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "ring or full_size_numeric" 2>&1 | tail -30 > gpurun_out/ring_tests.log
tail -15 gpurun_out/ring_tests.log
timeout 900 python tools/conv_sweep.py --json gpurun_out/conv_sweep.json > gpurun_out/conv_sweep.log 2>&1
tail -45 gpurun_out/conv_sweep.log
export ISF_BENCH_FRAME_CACHE=/tmp/isf_bench_frames
for r in "" "0,0,0" "0,0,3" "4,2,2" "4,3,2"; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ${r:+--ring $r} > gpurun_out/bench_ring_${r:-off}.json 2> gpurun_out/bench_ring_${r:-off}.err
  python - "$r" <<'PY'
import json, sys
r = sys.argv[1] or "off"
try:
    d = json.loads(open(f"gpurun_out/bench_ring_{r}.json").read().strip().splitlines()[-1])
    print("ring", r, d["value"], "frames/s", d["ms_per_step"], "ms/step conv", d["roofline"]["conv_ms_per_step"],
          {k.replace("spconv_mfma", ""): v["ms"] for k, v in d["roofline"]["per_kernel"].items()})
except Exception as e:
    print("ring", r, "unreadable", e)
PY
done
