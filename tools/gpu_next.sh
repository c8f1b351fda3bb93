#!/bin/bash
# First GPU call of the next round (about 8 GPU-minutes):   gpurun --timeout 1200 -- 'bash tools/gpu_next.sh'
#  1. the tests written after round 1's GPU budget was spent (marker gpu_next) -- every failure is listed, no -x
#  2. two questions that decide the conv restructuring: the vmcnt ordering probe and the
#     single-pass (one MFMA per product) diagnostic bench next to the default bench
#  3. the validated tier, to make sure nothing regressed
# Second call: `bash tools/conv_knockout.sh` (about 7 GPU-minutes): knock-out decomposition + every queued conv variant.
set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu_next -p no:cacheprovider 2>&1 | tail -60 | tee gpurun_out/gpu_next_tests.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/vmcnt_probe.hip -o gpurun_out/vmcnt_probe \
  && timeout 120 gpurun_out/vmcnt_probe | tee gpurun_out/vmcnt_probe.json
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --f16 > gpurun_out/bench_f16_diag.json 2> gpurun_out/bench_f16_diag.err
python - <<'PY'
import json
for name in ("bench_default", "bench_f16_diag"):
    try:
        d = json.loads(open(f"gpurun_out/{name}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(name, d["value"], d["unit"], "ms/step", d["ms_per_step"], "conv ms", r["conv_ms_per_step"],
              {k: v["ms"] for k, v in r["per_kernel"].items()})
    except Exception as e:   # noqa: BLE001
        print(name, "unreadable:", e)
PY
timeout 600 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/gpu_tests.log
