#!/bin/bash
# Knock-out decomposition of the sparse-conv time as one repeatable measurement:
#   gpurun --timeout 600 -- 'bash tools/conv_knockout.sh'      (5 bench runs: about 2 GPU-minutes)
# --conv-diag: 2 = no activation gathers, 4 = no weight DMA, 6 = neither, 8 = no main loop.  The outputs of the
# diagnostic kernels are garbage; only the per-kernel times are read.  The diagnostics run on the 4-wave workgroup shape.
set -u
mkdir -p gpurun_out
export ISF_BENCH_FRAME_CACHE=/tmp/isf_bench_frames     # the 4 x 300 k-point frames take 11 s of CPU to generate
for mode in 0 2 4 6 8; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --conv-diag $mode \
      > gpurun_out/knock_$mode.json 2> gpurun_out/knock_$mode.err
done
python - <<'PY'
import json
for mode, name in ((0, "full"), (2, "nogather"), (4, "nodma"), (6, "neither"), (8, "noloop")):
    try:
        d = json.loads(open(f"gpurun_out/knock_{mode}.json").read().strip().splitlines()[-1])
        print("%-9s conv ms/step %-8s step ms %-8s" % (name, d["roofline"]["conv_ms_per_step"], d["ms_per_step"]),
              {k.replace("spconv_mfma", ""): v["ms"] for k, v in d["roofline"]["per_kernel"].items()})
    except Exception as e:   # noqa: BLE001
        print(name, "unreadable:", e)
PY
