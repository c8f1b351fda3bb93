#!/bin/bash
# Knock-out decomposition of the sparse-conv time and a sweep over every queued kernel variant
# (DESIGN.md section 10), as one repeatable measurement:
#   gpurun --timeout 1500 -- 'bash tools/conv_knockout.sh'      (22 bench runs + 4 micro-checks: about 10 GPU-minutes)
# ISF_CONV16_DIAG: 2 = no activation gathers, 4 = no weight DMA, 6 = neither, 8 = no main loop.  The outputs of the
# diagnostic kernels are garbage; only `conv_ms_per_step` is read.  All variants use the default workgroup shape
# (ISF_CONV16_NW=4), so the reference line is measured with that shape too.
set -u
mkdir -p gpurun_out
export ISF_BENCH_FRAME_CACHE=/tmp/isf_bench_frames     # the 4 x 300 k-point frames take 11 s of CPU to generate
run() {  # name, env...
  local name=$1; shift
  env ISF_CONV16_NW=4 "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline \
      > gpurun_out/knock_$name.json 2> gpurun_out/knock_$name.err
}
run full
run nogather ISF_CONV16_DIAG=2
run nodma ISF_CONV16_DIAG=4
run neither ISF_CONV16_DIAG=6
run noloop ISF_CONV16_DIAG=8
run prio ISF_CONV16_PRIO=1          # experiment (valid results): s_setprio around the MFMA block
run tepi ISF_CONV16_TEPI=1          # experiment (valid results): transposed accumulators, LDS-free epilogue
run tepi_prio ISF_CONV16_TEPI=1 ISF_CONV16_PRIO=1
run tps ISF_CONV16_TPS=1            # experiment (valid results): 4 / 2 taps per step for the narrow layers
run tps_tepi ISF_CONV16_TPS=1 ISF_CONV16_TEPI=1
run wind ISF_CONV16_WIND=1          # experiment (valid results): wave-independent loop for the narrow layers
run wind_tepi ISF_CONV16_WIND=1 ISF_CONV16_TEPI=1
run vepi ISF_CONV16_VEPI=1          # experiment (valid results): vector loads of the BN scale / shift in the epilogue
run vepi_wind ISF_CONV16_VEPI=1 ISF_CONV16_WIND=1
run rg4 ISF_CONV16_RG=4             # existing variant: 64-row waves
run rg1 ISF_CONV16_RG=1             # experiment: 64-row workgroups for the <= 64-column layers
# workgroup-shape variants: these are NOT forced to 4 waves (the reference for them is the production heuristic)
runfree() { local name=$1; shift; env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline \
      > gpurun_out/knock_$name.json 2> gpurun_out/knock_$name.err; }
runfree heuristic
runfree nw8 ISF_CONV16_NW=8         # experiment: 8-wave (256-row) workgroups for the 128-column layers
runfree nw16 ISF_CONV16_NW=16
runfree deep ISF_CONV16_DEEP=1      # experiment: 128-row tiles on 8 waves x 16 rows for the 128-column layers of the small levels
runfree wind_free ISF_CONV16_WIND=1   # narrow layers wave-independent, the others on the production heuristic
runfree tps_free ISF_CONV16_TPS=1
python - <<'PY'
import json
rows = []
for name in ("full", "nogather", "nodma", "neither", "noloop", "prio", "tepi", "tepi_prio", "tps", "tps_tepi", "wind", "wind_tepi", "vepi", "vepi_wind", "rg4", "rg1", "heuristic", "nw8",
             "nw16", "deep", "wind_free", "tps_free"):
    try:
        d = json.loads(open(f"gpurun_out/knock_{name}.json").read().strip().splitlines()[-1])
        rows.append((name, d["roofline"]["conv_ms_per_step"], d["ms_per_step"]))
        print(name, {k.replace("spconv_mfma", ""): v["ms"] for k, v in d["roofline"]["per_kernel"].items()})
    except Exception as e:   # noqa: BLE001
        rows.append((name, None, str(e)))
for r in rows:
    print("%-10s conv ms/step %-8s step ms %s" % r)
open("gpurun_out/conv_knockout.json", "w").write(json.dumps(rows))
PY
# Point-to-Grid: default vs the pipelined variant (digest, max, ms per call)
python tools/p2g_variant_check.py 15000 | tee gpurun_out/p2g_default.txt
ISF_P2G_PIPE=1 python tools/p2g_variant_check.py 15000 | tee gpurun_out/p2g_pipe.txt
# fused linear kernel: default vs batched epilogue loads (timings at the encoder's shapes)
python tools/linear_variant_check.py gpurun_out/linear_default.npz | tee gpurun_out/linear_default.txt
ISF_LINEAR_VEPI=1 python tools/linear_variant_check.py gpurun_out/linear_vepi.npz | tee gpurun_out/linear_vepi.txt
rm -f gpurun_out/linear_default.npz gpurun_out/linear_vepi.npz
