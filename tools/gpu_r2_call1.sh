#!/bin/bash
# Round 2, GPU call 1: the gpu_next tier (every failure listed), the vmcnt ordering probe, the knock-out / variant sweep,
# the single-pass diagnostic, then the validated tier.
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu_next -p no:cacheprovider 2>&1 | tail -80 > gpurun_out/gpu_next_tests.log
tail -30 gpurun_out/gpu_next_tests.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/vmcnt_probe.hip -o /tmp/vmcnt_probe \
  && timeout 120 /tmp/vmcnt_probe | tee gpurun_out/vmcnt_probe.json
bash tools/conv_knockout.sh 2>&1 | tail -40 > gpurun_out/knockout.log
cat gpurun_out/knockout.log
ISF_BENCH_FRAME_CACHE=/tmp/isf_bench_frames timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --f16 > gpurun_out/bench_f16_diag.json 2> gpurun_out/bench_f16_diag.err
tail -c 600 gpurun_out/bench_f16_diag.json
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -5 | tee gpurun_out/gpu_tests.log
