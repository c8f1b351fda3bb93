#!/bin/bash
# First GPU call of the next round: run the tests written after round 1's GPU budget was spent (marker gpu_next),
# then the validated tier.   gpurun --timeout 900 -- 'bash tools/gpu_next.sh'
set -u
mkdir -p gpurun_out
python -m pytest tests -q -m gpu_next 2>&1 | tail -25 | tee gpurun_out/gpu_next_tests.log
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/gpu_tests.log
# hardware question behind the deeper conv prefetch ring: are LDS-DMA and register loads of one
# wave retired strictly in issue order under counted s_waitcnt?
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/vmcnt_probe.hip -o gpurun_out/vmcnt_probe \
  && timeout 120 gpurun_out/vmcnt_probe | tee gpurun_out/vmcnt_probe.json
# diagnostic: the same conv kernels with one MFMA pass instead of three (reduced precision, NOT the headline) -- shows
# how much of the conv time is the matrix pipe and how much is everything else
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --f16 > gpurun_out/bench_f16_diag.json 2> gpurun_out/bench_f16_diag.err
tail -c 600 gpurun_out/bench_default.json; tail -c 600 gpurun_out/bench_f16_diag.json
