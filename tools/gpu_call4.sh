#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_train.py -q -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/call4_train_tests.log
tail -25 gpurun_out/call4_train_tests.log
timeout 300 python tools/train_step.py --steps 3 > gpurun_out/train_step.json 2> gpurun_out/train_step.err; tail -2 gpurun_out/train_step.json; tail -3 gpurun_out/train_step.err
export ISF_BENCH_FRAME_CACHE=/tmp/isf_bench_frames
timeout 600 python bench.py --config 3 > gpurun_out/bench_cfg3.json 2> gpurun_out/bench_cfg3.err; tail -c 3000 gpurun_out/bench_cfg3.json; tail -3 gpurun_out/bench_cfg3.err
timeout 600 python bench.py > gpurun_out/bench_cfg2.json 2> gpurun_out/bench_cfg2.err; tail -c 2500 gpurun_out/bench_cfg2.json; tail -3 gpurun_out/bench_cfg2.err
