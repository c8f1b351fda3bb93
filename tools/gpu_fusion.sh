#!/bin/bash
# Runs on the GPU box: the HSF / IGF parity tests only (fast iteration), with per-test results.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_fusion.py -m gpu -q --maxfail=40 -p no:cacheprovider 2>&1 | grep -v "Warn" | tail -120 > gpurun_out/test_fusion.log
tail -60 gpurun_out/test_fusion.log
