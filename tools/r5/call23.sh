#!/bin/bash
set -u
O=gpurun_out/r5c23; mkdir -p $O
timeout 100 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "small_launch" --tb=short 2>&1 | tail -25 > $O/small.txt
cat $O/small.txt
