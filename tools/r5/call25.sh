#!/bin/bash
set -u
O=gpurun_out/r5c25; mkdir -p $O
timeout 26 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lidar_branch_dma or lds_staged or lidar_branch_line or autograd_switch or independent_workspaces" --tb=short 2>&1 | tail -25 > $O/more.txt
cat $O/more.txt
