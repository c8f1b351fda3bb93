#!/bin/bash
set -u
O=gpurun_out/r5c24; mkdir -p $O
timeout 78 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile_order or tile_table or cu_unit" --tb=short 2>&1 | tail -25 > $O/tiles.txt
cat $O/tiles.txt
