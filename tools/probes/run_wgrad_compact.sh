#!/bin/bash
# GPU box (next round): the compacted-rows dW kernel (isf_spconv_bwd.hip under -DISF_WGRAD_COMPACT=1) against the shipped one.
# Build the side library FIRST, in the container:  bash tools/probes/build_side_lib.sh isf_spconv_bwd.hip ISF_WGRAD_COMPACT=1 wgrad_compact
#   bash tools/probes/run_wgrad_compact.sh [out dir under gpurun_out/]
set -u
OUT=gpurun_out/${1:-wgrad_compact}
LIB=tools/probes/_build/libisf_hip_wgrad_compact.so
mkdir -p $OUT
( timeout 600 python -m pytest tests -x -q -m gpu -k "backward or grad or train" --isf-lib $LIB 2>&1 | tail -5 ) | tee $OUT/pytest_compact.txt
for shape in "40000 256 256" "120000 128 128" "300000 32 32" "300000 64 64"; do
  set -- $shape
  for lib in "" "--lib $LIB"; do
    timeout 120 python tools/wgrad_bench.py --rows $1 --cin $2 --cout $3 $lib 2>&1 | tail -1
  done
done | tee $OUT/wgrad_bench.txt
