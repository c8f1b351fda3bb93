#!/bin/bash
# Side builds of libisf_hip.so whose narrow-layer kernel (isf_spconv_dma.hip) leaves a part of its step out
# (-DISF_DMA_KNOCKOUT=<bits>: 1 no row gathers, 2 no weight staging, 4 no MFMAs, 8 no per-step barrier), for
#   python bench.py --lib tools/probes/_build/libisf_hip_ko<bits>.so --steps 20 --warmup 5 --no-cfg3 --no-cfg4 --no-cfg5 \
#       --no-pipelined --no-cpu-baseline
# The other objects are the in-tree build's (make -C is-fusion_amd/csrc first).  Usage: bash tools/probes/build_dma_knockouts.sh "1 2 3 4 8 7"
set -eu
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/is-fusion_amd/csrc
B=$R/tools/probes/_build
mkdir -p $B
OBJS=$(ls $C/*.o | grep -v isf_spconv_dma.o)
for ko in ${1:-"1 2 3 4 8 7"}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I$R/include -I$C -Wall -Wno-unused-result \
      -DISF_DMA_KNOCKOUT=$ko -c $C/isf_spconv_dma.hip -o $B/isf_spconv_dma_ko$ko.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libisf_hip_ko$ko.so $OBJS $B/isf_spconv_dma_ko$ko.o
  rm -f $B/isf_spconv_dma_ko$ko.o
  echo built $B/libisf_hip_ko$ko.so
done
