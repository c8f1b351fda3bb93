#!/bin/bash
# A side build of libisf_hip.so with ONE source file compiled under a probe macro; the other objects are the in-tree
# build's (make -C is-fusion_amd/csrc first).  The shipped library is never touched.
#   bash tools/probes/build_side_lib.sh isf_spconv_dma.hip ISF_DMA_KNOCKOUT=1 [name]
#   -> tools/probes/_build/libisf_hip_<name>.so   (load with bench.py --lib / tools/wgrad_bench.py --lib, or
#      isfusion_amd._lib.LIB_PATH = ... before the first call)
set -eu
SRC=$1
DEF=$2
NAME=${3:-$(echo "$DEF" | tr '=' '_' | tr 'A-Z' 'a-z')}
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/is-fusion_amd/csrc
B=$R/tools/probes/_build
mkdir -p $B
OBJS=$(ls $C/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I$R/include -I$C -Wall -Wno-unused-result \
    -D$DEF -c $C/$SRC -o $B/${SRC%.hip}_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libisf_hip_$NAME.so $OBJS $B/${SRC%.hip}_$NAME.o
rm -f $B/${SRC%.hip}_$NAME.o
echo built $B/libisf_hip_$NAME.so
