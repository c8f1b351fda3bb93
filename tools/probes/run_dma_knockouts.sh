#!/bin/bash
# GPU box: per-layer times of the narrow layers with parts of the LDS-DMA kernel's step left out (see build_dma_knockouts.sh)
set -u
OUT=gpurun_out/${1:-r04_dma_knockouts}
mkdir -p $OUT
export ISF_BENCH_FRAME_CACHE=/tmp/isf_frames
KOS=${2:-"1 2 3 4 8 7 15"}
for ko in 0 $KOS; do
  LIB=""
  [ "$ko" != "0" ] && LIB="--lib tools/probes/_build/libisf_hip_ko$ko.so"
  timeout 120 python bench.py $LIB --steps 20 --warmup 5 --no-cfg3 --no-cfg4 --no-cfg5 --no-pipelined --no-cpu-baseline \
      > $OUT/bench_ko$ko.json 2> $OUT/bench_ko$ko.err
  python - <<PY
import json
try:
    l = json.loads(open("$OUT/bench_ko$ko.json").read().strip().splitlines()[-1])
    pk = l["roofline"]["per_kernel"]
    print("knockout", $ko, l["value"], "frames/s;", {k.replace("spconv_mfma", ""): round(v["ms"], 3) for k, v in pk.items()})
except Exception as e:
    print("knockout", $ko, "FAILED", e)
PY
done 2>&1 | tee $OUT/summary.txt
