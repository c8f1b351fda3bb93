#!/bin/bash
# small-grid tile rule at the headline batch (B=4): 64-row tiles for the 256-column layers below 48 k rows; one-group waves
# for the 128-column layers
set -u
O=gpurun_out/r5c20; mkdir -p $O
P=tools/probes/_build
C="--no-cpu-baseline --no-cfg3 --no-cfg4 --no-cfg5 --no-pipelined"
for r in 1 2; do
python bench.py --steps 50 $C                               > $O/b4_base_$r.json 2>> $O/err.txt
python bench.py --steps 50 $C --lib $P/libisf_hip_small48.so  > $O/b4_small48_$r.json 2>> $O/err.txt
python bench.py --steps 50 $C --lib $P/libisf_hip_small128.so > $O/b4_small128_$r.json 2>> $O/err.txt
done
for f in $O/*.json; do echo $f; python tools/r5/line_brief.py < $f; done
