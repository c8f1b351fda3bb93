#!/bin/bash
# small-launch rule: + one-group 8-wave workgroups for the 128-column layers below 64 k rows (B = 1, 2 and cfg3)
set -u
O=gpurun_out/r5c21; mkdir -p $O
P=tools/probes/_build
C="--no-cpu-baseline --no-cfg3 --no-cfg4 --no-cfg5 --no-pipelined"
for b in 1 2; do
python bench.py --batch $b --steps 50 $C --lib $P/libisf_hip_small.so     > $O/b${b}_small.json 2>> $O/err.txt
python bench.py --batch $b --steps 50 $C --lib $P/libisf_hip_smallboth.so > $O/b${b}_smallboth.json 2>> $O/err.txt
done
python bench.py --config 3 --steps 30 $C                                 > $O/cfg3_base.json 2>> $O/err.txt
python bench.py --config 3 --steps 30 $C --lib $P/libisf_hip_small.so     > $O/cfg3_small.json 2>> $O/err.txt
python bench.py --config 3 --steps 30 $C --lib $P/libisf_hip_smallboth.so > $O/cfg3_smallboth.json 2>> $O/err.txt
for f in $O/*.json; do echo $f; python tools/r5/line_brief.py < $f; done
