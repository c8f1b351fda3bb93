#!/bin/bash
# small-grid tile rule, second pass (half tiles off for one-group waves): single-sweep line A/B + parity on the side build
set -u
O=gpurun_out/r5c19; mkdir -p $O
S=tools/probes/_build/libisf_hip_small.so
C="--no-cpu-baseline --no-cfg3 --no-cfg4 --no-cfg5 --no-pipelined"
for r in 1 2; do
python bench.py --batch 1 --steps 50 $C             > $O/b1_base_$r.json 2>> $O/err.txt
python bench.py --batch 1 --steps 50 $C --lib $S    > $O/b1_small_$r.json 2>> $O/err.txt
done
python bench.py --batch 2 --steps 50 $C             > $O/b2_base.json 2>> $O/err.txt
python bench.py --batch 2 --steps 50 $C --lib $S    > $O/b2_small.json 2>> $O/err.txt
timeout 400 python -m pytest tests/test_gpu_parity.py -q -m gpu --isf-lib $S 2>&1 | tail -4 > $O/parity.txt
for f in $O/*.json; do echo $f; python tools/r5/line_brief.py < $f; done
cat $O/parity.txt
