#!/bin/bash
# small-grid tile rule (ISF_SMALL_ROWS side build) A/B: cfg3 (B=2) and the single-sweep latency line
set -u
O=gpurun_out/r5c18; mkdir -p $O
S=tools/probes/_build/libisf_hip_small.so
C="--no-cpu-baseline --no-cfg3 --no-cfg4 --no-cfg5 --no-pipelined"
for r in 1 2; do
python bench.py --config 3 --steps 30 $C            > $O/cfg3_base_$r.json 2> $O/err.txt
python bench.py --config 3 --steps 30 $C --lib $S   > $O/cfg3_small_$r.json 2>> $O/err.txt
done
python bench.py --batch 1 --steps 50 $C             > $O/b1_base.json 2>> $O/err.txt
python bench.py --batch 1 --steps 50 $C --lib $S    > $O/b1_small.json 2>> $O/err.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "conv" --isf-lib $S 2>&1 | tail -3 > $O/parity.txt
for f in $O/*.json; do echo $f; python tools/r5/line_brief.py < $f; done
cat $O/parity.txt
