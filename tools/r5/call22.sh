#!/bin/bash
# small-launch rule as the default: parity suite, cfg3 line, smoke, then as much of the rest of the GPU suite as fits
set -u
O=gpurun_out/r5c22; mkdir -p $O
C="--no-cpu-baseline --no-cfg3 --no-cfg4 --no-cfg5 --no-pipelined"
timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4 > $O/parity.txt
python bench.py --config 3 --steps 30 $C > $O/cfg3.json 2> $O/err.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
timeout 215 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_widened.py tests/test_gpu_fusion.py -x -q -m gpu -k "not soak" 2>&1 | tail -4 > $O/rest.txt
cat $O/parity.txt; python tools/r5/line_brief.py < $O/cfg3.json; tail -2 $O/smoke.txt; cat $O/rest.txt
