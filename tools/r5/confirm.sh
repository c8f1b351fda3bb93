#!/bin/bash
# round 5: a driver-style default run on another fresh box (python bench.py with no flags), the line as it is judged
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r5confirm
cd $R
python bench.py 2>/dev/null | tail -1 > gpurun_out/r5confirm/bench_line.json
python - <<'PY'
import json
j = json.load(open("gpurun_out/r5confirm/bench_line.json"))
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["pipelined"]["value"], j["cfg3"]["value"], j["cfg5_f16"]["value"], j["cfg4_train"]["value"], j["cfg4_train"]["launches_per_step"])
PY
