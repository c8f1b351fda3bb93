#!/bin/bash
# Whole GPU tier + smoke, as the driver runs them at round end.
set -u
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/smoke.log
