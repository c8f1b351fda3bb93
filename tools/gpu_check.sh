#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, smoke, short bench.  Logs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider 2>&1 | grep -v "^E  *+\|Warn" | tail -40 | tee gpurun_out/test_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -5 | tee gpurun_out/smoke.log
timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -3 > gpurun_out/bench.log
for v in $EXTRA_BENCH; do
  env $v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -1 > gpurun_out/bench_$v.log
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/bench*.log')):
    for line in open(f):
        if line.startswith('{'):
            d = json.loads(line); r = d['roofline']
            print(f, d['value'], 'frames/s', d['ms_per_step'], 'ms; conv', r['conv_ms_per_step'], 'ms;', r['kernel'], r['frac'])
            print('   ', {k.replace('spconv_mfma',''): v['ms'] for k, v in r['per_kernel'].items()})
PY
