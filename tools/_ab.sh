python -m pytest tests/test_gpu_parity.py tests/test_gpu_widened.py -x -q -m gpu -k "vfe or lidar_branch or full_size or scatter or config" 2>&1 | tail -3
for i in 1 2; do python bench.py --no-cpu-baseline --no-cfg3 --steps 40 > gpurun_out/ab_$i.json 2>/dev/null; python - <<PY
import json
l=json.loads(open("gpurun_out/ab_$i.json").read().strip().splitlines()[-1]); r=l["roofline"]
print(l["value"], l["ms_per_step"], r["conv_ms_per_step"])
PY
done
