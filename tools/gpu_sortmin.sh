export ISF_BENCH_FRAME_CACHE=/tmp/isf_frames
mkdir -p gpurun_out/r06_sortmin
for rep in 1 2; do
for m in 4096 32768 100000000; do
  ISF_ROW_SORT_MIN_ROWS=$m timeout 600 python bench.py --config 3 --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r06_sortmin/cfg3_$m_$rep.json
  python - <<PY
import json
l=[json.loads(x) for x in open("gpurun_out/r06_sortmin/cfg3_$m_$rep.json") if x.startswith("{")][-1]
print("sort_min $m rep $rep cfg3:", l["value"], "frames/s", l["ms_per_step"], "ms; launches", l.get("launches_per_forward"), "lidar", l["roofline"]["stages_ms"]["lidar_branch"])
PY
done
done
bash tools/gpu_r4_ab.sh r06_narrowtiles NONE "0 268435456" "1 2"
