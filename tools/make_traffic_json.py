"""profiles/rNN_vM_round_profile.txt (tools/gpu_round_profile.sh) -> profiles/rNN_vM_traffic.json read by bench.py.
    python tools/make_traffic_json.py profiles/r01_v7_round_profile.txt"""
import json
import re
import sys

src = sys.argv[1]
out = {}
for line in open(src):
    m = re.match(r"^(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([\d.]+)", line)
    if m and ("spconv_f16x3_kernel" in m.group(1) or "spconv_dma_kernel" in m.group(1)):
        d = out.setdefault(m.group(1).strip(), {})
        kb = float(m.group(4))
        if m.group(2) == "FETCH_SIZE":
            d["fetch_bytes"] = kb * 1024 * 2   # gfx950: FETCH_SIZE counts half of wide reads (MI355X_MICROARCH.md)
        else:
            d["write_bytes"] = kb * 1024
        d["dispatches"] = int(m.group(3))
dst = src.replace("_round_profile.txt", "_traffic.json")
json.dump({"source": f"{src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --steps 10 --warmup 3; "
                     "FETCH_SIZE x2 per the gfx950 note of MI355X_MICROARCH.md)", "per_launch": out}, open(dst, "w"), indent=1)
print("wrote", dst, len(out), "kernels")
