"""Summarise bench.py outputs of an A/B run: python tools/ab_summary.py gpurun_out/<tag>/bench_diag*.json
Reads the final line (frames/s, ms) and the "headline_detail" line (per-kernel ms per step)."""
import json
import sys


def read(path):
    final, detail = None, None
    for ln in open(path).read().splitlines():
        if not ln.startswith("{"):
            continue
        d = json.loads(ln)
        if d.get("leg") == "headline_detail":
            detail = d
        elif "metric" in d and "leg" not in d:
            final = d
    return final, detail


for p in sys.argv[1:]:
    try:
        f, d = read(p)
        roof = (d or f)["roofline"]
        pk = roof.get("per_kernel", {})
        print(p.split("/")[-1], f["value"], "frames/s", f["ms_per_step"], "ms; conv", roof.get("conv_ms_per_step"),
              {k.replace("spconv_mfma", ""): v["ms"] for k, v in pk.items()})
    except Exception as e:
        print(p, "FAILED", repr(e))
