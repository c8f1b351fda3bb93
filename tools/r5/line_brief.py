"""one bench.py JSON line on stdin -> frames/s, ms per step, conv ms per kernel (A/B scripts of round 5)"""
import json
import sys

j = json.loads(sys.stdin.read())
pk = j.get("roofline", {}).get("per_kernel", {})
print(j["value"], j["ms_per_step"], "conv", j.get("roofline", {}).get("conv_ms_per_step"),
      " ".join(f"{k.replace('spconv_', '').replace('_kernel', '')}={v['ms']}" for k, v in pk.items()))
