"""Micro-benchmark of isf_linear_forward at the fusion encoder's shapes (one GPU)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isfusion_amd import fusion_ops as ops  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    M = 129600
    for K, N in ((128, 384), (128, 128), (256, 256)):
        Mk = M if K == 128 else M // 4
        x = torch.randn(Mk, K, device=dev)
        w = torch.randn(N, K, device=dev) * K ** -0.5
        b = torch.randn(N, device=dev)
        pl = ops.PackedLinear(w, b)
        tab = torch.randn(36, N, device=dev)
        idx = torch.randint(0, 36, (Mk,), device=dev, dtype=torch.int32)
        res = torch.randn(Mk, N, device=dev)
        ln = torch.nn.LayerNorm(N).to(dev) if N <= 256 else None
        gb = (Mk * K + Mk * N) * 4 / 1e9
        t0 = timed(lambda: ops.linear(x, pl))
        t1 = timed(lambda: ops.linear(x, pl, table=tab, index=idx))
        t2 = timed(lambda: ops.linear(x, pl, act=ops.ACT_GELU))
        t3 = timed(lambda: ops.linear(x, pl, residual=res, ln=ln)) if ln is not None else float("nan")
        tt = timed(lambda: torch.nn.functional.linear(x, w, b))
        print(f"M={Mk} K={K} N={N}: plain {t0:.1f} us ({gb / t0 * 1e6:.0f} GB/s)  +table {t1:.1f}  +gelu {t2:.1f}  "
              f"+res+LN {t3:.1f}  torch(hipBLASLt fp32) {tt:.1f}")


if __name__ == "__main__":
    main()
