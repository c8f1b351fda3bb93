"""Fused linear kernel on seeded cases covering every epilogue / layout combination -> outputs saved to an .npz and
the timings of the fusion encoder's big shapes printed; used to check ISF_LINEAR_VEPI (read once per process) against
the default epilogue.

    ISF_LINEAR_VEPI=1 python tools/linear_variant_check.py out.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

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
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)   # noqa: E731
    out = {}
    # (rows, K, N): ragged row counts, every K the kernel is built for, N below / at / above one column chunk
    for M, K, N in ((1000, 32, 48), (777, 64, 128), (4097, 128, 384), (5000, 128, 128), (2500, 256, 256), (200, 128, 256)):
        x, w, b = rnd(M, K), rnd(N, K) * K ** -0.5, rnd(N)
        pl = ops.PackedLinear(w, b)
        tab = rnd(36, N)
        idx = torch.randint(0, 36, (M,), generator=g).to(dev).int()
        res = rnd(M, N)
        tag = f"{M}x{K}x{N}"
        out[tag + ".plain"] = ops.linear(x, pl)
        out[tag + ".table"] = ops.linear(x, pl, table=tab, index=idx)
        out[tag + ".gelu"] = ops.linear(x, pl, act=ops.ACT_GELU)
        out[tag + ".relu_res"] = ops.linear(x, pl, act=ops.ACT_RELU, residual=res)
        if N <= 256:
            ln = torch.nn.LayerNorm(N).to(dev)
            with torch.no_grad():
                ln.weight.copy_(rnd(N) * 0.2 + 1.0)
                ln.bias.copy_(rnd(N) * 0.1)
            out[tag + ".res_ln"] = ops.linear(x, pl, residual=res, ln=ln)
            out[tag + ".all"] = ops.linear(x, pl, table=tab, index=idx, act=ops.ACT_GELU, residual=res, ln=ln)
    # channels-first input / residual / output ([B, C, H, W] maps, hw % 4 == 0)
    B, S = 2, 12
    for K, N in ((128, 128), (256, 128)):
        xm, w, b = rnd(B, K, S, S), rnd(N, K) * K ** -0.5, rnd(N)
        pl = ops.PackedLinear(w, b)
        resm = rnd(B, N, S, S)
        tag = f"cf{K}x{N}"
        out[tag + ".in"] = ops.linear(xm, pl)
        out[tag + ".in_out"] = ops.linear(xm, pl, act=ops.ACT_RELU, out_nchw=(B, S, S))
        out[tag + ".in_res_out"] = ops.linear(xm, pl, residual=resm, out_nchw=(B, S, S))
    np.savez(sys.argv[1] if len(sys.argv) > 1 else "/tmp/linear_variant.npz", **{k: v.cpu().numpy() for k, v in out.items()})
    # timings at the encoder's shapes
    M = 129600
    x, w, b = rnd(M, 128), rnd(384, 128) * 128 ** -0.5, rnd(384)
    pl = ops.PackedLinear(w, b)
    tab, idx = rnd(36, 384), torch.randint(0, 36, (M,), generator=g).to(dev).int()
    w2, b2, res = rnd(128, 128) * 128 ** -0.5, rnd(128), rnd(M, 128)
    pl2, ln = ops.PackedLinear(w2, b2), torch.nn.LayerNorm(128).to(dev)
    print("us: 128->384 plain %.1f  +table %.1f  128->128 +res+LN %.1f  +gelu %.1f" % (
        timed(lambda: ops.linear(x, pl)), timed(lambda: ops.linear(x, pl, table=tab, index=idx)),
        timed(lambda: ops.linear(x, pl2, residual=res, ln=ln)), timed(lambda: ops.linear(x, pl2, act=ops.ACT_GELU))))


if __name__ == "__main__":
    main()
