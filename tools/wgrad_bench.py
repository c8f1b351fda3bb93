"""tools/wgrad_bench.py -- time isf_sparse_conv_backward_filter (sparse-conv dW) on synthetic level-3-like geometry.

    python tools/wgrad_bench.py [--rows 40000] [--cin 256] [--cout 256]

Prints ms per call and the fp32-MFMA fraction (2 * pairs * Cin * Cout flops against 157.3 TFLOP/s); with --f16x3 the
round-5 kernel (isf_sparse_conv_backward_filter_f16x3: f16x3 split on v_mfma_f32_16x16x32_f16, pair lists) and its
fraction of the f16x3 roofline (2500 / 3 TFLOP/s), the gradient split and the pair-list build timed beside it."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=40000)
    ap.add_argument("--cin", type=int, default=256)
    ap.add_argument("--cout", type=int, default=256)
    ap.add_argument("--lib", default="", help="A/B: load this build of libisf_hip.so instead of the in-tree one")
    ap.add_argument("--level", type=int, default=0, choices=[0, 2, 3],
                    help="2 / 3: the benchmark geometry's level-2 / level-3 SubM rulebook (B = 4 x 300 k-point synthetic "
                         "sweeps: 119 k / 40.7 k rows, 16 / 15.4 pairs per row) instead of the random-sparse grid")
    ap.add_argument("--f16x3", action="store_true", help="the f16 matrix-core dW (round 5) instead of the fp32-MFMA kernel")
    a = ap.parse_args()
    from isfusion_amd import _lib, spconv as sp
    if a.lib:
        _lib.LIB_PATH = os.path.abspath(a.lib)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    B, shape = 2, [12, 96, 96]
    while a.rows > 0.4 * B * shape[0] * shape[1] * shape[2]:
        shape = [shape[0], shape[1] * 2, shape[2] * 2]
    cells = B * int(np.prod(shape))
    lin = np.sort(rng.choice(cells, a.rows, replace=False))
    D, H, W = shape
    idx = np.stack([lin // (D * H * W), (lin // (H * W)) % D, (lin // W) % H, lin % W], 1).astype(np.int32)
    rb = sp.build_rulebook(torch.from_numpy(idx).to(dev), B, shape, [3, 3, 3], [1, 1, 1], [1, 1, 1], True)
    if a.level:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import bench
        from conv_trace import level_rulebooks
        rb = level_rulebooks([torch.from_numpy(p).to(dev) for p in bench.make_frames(0, 1, 4, 300000, 0)], 4, a.level)
        a.rows = rb.num_in
    pairs = int((rb.nbr.view(27, rb.stride)[:, :rb.num_out] >= 0).sum().item())
    x = torch.randn(a.rows, a.cin, device=dev)
    g = torch.randn(rb.num_out, a.cout, device=dev)
    dw = torch.empty(27, a.cin, a.cout, device=dev)
    lib = _lib.load()

    def call():
        _lib.check(lib.isf_sparse_conv_backward_filter(_lib.ptr(x), rb.num_in, a.cin, _lib.ptr(g), rb.num_out, a.cout,
                                                       _lib.ptr(rb.nbr), rb.stride, 27, _lib.ptr(dw), _lib.stream()),
                   "isf_sparse_conv_backward_filter")
    if a.f16x3:
        xs = sp.to_split(x)
        g.mul_(1e-6)                       # the size a real backward carries: exercises the power-of-two scale

        def timed(fn, n=10):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(n):
                fn()
            t1.record()
            torch.cuda.synchronize()
            return t0.elapsed_time(t1) / n

        def build_pairs():
            rb._pairs = None
            sp.pair_lists(rb)
        ms_pairs = timed(build_pairs)
        ms_split = timed(lambda: sp.grad_to_split(g))
        gs, sc = sp.grad_to_split(g)
        out = {}

        def call():
            out["dw"] = sp.sparse_conv_backward_filter_f16x3(xs, a.cin, gs, a.cout, rb, sc[1:], (27, a.cin, a.cout))
        ms = timed(call)
        dw = out["dw"]
        fl = 2.0 * pairs * a.cin * a.cout
        nb = rb.nbr.view(27, rb.stride)[:, :rb.num_out]
        err = 0.0
        for k in (0, 13, 26):
            m = nb[k] >= 0
            ref = x[nb[k][m].long()].double().T.mm(g[m].double())
            err = max(err, float((dw[k].double() - ref).abs().max() / ref.abs().max()))
        print(f"rows {a.rows} {a.cin}->{a.cout}: {pairs} pairs, f16x3 dW {ms:.3f} ms = {fl / ms / 1e9:.1f} TFLOP/s = "
              f"{fl / ms / 1e9 / 833.3:.3f} of the f16x3 roofline (+ once per layer: gradient split {ms_split:.3f} ms; once "
              f"per rulebook: pair lists {ms_pairs:.3f} ms); rel err vs float64 {err:.1e}")
        return
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * pairs * a.cin * a.cout
    ref = torch.zeros_like(dw)
    nb = rb.nbr.view(27, rb.stride)[:, :rb.num_out]
    for k in (0, 13, 26):                      # spot check three taps against torch
        m = nb[k] >= 0
        ref[k] = x[nb[k][m].long()].double().T.mm(g[m].double()).float()
    err = max(float((dw[k] - ref[k]).abs().max() / ref[k].abs().max()) for k in (0, 13, 26))
    print(f"rows {a.rows} {a.cin}->{a.cout}: {pairs} pairs, {ms:.3f} ms per dW, {fl / ms / 1e9:.1f} TFLOP/s = "
          f"{fl / ms / 1e9 / 157.3:.3f} of the fp32 MFMA peak; rel err vs float64 {err:.1e}")


if __name__ == "__main__":
    main()
