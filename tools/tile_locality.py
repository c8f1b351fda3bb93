"""tools/tile_locality.py -- does the ORDER in which a multi-round sparse-conv launch walks its tiles matter (L2 locality)?

    python tools/tile_locality.py [--level 1] [--batch 4] [--points 300000] [--reps 20]

Level-1 geometry of the benchmark (346 k rows, 64 -> 64, the LDS-DMA gather kernel: 5 workgroups per CU = 5 MB of rows
in flight per XCD against a 4 MB L2).  Times the layer with the tiles of every XCD part walked in raster order (the
default: rows are (b, z, y, x)-sorted), in random order, and band by band: all z-planes of one y-band before the next
y-band, so that the z +- 1 neighbours of the tiles in flight are rows of tiles in flight too."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=1, choices=[0, 1])
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--points", type=int, default=300000)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    import conv_trace
    from isfusion_amd import spconv
    dev = torch.device("cuda", 0)
    pts = [torch.from_numpy(p).to(dev) for p in bench.make_frames(0, 1, args.batch, args.points, 0)]
    rb = conv_trace.level_rulebooks(pts, args.batch, args.level)
    C = 64 if args.level == 1 else 32
    n = rb.num_out
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(rb.num_in, C, generator=g).to(dev)
    w = (torch.randn(3, 3, 3, C, C, generator=g) * (1.0 / (9 * C)) ** 0.5).to(dev)
    packed = spconv.pack_filters_f16x3(w)
    xs = spconv.to_split(x)
    ys = torch.empty(n * C * 4, dtype=torch.uint8, device=dev)
    scale, shift = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    # the launch's tile plan (uniform 128-row tiles, 8 XCD parts: conv16_plan for launches of several rounds)
    TM, parts = 128, 8
    groups = -(-(-(-n // 16)) // parts)
    tiles = -(-groups // (TM // 16))
    part_rows = groups * 16
    coords = rb.out_indices.cpu().numpy()          # [n, 4] (b, z, y, x)
    first = np.minimum(np.arange(parts)[:, None] * part_rows + np.arange(tiles)[None, :] * TM, n - 1)
    b0, z0, y0 = coords[first, 0], coords[first, 1], coords[first, 2]
    print(f"level {args.level}: {n} rows, {C} -> {C}, {parts} parts x {tiles} tiles; grid y extent {coords[:, 2].max() + 1}")
    rng = np.random.default_rng(0)
    orders = {"raster (default)": None,
              "identity table": np.tile(np.arange(tiles, dtype=np.int32), (parts, 1)),
              "random": np.stack([rng.permutation(tiles).astype(np.int32) for _ in range(parts)])}
    for band in (16, 32, 64, 128):
        key = (b0.astype(np.int64) * 4096 + y0 // band) * 64 + z0
        orders[f"y-band {band} x all z"] = np.argsort(key, axis=1, kind="stable").astype(np.int32)
    lib = spconv._lib.load()
    for name, o in orders.items():
        ot = None if o is None else torch.from_numpy(np.ascontiguousarray(o)).to(dev)
        def run():
            spconv._lib.check(lib.isf_sparse_conv_forward_dma(
                spconv._lib.ptr(xs), rb.num_in, C, spconv._lib.ptr(packed), 27, C, spconv._lib.ptr(rb.nbr), rb.stride, n,
                spconv._lib.ptr(scale), spconv._lib.ptr(shift), None, 1, spconv._lib.ptr(ys), 0, spconv._lib.ptr(ot),
                spconv._lib.stream()), "isf_sparse_conv_forward_dma")
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        out = ys.clone()
        if name == "raster (default)":
            ref = out
        print(f"   {name:24s} {e0.elapsed_time(e1) / args.reps * 1e3:8.1f} us per launch   same bits: {torch.equal(out, ref)}")


if __name__ == "__main__":
    main()
