"""tools/conv_trace.py -- where the time of ONE deep sparse-conv launch goes, per workgroup (isf_sparse_conv_trace).

    python tools/conv_trace.py [--level 3] [--batch 4] [--points 300000] [--reps 5]
    python tools/conv_trace.py --level 0 | 1      # the NARROW layers' LDS-DMA kernel (isf_sparse_conv_dma_trace, round 6)

Builds the benchmark geometry (B synthetic sweeps -> voxels -> the three strided levels), runs the production launch of
the level's SubM layer (level 3: 256 -> 256, level 2: 128 -> 128) with the per-workgroup trace on, with tiles in
launch order and in isf_sparse_conv_tile_order order, and prints: the span of the launch, the dispatch skew, the
duration of a workgroup against its step count (latency per step), prologue / epilogue shares, and how evenly the CUs
finish -- the measurements behind DESIGN.md section 5.1."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TICK_US = 0.01   # wall_clock64: 100 MHz


def level_rulebooks(points_list, B, level):
    import torch
    from isfusion_amd import spconv, voxelize
    voxel, rng = (0.075, 0.075, 0.2), (-54.0, -54.0, -5.0, 54.0, 54.0, 3.0)
    _, coors = voxelize.dynamic_voxelize_batched(points_list, voxel, rng)
    coors = coors[(coors >= 0).all(1)]
    idx = torch.unique(coors, dim=0).int().contiguous()   # (b, z, y, x), lexicographically sorted
    shape = [41, 1440, 1440]
    pads = [(1, 1, 1), (1, 1, 1), (0, 1, 1)]
    for lvl in range(level):
        rb = spconv.build_rulebook(idx, B, shape, (3, 3, 3), (2, 2, 2), pads[lvl], False)
        idx, shape = rb.out_indices, rb.out_shape
    return spconv.build_rulebook(idx, B, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), True)


def describe(name, tr):
    import numpy as np
    tr = tr[tr[:, 3] != 0]   # workgroups without a tile write nothing
    t0, t1, t2, t3, steps, hw, xcc = (tr[:, i].astype(np.float64) for i in range(7))
    half = (tr[:, 7] >> 32) != 0
    base = t0.min()
    span = (t3.max() - base) * TICK_US
    dur = (t3 - t0) * TICK_US
    loop = (t2 - t1) * TICK_US
    print(f"== {name}: {len(tr)} workgroups ({int(half.sum())} half tiles), launch span {span:.1f} us")
    print(f"   entry skew: last workgroup starts {((t0.max() - base) * TICK_US):.1f} us after the first; "
          f"p50 {np.percentile((t0 - base) * TICK_US, 50):.1f} p99 {np.percentile((t0 - base) * TICK_US, 99):.1f}")
    print(f"   workgroup duration us: mean {dur.mean():.1f} p10 {np.percentile(dur, 10):.1f} p50 {np.percentile(dur, 50):.1f} "
          f"p90 {np.percentile(dur, 90):.1f} max {dur.max():.1f}")
    print(f"   prologue {((t1 - t0) * TICK_US).mean():.1f} us, multiply loop {loop.mean():.1f} us, epilogue "
          f"{((t3 - t2) * TICK_US).mean():.1f} us (means)")
    ok = steps > 0
    per = loop[ok] / steps[ok]
    print(f"   steps: mean {steps.mean():.0f} min {steps.min():.0f} max {steps.max():.0f}; loop time per step us: "
          f"mean {per.mean():.3f} p10 {np.percentile(per, 10):.3f} p90 {np.percentile(per, 90):.3f}")
    A = np.stack([steps[ok], np.ones(ok.sum())], 1)
    coef = np.linalg.lstsq(A, loop[ok], rcond=None)[0]
    print(f"   fit loop_us = {coef[0]:.3f} * steps + {coef[1]:.1f}   (corr {np.corrcoef(steps[ok], loop[ok])[0, 1]:.3f})")
    for hsel, label in ((~half, "full"), (half, "half")):
        if hsel.sum() > 4:
            sel = hsel & ok
            print(f"   {label} tiles: loop per step {np.mean(loop[sel] / steps[sel]):.3f} us, duration {dur[hsel].mean():.1f} us")
    # CU identity: (XCC_ID, SE, SH, CU) from HW_ID (gfx9 layout: cu_id [11:8], sh_id [12], se_id [15:13])
    hwi = hw.astype(np.int64)
    cu = ((xcc.astype(np.int64) & 0xf) << 8) | (hwi >> 8 & 0xff)
    ids = np.unique(cu)
    fin = np.array([(t3[cu == c].max() - base) * TICK_US for c in ids])
    cnt = np.array([(cu == c).sum() for c in ids])
    busy = np.array([((t3[cu == c]).max() - (t0[cu == c]).min()) * TICK_US for c in ids])
    work = np.array([steps[cu == c].sum() for c in ids])
    print(f"   {len(ids)} CUs seen; workgroups per CU min {cnt.min()} max {cnt.max()}; CU finish time us: "
          f"mean {fin.mean():.1f} p10 {np.percentile(fin, 10):.1f} p50 {np.percentile(fin, 50):.1f} p90 "
          f"{np.percentile(fin, 90):.1f} max {fin.max():.1f}")
    print(f"   steps per CU: mean {work.mean():.0f} max {work.max():.0f}; corr(CU finish, CU steps) "
          f"{np.corrcoef(fin, work)[0, 1]:.3f}; busy span mean {busy.mean():.1f}")
    return span


def main_narrow(args, rb, dev):
    """levels 0 / 1: 32 -> 32 / 64 -> 64 on the LDS-DMA kernel; the per-workgroup record carries wave 0's cycle account"""
    import numpy as np
    import torch
    from isfusion_amd import spconv
    C = 32 if args.level == 0 else 64
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(rb.num_in, C, generator=g).to(dev)
    w = (torch.randn(3, 3, 3, C, C, generator=g) * (1.0 / (9 * C)) ** 0.5).to(dev)
    packed = spconv.pack_filters_f16x3(w)
    xs = spconv.to_split(x)
    scale, shift = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    res = spconv.to_split(torch.randn(rb.num_out, C, generator=g).to(dev))
    print(f"level {args.level}: {rb.num_out} rows, {C} -> {C}, {'dense table' if args.dense_table else 'line table'}")
    spans = []
    for r in range(args.reps + 2):
        ys, tr = spconv.sparse_conv_dma_trace(xs, packed, 27, C, C, rb, scale, shift, res, True, lines=not args.dense_table)
        torch.cuda.synchronize()
        if r >= 2:
            spans.append((tr[:, 3].max() - tr[tr[:, 3] != 0][:, 0].min()).item() * TICK_US)
    print(f"-- launch span over {args.reps} runs: " + " ".join(f"{s:.1f}" for s in spans) + " us")
    t = tr.cpu().numpy()
    describe("narrow launch", t[:, :8])
    t = t[t[:, 3] != 0]
    cyc = t[:, 8:12].astype(np.float64)
    steps = np.maximum(t[:, 4].astype(np.float64), 1)
    tot = cyc.sum(1)
    names = ("vmcnt(0) wait", "barrier", "read + issue", "multiply")
    print("   wave 0 of every workgroup, shader-clock cycles of the loop: " +
          ", ".join(f"{n} {100 * cyc[:, i].sum() / tot.sum():.1f} %" for i, n in enumerate(names)))
    print("   cycles per step (mean over workgroups): " +
          ", ".join(f"{n} {np.mean(cyc[:, i] / steps):.0f}" for i, n in enumerate(names)) +
          f"; total {np.mean(tot / steps):.0f}")
    rd, adv = t[:, 12].astype(np.float64), t[:, 13].astype(np.float64)
    print(f"   of read + issue, cycles per step: fragment reads (LDS round trip) {np.mean(rd / steps):.0f}, index arithmetic + "
          f"weight run {np.mean(adv / steps):.0f}, row gathers {np.mean((cyc[:, 2] - rd - adv) / steps):.0f}")
    loop_us = (t[:, 2] - t[:, 1]) * TICK_US
    print(f"   loop wall time per step {np.mean(loop_us / steps):.3f} us = {np.mean(loop_us / steps) * 1e3:.0f} ns; "
          f"cycles / wall => shader clock ~ {np.mean(tot / np.maximum(loop_us, 1e-9)) / 1e3:.2f} GHz")
    if args.dump:
        np.savez(args.dump, narrow=t)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=3, choices=[0, 1, 2, 3])
    ap.add_argument("--dense-table", action="store_true", help="levels 0 / 1: the dense neighbour table instead of lines")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--points", type=int, default=300000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--dump", default="", help="save the raw traces (npz)")
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from isfusion_amd import spconv
    dev = torch.device("cuda", 0)
    pts = [torch.from_numpy(p).to(dev) for p in bench.make_frames(0, 1, args.batch, args.points, 0)]
    rb = level_rulebooks(pts, args.batch, args.level)
    if args.level < 2:
        return main_narrow(args, rb, dev)
    C = 256 if args.level == 3 else 128
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(rb.num_in, C, generator=g).to(dev)
    w = (torch.randn(3, 3, 3, C, C, generator=g) * (1.0 / (9 * C)) ** 0.5).to(dev)
    packed = spconv.pack_filters_f16x3(w)
    xs = spconv.to_split(x)
    scale = torch.ones(C, device=dev)
    shift = torch.zeros(C, device=dev)
    order = spconv.tile_order(rb, C, C)
    print(f"level {args.level}: {rb.num_out} rows, {C} -> {C}; tile order {'built' if order is not None else 'not applicable'}")
    if order is not None:
        work = rb._tile_order[(C, C, 0)][1].cpu().numpy()
        print(f"   tile work (row group, tap) pairs: mean {work.mean():.0f} min {work.min()} max {work.max()}")
    out = {}
    for name, o in (("launch order", None), ("tile order", order)):
        if name == "tile order" and order is None:
            continue
        spans = []
        for r in range(args.reps + 2):
            ys, tr = spconv.sparse_conv_trace(xs, packed, 27, C, C, rb, scale, shift, None, True, order=o)
            torch.cuda.synchronize()
            if r >= 2:
                spans.append((tr[:, 3].max() - tr[tr[:, 3] != 0][:, 0].min()).item() * TICK_US)
        print(f"-- {name}: launch span over {args.reps} runs: " + " ".join(f"{s:.1f}" for s in spans) + " us")
        out[name] = tr.cpu().numpy()
        describe(name, out[name])
    if args.dump:
        np.savez(args.dump, **{k.replace(" ", "_"): v for k, v in out.items()})


if __name__ == "__main__":
    main()
