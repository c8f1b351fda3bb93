"""tools/stage_sweep.py -- per-layer conv times of the LiDAR branch for a list of LDS-staging settings, one process
(frames and weights built once): which layers gain from isf_sparse_conv_forward_staged and at which LDS share.

    python tools/stage_sweep.py [--rows -1,160,320,448,640,896,1216] [--steps 8] [--batch 4] [--points 300000]

Prints one line per conv layer: cin->cout, rows, then the mean hipEvent time (us) of the layer under every setting
(-1 = gather kernel), and the per-setting totals; ends with the best setting per channel shape."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="-1,160,320,448,640,896,1216")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--points", type=int, default=300000)
    ap.add_argument("--f16", action="store_true")
    args = ap.parse_args()
    import numpy as np
    import torch
    import isfusion_amd as m
    import bench
    dev = torch.device("cuda", 0)
    lb = m.LidarBranch().randomize_weights_(0).randomize_bn_(1).eval().to(dev).freeze()
    sets = [[torch.from_numpy(p).to(dev) for p in bench.make_frames(0, 1, args.batch, args.points, fs)] for fs in range(2)]
    tab = lb.conv_layer_table()
    nl = len(tab)
    settings = [int(v) for v in args.rows.split(",")]
    prec = 2 if args.f16 else 0
    res = {}
    for rows in settings:
        for i in range(3):
            lb(sets[i % 2], stage_rows=rows, precision=prec)
        torch.cuda.synchronize()
        acc = np.zeros(nl)
        import time
        t0 = time.perf_counter()
        for s in range(args.steps):
            lb(sets[s % 2], stage_rows=rows, precision=prec)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / args.steps * 1e3
        for s in range(args.steps):
            lb(sets[s % 2], time_layers=True, stage_rows=rows, precision=prec)
            st = lb.last_stats
            acc += np.array([st.ms[i] for i in range(nl)])
        res[rows] = (acc / args.steps * 1e3, wall)
        n_out = [st.num_out[i] for i in range(nl)]
    print("layer  shape      rows     " + " ".join(f"{r:>8d}" for r in settings))
    for i, (kind, cin, cout, K) in enumerate(tab):
        print(f"{i:2d} {kind[:5]:5s} {cin:3d}->{cout:3d} {n_out[i]:8d} " + " ".join(f"{res[r][0][i]:8.1f}" for r in settings))
    print("conv total (us)               " + " ".join(f"{res[r][0].sum():8.1f}" for r in settings))
    print("step wall (ms, untimed steps) " + " ".join(f"{res[r][1]:8.3f}" for r in settings))
    shapes = {}
    for i, (kind, cin, cout, K) in enumerate(tab):
        shapes.setdefault((cin, cout, K), []).append(i)
    print("best setting per shape:")
    for (cin, cout, K), idx in shapes.items():
        tot = {r: sum(res[r][0][i] for i in idx) for r in settings}
        best = min(tot, key=tot.get)
        print(f"  {cin:3d}->{cout:3d} K={K:2d} x{len(idx)}: best rows {best:5d} {tot[best]:8.1f} us vs gather {tot.get(-1, float('nan')):8.1f} us")


if __name__ == "__main__":
    main()
