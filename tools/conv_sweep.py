"""Per-layer sweep of the sparse-conv ring kernel (isf_spconv_ring.hip) on the REAL geometry of the bench workload:
the 21 conv layers of the isfusion_0075voxel SparseEncoder on B x P-point synthetic frames (voxel sets and rulebooks
from the library itself), each layer timed in isolation with HIP events for

    ref                       the one-step-prefetch kernel (isf_tune_conv_ring(0, ...)), the validated baseline
    ring (NW, RG, PA) ...     every built workgroup shape / prefetch distance (isf_tune_conv_ring(1, NW, RG, PA))

and every ring result compared BIT FOR BIT with ref (same products, same summation order).  Prints one table per
distinct layer shape and a JSON summary (best configuration per shape).

    python tools/conv_sweep.py [--batch 4] [--points 300000] [--reps 20] [--json out.json]
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CONFIGS = [(4, 2, 1), (4, 2, 2), (4, 2, 3), (8, 2, 1), (8, 2, 2), (8, 2, 3), (4, 3, 2), (8, 3, 2)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--points", type=int, default=300000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    import isfusion_amd as m
    from isfusion_amd import _lib, spconv, synthetic
    from isfusion_amd.voxelize import dynamic_voxelize_batched
    lib = _lib.load()
    dev = torch.device("cuda:0")
    lb = m.LidarBranch().randomize_weights_(0).randomize_bn_(1).eval().to(dev)
    frames = [torch.from_numpy(synthetic.lidar_sweeps(1234 + 2000 + i, args.points)).to(dev) for i in range(args.batch)]
    # active level-0 voxels = distinct in-range coordinates, (b, z, y, x)-sorted like the VFE emits them
    _, coors = dynamic_voxelize_batched(frames, lb.voxel_size, lb.point_cloud_range)
    coors = torch.unique(coors[(coors[:, 1:] >= 0).all(1)], dim=0).int().contiguous()
    plan = lb.pts_middle_encoder.export_plan()
    shape = list(plan["sparse_shape"])
    g = torch.Generator(device="cpu").manual_seed(0)
    results, seen = [], {}
    idx = coors
    sub_rb = None
    for li, L in enumerate(plan["layers"]):
        subm = L["kind"] == "subm"
        if subm:
            if sub_rb is None:
                sub_rb = spconv.build_rulebook(idx, args.batch, shape, L["ksize"], L["stride"], L["padding"], True)
            rb = sub_rb
        else:
            rb = spconv.build_rulebook(idx, args.batch, shape, L["ksize"], L["stride"], L["padding"], False)
        cin, cout, K = L["c_in"], L["c_out"], int(np.prod(L["ksize"]))
        key = (cin, cout, K, rb.num_in, rb.num_out)
        if key not in seen:
            seen[key] = True
            pairs = int((rb.nbr[:, :rb.num_out] >= 0).sum().item())
            x = torch.randn((rb.num_in, cin), generator=g).to(dev)
            w = (torch.randn((K, cin, cout), generator=g) / np.sqrt(K * cin)).to(dev)
            scale = (1 + 0.1 * torch.randn(cout, generator=g)).to(dev)
            shift = (0.1 * torch.randn(cout, generator=g)).to(dev)
            res = torch.randn((rb.num_out, cout), generator=g).to(dev) if subm and cin == cout else None
            xs, rs = spconv.to_split(x), (spconv.to_split(res) if res is not None else None)
            packed = spconv.pack_filters_f16x3(w.view(*L["ksize"], cin, cout))
            ys = torch.empty(rb.num_out * cout * 4, dtype=torch.uint8, device=dev)
            gm = rb.group_masks()

            def run():
                _lib.check(lib.isf_sparse_conv_forward_f16x3(
                    _lib.ptr(xs), rb.num_in, cin, _lib.ptr(packed), K, cout, _lib.ptr(rb.nbr), rb.stride, rb.num_out,
                    _lib.ptr(gm), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(rs) if rs is not None else None, 1, _lib.ptr(ys),
                    _lib.stream()), "conv")

            def timed():
                run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    run()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / args.reps * 1e3   # us

            _lib.check(lib.isf_tune_conv_ring(0, 0, 0, 0))
            t_ref = timed()
            ref = ys.clone()
            flops = 2.0 * pairs * cin * cout
            row = dict(layer=li, cin=cin, cout=cout, K=K, n_in=rb.num_in, n_out=rb.num_out, pairs=pairs,
                       ref_us=round(t_ref, 1), ideal_mfma_us=round(3 * flops / 2.5e15 * 1e6, 1), ring={})
            print(f"layer {li:2d} {cin:3d}->{cout:3d} K={K:2d} n_out={rb.num_out:7d} pairs/row={pairs / max(rb.num_out, 1):5.1f} "
                  f"ideal {row['ideal_mfma_us']:6.1f} us | ref {t_ref:7.1f} us", flush=True)
            for nw, rgc, pa in CONFIGS:
                kch = cin // 32 if (cin <= 64 and cout <= 64) else 1
                nt = min(cout // 16, 8)
                if (kch * nt * 2) % nw:
                    continue
                _lib.check(lib.isf_tune_conv_ring(1, nw, rgc, pa))
                ys.zero_()
                try:
                    t = timed()
                except _lib.IsfError as e:
                    print("      ", (nw, rgc, pa), "not built:", str(e)[:60])
                    continue
                same = bool(torch.equal(ys, ref))
                row["ring"][f"{nw},{rgc},{pa}"] = dict(us=round(t, 1), bit_equal=same)
                print(f"        ring NW={nw} RG={rgc} PA={pa}: {t:7.1f} us  {'==' if same else 'MISMATCH'}", flush=True)
            _lib.check(lib.isf_tune_conv_ring(0, 0, 0, 0))
            results.append(row)
        if not subm:
            idx, shape, sub_rb = rb.out_indices.contiguous(), rb.out_shape, None
    bad = [(r["layer"], k) for r in results for k, v in r["ring"].items() if not v["bit_equal"]]
    summary = dict(batch=args.batch, points=args.points, layers=results, mismatches=bad)
    print("MISMATCHES:", bad if bad else "none")
    for r in results:
        if r["ring"]:
            k, v = min(r["ring"].items(), key=lambda kv: kv[1]["us"])
            print(f"  {r['cin']:3d}->{r['cout']:3d} K={r['K']:2d} n={r['n_out']:7d}: ref {r['ref_us']:7.1f}  best ring ({k}) {v['us']:7.1f} us"
                  f"  x{r['ref_us'] / v['us']:.2f}   ideal {r['ideal_mfma_us']}")
    if args.json:
        json.dump(summary, open(args.json, "w"), indent=1)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
