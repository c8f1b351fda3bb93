"""Census of the fused-Linear launches (isf_linear_forward) of one full forward (BASELINE configs[2], B = 2): shape, epilogue
flags and stand-alone time of every distinct call, sorted by time per forward.
gpurun --timeout 600 -- 'python tools/linear_census.py > gpurun_out/linear_census.txt 2>&1'"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from isfusion_amd import fusion_ops as ops, synthetic  # noqa: E402
from isfusion_amd.detector import ISFusionPtsPath  # noqa: E402
from isfusion_amd.fusion_modules import seeded_state_dict  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    bench.CFG_ID = 3
    net = ISFusionPtsPath().eval()
    net._lidar.randomize_weights_(0).randomize_bn_(1)
    for mod, seed in ((net.fusion_encoder, 100), (net.pts_backbone, 200), (net.pts_neck, 250), (net.pts_bbox_head, 300)):
        mod.load_state_dict(seeded_state_dict(mod, seed))
    net = net.to(dev)
    net.freeze()
    B = 2
    pts = [torch.from_numpy(p).to(dev) for p in bench.make_frames(0, 1, B, 300000, 10)]
    inp = synthetic.fusion_inputs(5, B)
    img_feats = tuple(torch.from_numpy(x).to(dev) for x in inp["img_feats"])
    kw = dict(lidar2img=torch.from_numpy(inp["lidar2img"]), img_aug_matrix=torch.from_numpy(inp["img_aug_matrix"]),
              lidar_aug_matrix=torch.from_numpy(inp["lidar_aug_matrix"]))
    metas = [dict(input_shape=inp["input_shape"]) for _ in range(B)]
    for _ in range(3):
        net.forward_pts(pts, img_feats, metas, **kw)
    torch.cuda.synchronize()
    calls = []
    real = ops.linear

    def spy(x, pl, **k):
        calls.append((x, pl, dict(k)))
        return real(x, pl, **k)

    ops.linear = spy
    net.forward_pts(pts, img_feats, metas, **kw)
    torch.cuda.synchronize()
    ops.linear = real
    rows = {}
    for x, pl, k in calls:
        M = x.size(0) * x.size(2) * x.size(3) if x.dim() == 4 else x.size(0)
        res = k.get("residual")
        sig = (M, pl.in_features, pl.out_features, "x-nchw" if x.dim() == 4 else "", "table" if k.get("table") is not None else "",
               {0: "", 1: "relu", 2: "gelu"}[k.get("act", 0)],
               "" if res is None else ("res-nchw" if res.dim() == 4 else "res"), "ln" if k.get("ln") is not None else "",
               "y-nchw" if k.get("out_nchw") is not None else "")
        if sig not in rows:
            # 20 launches captured in a HIP graph: device time without the host's launch pace (a 400-row launch takes
            # the host longer to issue than the GPU to run)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    real(x, pl, **k)
                side.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    for _ in range(20):
                        real(x, pl, **k)
                g.replay()
                e0.record(side)
                g.replay()
                e1.record(side)
                side.synchronize()
            rows[sig] = [0, e0.elapsed_time(e1) / 20 * 1e3]
        rows[sig][0] += 1
    tot = 0.0
    print("calls   us each  us/fwd   GB/s   TF/s(x3)   M      K    N   flags")
    for sig, (c, us) in sorted(rows.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        M, K, N = sig[:3]
        gb = (M * K + M * N * (2 if sig[6] else 1)) * 4 / 1e9
        tf = 3 * 2.0 * M * K * N / 1e12
        tot += c * us
        print(f"{c:5d} {us:9.1f} {c * us:7.1f} {gb / us * 1e6:7.0f} {tf / us * 1e6:8.1f}   {M:6d} {K:4d} {N:4d}   {' '.join(s for s in sig[3:] if s)}")
    print(f"# {len(calls)} launches, {tot:.0f} us per forward (stand-alone times)")


if __name__ == "__main__":
    main()
