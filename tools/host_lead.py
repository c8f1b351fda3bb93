"""tools/host_lead.py -- does the host keep ahead of the GPU in the full forward (BASELINE configs[2])?

    python tools/host_lead.py [--steps 12] [--batch 2] [--graph]

Per step: when the host entered / left forward_pts and when the GPU started / finished that step's work (events on the
launch stream, host clock aligned at a device sync).  lead = GPU finish - host exit: how far the GPU still had to go
when the host was done launching; ~0 means the GPU waits for launches (host-bound), idle = the GPU gap between steps."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--points", type=int, default=300000)
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--fold-once", action="store_true", help="fold the camera matrices once (p2g_cam passed in)")
    ap.add_argument("--cfg-id", type=int, default=3, help="bench.CFG_ID: seeds of the synthetic sweeps")
    ap.add_argument("--same-set", action="store_true", help="the same frame set every forward")
    ap.add_argument("--sync-after", default="", help="device sync after these warm-up forwards, e.g. 0,1 (fault bisection)")
    ap.add_argument("--spin", action="store_true", help="hipSetDeviceFlags(hipDeviceScheduleSpin) before the context exists")
    args = ap.parse_args()
    if args.spin:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        print("hipSetDeviceFlags(hipDeviceScheduleSpin) ->", hip.hipSetDeviceFlags(1))
    import torch
    import bench
    from isfusion_amd import synthetic, fusion_ops as ops
    from isfusion_amd.detector import ISFusionPtsPath
    from isfusion_amd.fusion_modules import seeded_state_dict
    bench.CFG_ID = args.cfg_id
    dev = torch.device("cuda", 0)
    net = ISFusionPtsPath().eval()
    net._lidar.randomize_weights_(0).randomize_bn_(1)
    for mod, seed in ((net.fusion_encoder, 100), (net.pts_backbone, 200), (net.pts_neck, 250), (net.pts_bbox_head, 300)):
        mod.load_state_dict(seeded_state_dict(mod, seed))
    net = net.to(dev)
    net.freeze()
    if args.graph:
        net.enable_graph()
    B = args.batch
    sets = []
    for fs in range(2):
        pts = [torch.from_numpy(p).to(dev) for p in bench.make_frames(0, 1, B, args.points, 10 + fs)]
        inp = synthetic.fusion_inputs(5 + fs, B)
        img_feats = tuple(torch.from_numpy(x).to(dev) for x in inp["img_feats"])
        kw = dict(lidar2img=torch.from_numpy(inp["lidar2img"]), img_aug_matrix=torch.from_numpy(inp["img_aug_matrix"]),
                  lidar_aug_matrix=torch.from_numpy(inp["lidar_aug_matrix"]))
        if args.fold_once:
            kw = dict(p2g_cam=ops.p2g_camera_params(kw["lidar2img"], kw["img_aug_matrix"], kw["lidar_aug_matrix"]).to(dev))
        sets.append((pts, img_feats, [dict(input_shape=inp["input_shape"]) for _ in range(B)], kw))
    if args.same_set:
        sets[1] = sets[0]
    sync_after = [int(v) for v in args.sync_after.split(",") if v != ""]
    for i in range(6):
        p, f, m, kw = sets[i % 2]
        net.forward_pts(p, f, m, **kw)
        if i in sync_after:
            torch.cuda.synchronize()
            print("warm-up forward", i, "done", flush=True)
    torch.cuda.synchronize()
    n = args.steps
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    base = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    base.record()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = []
    for i in range(n):
        p, f, m, kw = sets[i % 2]
        h0 = time.perf_counter()
        ev[i][0].record()
        net.forward_pts(p, f, m, **kw)
        ev[i][1].record()
        host.append((h0 - t0, time.perf_counter() - t0))
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"{n} steps, {wall / n * 1e3:.3f} ms per step, {B * n / wall:.1f} frames/s  (graph {args.graph}, fold once {args.fold_once})")
    print(" step   host enter   host exit    gpu start   gpu finish   lead(gpu finish - host exit)  gpu idle before start   [ms]")
    prev_fin = None
    for i in range(n):
        gs, gf = base.elapsed_time(ev[i][0]), base.elapsed_time(ev[i][1])
        idle = gs - prev_fin if prev_fin is not None else 0.0
        print(f"{i:5d} {host[i][0] * 1e3:12.3f} {host[i][1] * 1e3:11.3f} {gs:12.3f} {gf:12.3f} {gf - host[i][1] * 1e3:14.3f} {idle:28.3f}")
        prev_fin = gf


if __name__ == "__main__":
    main()
