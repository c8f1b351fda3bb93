"""Which torch ops (layout copies, elementwise glue, stock convs) run between the HIP kernels of the cfg-3 path, and from
which line of this package?  torch.profiler over a few forwards of ISFusionPtsPath.forward_pts (B = 2), grouped by the
python call site.   gpurun --timeout 600 -- 'python tools/glue_profile.py > gpurun_out/glue_profile.txt 2>&1'"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from isfusion_amd import synthetic  # noqa: E402
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
    net.freeze()              # inference deployment: weights are static, the caches skip their change scans
    B = 2
    pts = [torch.from_numpy(p).to(dev) for p in bench.make_frames(0, 1, B, 300000, 10)]
    inp = synthetic.fusion_inputs(5, B)
    img_feats = tuple(torch.from_numpy(x).to(dev) for x in inp["img_feats"])
    kw = dict(lidar2img=torch.from_numpy(inp["lidar2img"]), img_aug_matrix=torch.from_numpy(inp["img_aug_matrix"]),
              lidar_aug_matrix=torch.from_numpy(inp["lidar_aug_matrix"]))
    metas = [dict(input_shape=inp["input_shape"]) for _ in range(B)]
    for _ in range(5):
        net.forward_pts(pts, img_feats, metas, **kw)
    torch.cuda.synchronize()
    n = 5
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
                 experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:   # verbose: python frames
        for _ in range(n):
            net.forward_pts(pts, img_feats, metas, **kw)
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=45,
                                                             max_name_column_width=60, max_shapes_column_width=70))
    # aten ops with device time, attributed to the innermost frame inside this package
    sites = {}
    for ev in prof.events():
        t = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
        if t <= 0 or not ev.name.startswith("aten::"):
            continue
        site = "?"
        for fr in ev.stack or []:
            if "is-fusion_amd/" in fr:
                site = fr.split("is-fusion_amd/")[-1]
                break
        key = (site, ev.name)
        a = sites.setdefault(key, [0.0, 0])
        a[0] += t
        a[1] += 1
    print("\n# device time of aten ops by call site (us per forward, calls per forward)")
    for (site, name), (t, c) in sorted(sites.items(), key=lambda kv: -kv[1][0])[:60]:
        print("%9.1f us  %5.1f x  %-28s %s" % (t / n, c / n, name, site))
    print("total aten device time per forward: %.1f us" % (sum(v[0] for v in sites.values()) / n))


if __name__ == "__main__":
    main()
