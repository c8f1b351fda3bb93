"""Which stock torch ops (layout copies, casts, elementwise glue) run in a training step of the point-cloud path, from which
line of this package (forward ops; backward ops run on the autograd thread and are listed by name + shape)?
torch.profiler over two steps at the configs[3] size.
gpurun --timeout 900 -- 'python tools/train_glue_profile.py > gpurun_out/train_glue.txt 2>&1'"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isfusion_amd import synthetic  # noqa: E402
from isfusion_amd.detector import ISFusionPtsPath  # noqa: E402
from isfusion_amd.fusion_modules import seeded_state_dict  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    B, points = 2, 300000
    net = ISFusionPtsPath().train()
    net._lidar.randomize_weights_(0).randomize_bn_(1)
    for mod, seed in ((net.fusion_encoder, 100), (net.pts_backbone, 200), (net.pts_neck, 250)):
        mod.load_state_dict(seeded_state_dict(mod, seed))
    for p in net.pts_bbox_head.parameters():
        p.requires_grad_(False)
    net = net.to(dev)
    opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=1e-4, momentum=0.9)
    pts = [torch.from_numpy(synthetic.lidar_sweeps(9000 + i, points)).to(dev) for i in range(B)]
    inp = synthetic.fusion_inputs(7, B)
    img = tuple(torch.from_numpy(x).to(dev) for x in inp["img_feats"])
    kw = dict(lidar2img=torch.from_numpy(inp["lidar2img"]), img_aug_matrix=torch.from_numpy(inp["img_aug_matrix"]),
              lidar_aug_matrix=torch.from_numpy(inp["lidar_aug_matrix"]))
    metas = [dict(input_shape=inp["input_shape"]) for _ in range(B)]

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out, hm = net.forward_train_pts(pts, img, metas, **kw)
            loss = (out[0].float() ** 2).mean() + hm.float().sigmoid().mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    n = 2
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
                 experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        for _ in range(n):
            step()
        torch.cuda.synchronize()
    sites = {}
    for ev in prof.events():
        t = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
        if t <= 0:
            continue
        site = "(autograd / other thread)"
        for fr in ev.stack or []:
            if "is-fusion_amd/" in fr or "tools/train_glue" in fr:
                site = fr.split("/")[-1]
                break
        shapes = str(getattr(ev, "input_shapes", ""))[:70] if site.startswith("(") else ""
        a = sites.setdefault((site, ev.name, shapes), [0.0, 0])
        a[0] += t
        a[1] += 1
    print("# device time of ops by call site (us per step, calls per step)")
    tot = 0.0
    for (site, name, shapes), (t, c) in sorted(sites.items(), key=lambda kv: -kv[1][0])[:110]:
        print(f"{t / n:9.1f} {c / n:7.1f}  {name[:44]:44s} {site[:60]} {shapes}")
    for (site, name, shapes), (t, c) in sites.items():
        if name.startswith("aten::"):
            tot += t
    print(f"# aten ops total: {tot / n / 1000:.2f} ms per step")


if __name__ == "__main__":
    main()
