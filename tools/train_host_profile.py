"""Where does the HOST time of a steady-state training step go?  cProfile over five steps of the point-cloud path at the
configs[3] size (no DDP wrapper), after three warm-up steps; the step is paced by the host, so this is the
profile that matters.   gpurun --timeout 900 -- 'python tools/train_host_profile.py > gpurun_out/train_host.txt 2>&1'"""
import cProfile
import os
import pstats
import sys
import time

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
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t_host = (time.perf_counter() - t0) / n          # host time to ENQUEUE a step (no sync inside)
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    print(f"# un-profiled: host enqueue {t_host * 1e3:.1f} ms per step, wall {t_all * 1e3:.1f} ms per step")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(40)
    st.sort_stats("cumulative").print_stats(60)


if __name__ == "__main__":
    main()
