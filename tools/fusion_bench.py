"""Time ISFusionEncoder.forward (+ SECONDV2 stages) on one GPU: whole forward and per stage (hipEvents).
    python tools/fusion_bench.py [--batch 4] [--steps 10]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from fusion_common import CONFIGS, build_modules, torch_inputs  # noqa: E402


def timed(fn, steps, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--pillars", type=int, default=15000)
    ap.add_argument("--unfused-window", action="store_true",
                    help="run the three-launch form of the window attention (qkv linear / attention / out-projection)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.unfused_window:
        from isfusion_amd import fusion_ops
        fusion_ops.WINDOW_BLOCK_FUSED = False
    cfg = dict(CONFIGS["full"], B=a.batch, num_pillars=a.pillars, seed=31)
    enc, bb = build_modules(cfg, dev)
    t = torch_inputs(cfg, dev)
    B = cfg["B"]
    kw = dict(pts_metas=dict(pillars=t["pillars"], pillar_coors=t["pillar_coors"]),
              img_metas=[dict(input_shape=t["input_shape"])], pts_backbone=bb, lidar2img=t["lidar2img"],
              img_aug_matrix=t["img_aug_matrix"], lidar_aug_matrix=t["lidar_aug_matrix"])
    res = {"batch": B, "pillars_per_sample": a.pillars}
    with torch.no_grad():
        ms, (feats, hm) = timed(lambda: enc(t["img_feats"], t["lidar_feats"], B, **kw), a.steps)
        res["forward_ms"] = ms
        res["frames_per_s"] = B / ms * 1e3
        ms, img_bev = timed(lambda: enc.img_fv_to_bev([t["img_feats"][1]], B, **kw), a.steps)
        res["p2g_ms"] = ms
        ms, bev = timed(lambda: enc.fuse(img_bev, t["lidar_feats"]), a.steps)
        res["conv_fusion_ms"] = ms
        ms, g0 = timed(lambda: enc.grid2region(0, bev), a.steps)
        res["g2r0_ms"] = ms
        ms, (ret, _) = timed(lambda: enc.instance_fusion(bev, g0, B), a.steps)
        res["instance_fusion_ms"] = ms
        ms, (nxt, _, f0) = timed(lambda: bb([ret], "stage1"), a.steps)
        res["second_stage1_ms"] = ms
        ms, g1 = timed(lambda: enc.grid2region(1, nxt), a.steps)
        res["g2r1_ms"] = ms
        ms, _ = timed(lambda: bb([g1], "stage2"), a.steps)
        res["second_stage2_ms"] = ms
    print(json.dumps(res))


if __name__ == "__main__":
    main()
