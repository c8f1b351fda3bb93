"""Point-to-Grid sampling on seeded inputs -> digest of the BEV canvas and the kernel time; used to check the
ISF_P2G_PIPE variant (switch read once per process) against the default kernel.

    ISF_P2G_PIPE=1 python tools/p2g_variant_check.py [pillars_per_sample]
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from isfusion_amd import fusion_ops as ops, synthetic  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
    B = 2
    dev = torch.device("cuda:0")
    inp = synthetic.fusion_inputs(21, B, num_pillars=n)
    args = (torch.from_numpy(inp["pillars"][..., :3]).to(dev), torch.from_numpy(inp["pillar_coors"]).to(dev),
            torch.from_numpy(inp["img_feats"][1]).to(dev), torch.from_numpy(inp["lidar2img"]),
            torch.from_numpy(inp["img_aug_matrix"]), torch.from_numpy(inp["lidar_aug_matrix"]), inp["input_shape"], B, 180)
    out = ops.p2g_sample(*args)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10):
        ops.p2g_sample(*args)
    ev[1].record()
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    print(hashlib.sha256(o.tobytes()).hexdigest(), float(np.abs(o).max()), round(ev[0].elapsed_time(ev[1]) / 10, 4))


if __name__ == "__main__":
    main()
