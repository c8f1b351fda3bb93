"""Run the LiDAR branch on seeded frames and print a digest of the BEV features -- used to check that an opt-in conv
kernel variant (ISF_CONV16_TEPI / ISF_CONV16_PRIO / ISF_CONV16_NW / ISF_CONV16_RG, read once when libisf_hip.so is
loaded, hence one process per variant) reproduces the default kernels' output.

    ISF_CONV16_TEPI=1 python tools/conv_variant_check.py [points_per_frame] [out.npy]
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import isfusion_amd as m  # noqa: E402
from isfusion_amd import synthetic  # noqa: E402


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    dev = torch.device("cuda:0")
    lb = m.LidarBranch().randomize_weights_(0).randomize_bn_(1).eval().to(dev)
    pts = [torch.from_numpy(synthetic.lidar_sweeps(555 + i, P)).to(dev) for i in range(2)]
    out = lb(pts).cpu().numpy()
    if len(sys.argv) > 2:
        np.save(sys.argv[2], out)
    print(hashlib.sha256(out.tobytes()).hexdigest(), float(np.abs(out).max()), bool(np.isfinite(out).all()))


if __name__ == "__main__":
    main()
