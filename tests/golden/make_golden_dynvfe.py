"""Generate tests/golden/dynvfe_ref.npz: the REFERENCE's DynamicVFE (models/voxel_encoders/voxel_encoder.py:287-547)
with the isfusion_0075voxel kwargs, running through the reference's Python DynamicScatter wrapper
(ops/voxel/scatter_points.py) over the oracle's restatement of the one compiled op underneath
(ref_harness.install_dynamic_vfe).  Pins cluster / voxel-centre decoration, map_voxel_center_to_point, the two
Linear + BN + ReLU + max layers, output order and key names.

    python tests/golden/make_golden_dynvfe.py            # authoring container only
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_harness  # noqa: E402
import isfusion_amd as m  # noqa: E402
import oracle  # noqa: E402
from isfusion_amd import synthetic  # noqa: E402
from isfusion_amd.norm import fold_bn  # noqa: E402

VS, RG = [0.075, 0.075, 0.2], [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]
CASES = {"b2": (41, 2, 3000), "b1": (43, 1, 5000)}          # name -> (seed, batch, points per sample)


def case_inputs(seed, B, P):
    pl = []
    for i in range(B):
        p = synthetic.lidar_sweeps(seed + i, P)
        c = oracle.dynamic_voxelize(p, VS, RG)
        pl.append(p[(c >= 0).all(1)])                          # the pipeline's PointsRangeFilter leaves in-range points
    coors = np.concatenate([np.concatenate([np.full((p.shape[0], 1), b, np.int32), oracle.dynamic_voxelize(p, VS, RG)], 1)
                            for b, p in enumerate(pl)])
    return np.concatenate(pl), coors


def main():
    oracle.build()
    ref = ref_harness.install_dynamic_vfe()["voxel_encoder"]
    store = {}
    for name, (seed, B, P) in CASES.items():
        lb = m.LidarBranch().randomize_weights_(seed).randomize_bn_(seed + 1).eval()
        mine = lb.pts_voxel_encoder
        cfg = dict(m.ISFUSION_0075["pts_voxel_encoder"], voxel_size=VS, point_cloud_range=RG)
        r = ref.DynamicVFE(**cfg).eval()
        print(name, r.load_state_dict(mine.state_dict(), strict=True))
        pts, coors = case_inputs(seed, B, P)
        with torch.no_grad():
            vf, vc = r(torch.from_numpy(pts), torch.from_numpy(coors))
        bn1 = [t.numpy() for t in fold_bn(mine.vfe_layers[0].norm)]
        bn2 = [t.numpy() for t in fold_bn(mine.vfe_layers[1].norm)]
        ovf, ovc, _ = oracle.dynamic_vfe(pts, coors, VS, RG, mine.vfe_layers[0].linear.weight.detach().numpy(), bn1,
                                         mine.vfe_layers[1].linear.weight.detach().numpy(), bn2)
        assert np.array_equal(vc.numpy(), ovc), "voxel order / coordinates differ"
        err = np.abs(vf.numpy() - ovf).max()
        print(name, tuple(vf.shape), "restatement vs reference DynamicVFE:", err, "max", np.abs(ovf).max())
        assert err < 1e-4
        store[name + ".voxel_feats_every4"] = vf.numpy()[::4]      # every 4th voxel row (fixture size)
        store[name + ".feat_sums"] = vf.numpy().astype(np.float64).sum(0)
        store[name + ".voxel_coors"] = vc.numpy().astype(np.int32)
    path = os.path.join(HERE, "dynvfe_ref.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
