"""Seeded cases for the SparseEncoder wiring goldens (tests/golden/make_golden_encoder.py and the tests)."""
import numpy as np

from isfusion_amd import ISFUSION_0075

ENCODER_CASES = {
    # the IS-Fusion configuration (configs/isfusion/isfusion_0075voxel.py:75-86: basicblock stages, paddings incl. [0,1,1])
    "isfusion": dict(cfg=dict(ISFUSION_0075["pts_middle_encoder"]), seed=21, B=2, voxels=2500, rng=5),
    # mmdet3d's default SECOND layout (conv_module blocks) on a smaller grid
    "conv_module": dict(cfg=dict(in_channels=16, sparse_shape=[41, 160, 160], output_channels=64,
                                 encoder_channels=((16,), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
                                 encoder_paddings=((1,), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)),
                                 block_type="conv_module"), seed=31, B=1, voxels=1500, rng=6),
}


def encoder_input(case):
    """-> voxel features float32 [N, Cin], coords int32 [N, 4] sorted (b, z, y, x), batch size: clustered active sites
    (neighbours exist) inside the case's sparse shape"""
    cfg = case["cfg"]
    D, H, W = cfg["sparse_shape"]
    g = np.random.default_rng(case["rng"])
    cells = set()
    while len(cells) < case["voxels"]:
        b = int(g.integers(0, case["B"]))
        z0, y0, x0 = int(g.integers(2, D - 3)), int(g.integers(4, H - 8)), int(g.integers(4, W - 8))
        for _ in range(40):                                     # a small blob around the seed cell
            z, y, x = z0 + int(g.integers(-2, 3)), y0 + int(g.integers(-4, 5)), x0 + int(g.integers(-4, 5))
            cells.add((b, z, y, x))
    coors = np.array(sorted(cells), np.int32)
    feats = g.normal(size=(coors.shape[0], cfg["in_channels"])).astype(np.float32)
    return feats, coors, case["B"]
