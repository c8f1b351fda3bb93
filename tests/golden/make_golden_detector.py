"""Generate tests/golden/detector_ref.npz: the REFERENCE's ISFusionDetector.extract_pts_feat (models/detectors/
isfusion.py:103-121) executed on the CPU -- reference detector code, reference sub-modules (Voxelization wrapper,
DynamicVFE, SparseEncoder, ISFusionEncoder, SECONDV2, SECONDFPN), compiled ops served by the CPU oracle
(ref_harness.install_detector) -- on the inputs / weights tests/test_gpu_e2e.py uses.  Cross-checks the oracle
composition that the GPU end-to-end test compares against, i.e. pins the glue between the pinned pieces.

    python tests/golden/make_golden_detector.py            # authoring container only
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_harness  # noqa: E402
import isfusion_amd  # noqa: E402,F401
import oracle  # noqa: E402
from detector_common import build_path, detector_inputs, oracle_extract_pts_feat  # noqa: E402
from isfusion_amd import registry  # noqa: E402


def main():
    ref = ref_harness.install_detector()
    cfg = registry.load_config(os.path.join(ref_harness.REF, "configs/isfusion/isfusion_0075voxel.py"))["model"]
    A = ref["AttrDict"]
    det = ref["isfusion"].ISFusionDetector(
        pts_voxel_layer=cfg["pts_voxel_layer"], pts_voxel_encoder=cfg["pts_voxel_encoder"],
        pts_middle_encoder=cfg["pts_middle_encoder"], pts_backbone=cfg["pts_backbone"], pts_neck=cfg["pts_neck"],
        pts_bbox_head=None, fusion_encoder=cfg["fusion_encoder"], detach=cfg["detach"], pc_range=cfg["pc_range"],
        voxel_size=cfg["voxel_size"], out_size_factor=cfg["out_size_factor"], train_cfg=None,
        test_cfg=A(pts=A(cfg["test_cfg"]["pts"]))).eval()
    det.pts_bbox_head = torch.nn.Identity()     # extract_pts_feat returns None without a head (with_pts_bbox, :105)
    net = build_path()
    sd = {k: v for k, v in net.state_dict().items() if not k.startswith("pts_bbox_head.")}
    res = det.load_state_dict(sd, strict=True)          # the reference detector takes this build's keys as they are
    print("state dict into the reference detector:", res, len(sd), "tensors")
    pts, inp, kw, metas = detector_inputs()
    with torch.no_grad():
        x = det.extract_pts_feat([torch.from_numpy(p) for p in pts], tuple(torch.from_numpy(a) for a in inp["img_feats"]),
                                 metas, **kw)
        out = x[0]
        f0, f1, hm, top = oracle_extract_pts_feat(net, pts, inp, kw)
        neck = net.pts_neck([f0, f1])[0]
    err = (out - neck).abs().max().item()
    print("reference extract_pts_feat", tuple(out.shape), "vs oracle composition + neck:", err, "max", out.abs().max().item())
    assert out.shape == neck.shape and err < 1e-3
    g = np.random.default_rng(2)
    flat = out.numpy().reshape(-1)
    pick = g.integers(0, flat.size, 40000)
    store = {"idx": pick, "val": flat[pick], "shape": np.array(out.shape),
             "mean_abs": np.array([np.abs(flat).mean()])}
    path = os.path.join(HERE, "detector_ref.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
