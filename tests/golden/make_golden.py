"""Generate the committed golden fixtures under tests/golden/ (run in the authoring container only).

Sources of truth, in order of strength:
  1. the REFERENCE's own voxelization C++ compiled in place (oracle/build_ref.py -> oracle/_ref) for
     dynamic_voxelize / hard_voxelize                                   -> voxelize_ref.npz
  2. the brute-force reference the reference's own test defines for DynamicScatter
     (tests/test_models/test_voxel_encoder/test_dynamic_scatter.py:57-66: unique(dim=0, sorted) +
     masked mean / max), evaluated with torch on CPU                    -> scatter_ref.npz
  3. the dense-conv3d identity for sparse convolution (torch.nn.functional.conv3d on the zero-filled
     dense grid, sampled at the active output sites; weight [kD,kH,kW,Cin,Cout] -> conv3d layout
     [Cout,Cin,kD,kH,kW], cross-correlation)                             -> spconv_dense_ref.npz
Nothing from /root/reference is copied: only inputs and expected outputs are stored.

Usage:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def boundary_points(rng, n, pc_range, vs):
    lo = np.array(pc_range[:3], np.float32)
    hi = np.array(pc_range[3:], np.float32)
    span = hi - lo
    pts = rng.random((n, 5), dtype=np.float32)
    pts[:, :3] = lo - 0.05 * span + pts[:, :3] * span * 1.1  # ~9 % of the rows fall outside per axis side
    # exact voxel boundaries, range limits and neighbours one ulp away
    k = 0
    for ax in range(3):
        for v in [lo[ax], hi[ax], np.nextafter(hi[ax], -np.inf, dtype=np.float32),
                  np.nextafter(lo[ax], -np.inf, dtype=np.float32), lo[ax] + np.float32(vs[ax]),
                  lo[ax] + np.float32(3) * np.float32(vs[ax]), np.float32(0.0), np.float32(vs[ax] * 7)]:
            pts[k, ax] = v
            k += 1
    # clustered points so that voxels hold several points (exercise max_points)
    c = rng.integers(0, n, 40)
    for i in c:
        m = rng.integers(0, n, 30)
        pts[m, :3] = pts[i, :3] + rng.normal(0, 0.05, (30, 3)).astype(np.float32)
    return np.ascontiguousarray(pts)


def gen_voxelize():
    from oracle import build_ref
    ref = build_ref.load_prebuilt() or build_ref.build()
    rng = np.random.default_rng(7)
    out = {}
    cases = {
        "pillar": dict(vs=[0.6, 0.6, 8.0], rg=[-54, -54, -5, 54, 54, 3], T=12, MV=30000),
        "pillar_capped": dict(vs=[0.6, 0.6, 8.0], rg=[-54, -54, -5, 54, 54, 3], T=12, MV=150),
        "fine": dict(vs=[0.075, 0.075, 0.2], rg=[-54, -54, -5, 54, 54, 3], T=10, MV=60000),
        "kitti": dict(vs=[0.5, 0.5, 0.5], rg=[0, -40, -3, 70.4, 40, 1], T=5, MV=20000),
    }
    for name, c in cases.items():
        pts = boundary_points(rng, 3000, c["rg"], c["vs"])
        if name == "kitti":
            pts = pts[:, :4].copy()
        t = torch.from_numpy(pts)
        coors = torch.zeros((pts.shape[0], 3), dtype=torch.int32)
        ref.dynamic_voxelize(t, coors, c["vs"], c["rg"], 3)
        vox = torch.zeros((c["MV"], c["T"], pts.shape[1]))
        vco = torch.zeros((c["MV"], 3), dtype=torch.int32)
        vnum = torch.zeros((c["MV"],), dtype=torch.int32)
        m = ref.hard_voxelize(t, vox, vco, vnum, c["vs"], c["rg"], c["T"], c["MV"], 3, True)
        out[name + "_points"] = pts
        out[name + "_cfg"] = np.array(c["vs"] + c["rg"] + [c["T"], c["MV"]], np.float64)
        out[name + "_dyn_coors"] = coors.numpy()
        out[name + "_voxels"] = vox[:m].numpy()
        out[name + "_coors"] = vco[:m].numpy()
        out[name + "_num"] = vnum[:m].numpy()
    np.savez_compressed(os.path.join(HERE, "voxelize_ref.npz"), **out)
    print("voxelize_ref.npz", {k: v.shape for k, v in out.items() if k.endswith("_coors")})


def gen_scatter():
    g = torch.Generator().manual_seed(11)
    feats = torch.rand((4000, 3), generator=g) * 100 - 50
    coors = torch.randint(-1, 12, (4000, 3), generator=g, dtype=torch.int32)
    ref_coors = coors.unique(dim=0, sorted=True)
    ref_coors = ref_coors[ref_coors.min(dim=-1).values >= 0]
    mean, mx = [], []
    for rc in ref_coors:
        mask = (coors == rc).all(dim=-1)
        mean.append(feats[mask].mean(dim=0))
        mx.append(feats[mask].max(dim=0).values)
    np.savez_compressed(os.path.join(HERE, "scatter_ref.npz"), feats=feats.numpy(), coors=coors.numpy(),
                        ref_coors=ref_coors.numpy(), ref_mean=torch.stack(mean).numpy(),
                        ref_max=torch.stack(mx).numpy())
    print("scatter_ref.npz", ref_coors.shape)


def dense_conv_reference(idx, feats, w, B, shape, stride, padding, subm):
    """Sparse conv through its definition: dense cross-correlation of the zero-filled grid."""
    D, H, W = shape
    Cin, Cout = w.shape[-2], w.shape[-1]
    dense = torch.zeros((B, Cin, D, H, W), dtype=torch.float64)
    occ = torch.zeros((B, 1, D, H, W), dtype=torch.float64)
    ii = torch.from_numpy(idx).long()
    dense[ii[:, 0], :, ii[:, 1], ii[:, 2], ii[:, 3]] = torch.from_numpy(feats).double()
    occ[ii[:, 0], 0, ii[:, 1], ii[:, 2], ii[:, 3]] = 1.0
    wt = torch.from_numpy(w).double().permute(4, 3, 0, 1, 2).contiguous()
    ks = w.shape[:3]
    if subm:
        y = torch.nn.functional.conv3d(dense, wt, padding=[k // 2 for k in ks])
        out_idx = idx
    else:
        y = torch.nn.functional.conv3d(dense, wt, stride=stride, padding=padding)
        act = torch.nn.functional.conv3d(occ, torch.ones((1, 1, *ks), dtype=torch.float64), stride=stride,
                                         padding=padding)
        out_idx = torch.nonzero(act[:, 0] > 0).int().numpy()  # sorted (b,z,y,x)
    oi = torch.from_numpy(np.asarray(out_idx)).long()
    out = y[oi[:, 0], :, oi[:, 1], oi[:, 2], oi[:, 3]].float().numpy()
    return np.asarray(out_idx, np.int32), out


def gen_spconv():
    rng = np.random.default_rng(3)
    out = {}
    cases = [
        ("subm_k3", dict(B=2, shape=[7, 12, 10], n=260, cin=16, cout=32, ks=[3, 3, 3], st=[1, 1, 1], pd=[1, 1, 1], subm=1)),
        ("conv_s2p1", dict(B=2, shape=[9, 12, 10], n=300, cin=16, cout=32, ks=[3, 3, 3], st=[2, 2, 2], pd=[1, 1, 1], subm=0)),
        ("conv_s2p011", dict(B=1, shape=[11, 10, 12], n=240, cin=32, cout=16, ks=[3, 3, 3], st=[2, 2, 2], pd=[0, 1, 1], subm=0)),
        ("conv_311", dict(B=2, shape=[5, 9, 8], n=200, cin=16, cout=16, ks=[3, 1, 1], st=[2, 1, 1], pd=[0, 0, 0], subm=0)),
        ("subm_5to16", dict(B=1, shape=[6, 10, 10], n=150, cin=5, cout=16, ks=[3, 3, 3], st=[1, 1, 1], pd=[1, 1, 1], subm=1)),
    ]
    for name, c in cases:
        cells = c["B"] * int(np.prod(c["shape"]))
        lin = rng.choice(cells, c["n"], replace=False)
        lin.sort()
        D, H, W = c["shape"]
        idx = np.stack([lin // (D * H * W), (lin // (H * W)) % D, (lin // W) % H, lin % W], 1).astype(np.int32)
        feats = rng.normal(0, 1, (c["n"], c["cin"])).astype(np.float32)
        w = rng.normal(0, 0.2, (*c["ks"], c["cin"], c["cout"])).astype(np.float32)
        oidx, y = dense_conv_reference(idx, feats, w, c["B"], c["shape"], c["st"], c["pd"], bool(c["subm"]))
        out[name + "_cfg"] = np.array([c["B"], *c["shape"], *c["ks"], *c["st"], *c["pd"], c["subm"]], np.int32)
        out[name + "_idx"] = idx
        out[name + "_feats"] = feats
        out[name + "_w"] = w
        out[name + "_out_idx"] = oidx
        out[name + "_out"] = y
    np.savez_compressed(os.path.join(HERE, "spconv_dense_ref.npz"), **out)
    print("spconv_dense_ref.npz", {k: v.shape for k, v in out.items() if k.endswith("_out")})


if __name__ == "__main__":
    gen_voxelize()
    gen_scatter()
    gen_spconv()
