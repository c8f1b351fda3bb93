"""CPU tests: the oracle (oracle/isf_oracle.c) against the committed golden vectors.

These pin the checker itself: reference-C++-generated voxelization vectors, the known-answer vector of the
reference's test_voxel_generator.py, the brute-force DynamicScatter reference of the reference's
test_dynamic_scatter.py, and the dense-conv3d identity for sparse convolution.
"""
import numpy as np
import pytest
import torch


def _cfg(a):
    vs, rg = list(a[:3]), list(a[3:9])
    return vs, rg, int(a[9]), int(a[10])


@pytest.mark.parametrize("case", ["pillar", "pillar_capped", "fine", "kitti"])
def test_voxelize_matches_reference_cpp(oracle_mod, golden, case):
    g = golden("voxelize_ref.npz")
    vs, rg, T, MV = _cfg(g[case + "_cfg"])
    pts = g[case + "_points"]
    assert np.array_equal(oracle_mod.dynamic_voxelize(pts, vs, rg), g[case + "_dyn_coors"])
    v, c, n = oracle_mod.hard_voxelize(pts, vs, rg, T, MV)
    assert np.array_equal(c, g[case + "_coors"])
    assert np.array_equal(n, g[case + "_num"])
    assert np.array_equal(v, g[case + "_voxels"])


def test_voxel_generator_known_answer(oracle_mod):
    # reference tests/test_models/test_voxel_encoder/test_voxel_generator.py:6-22
    np.random.seed(0)
    pts = np.random.rand(1000, 4).astype(np.float32)
    _, coors, num = oracle_mod.hard_voxelize(pts, [0.5, 0.5, 0.5], [0, -40, -3, 70.4, 40, 1], 1000, 20000)
    exp = np.array([[7, 81, 1], [6, 81, 0], [7, 80, 1], [6, 81, 1], [7, 81, 0], [6, 80, 1], [7, 80, 0], [6, 80, 0]])
    assert np.array_equal(coors, exp)
    assert np.array_equal(num, [120, 121, 127, 134, 115, 127, 125, 131])


def test_dynamic_scatter_matches_bruteforce(oracle_mod, golden):
    g = golden("scatter_ref.npz")
    for red, key in (("mean", "ref_mean"), ("max", "ref_max")):
        f, c, cmap, cnt = oracle_mod.dynamic_scatter(g["feats"], g["coors"], red)
        assert np.array_equal(c, g["ref_coors"])
        # tolerance of the reference test (test_dynamic_scatter.py:80-84)
        assert np.allclose(f, g[key], atol=1e-2, rtol=1e-5)
        valid = (g["coors"] >= 0).all(1)
        assert ((cmap >= 0) == valid).all()
        assert np.array_equal(c[cmap[valid]], g["coors"][valid])
        assert cnt.sum() == valid.sum()


def test_dynamic_scatter_edge_cases(oracle_mod):
    # test_dynamic_scatter.py:23-54: empty input, every row invalid
    f, c, m, n = oracle_mod.dynamic_scatter(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32), "mean")
    assert f.shape == (0, 3) and c.shape == (0, 3)
    feats = np.random.default_rng(0).random((500, 3), dtype=np.float32)
    coors = -np.ones((500, 3), np.int32)
    f, c, m, n = oracle_mod.dynamic_scatter(feats, coors, "max")
    assert f.shape[0] == 0 and (m == -1).all()
    g = oracle_mod.dynamic_scatter_backward(np.zeros((0, 3), np.float32), feats, f, m, n, "max")
    assert (g == 0).all()


def test_dynamic_scatter_backward_matches_autograd(oracle_mod):
    rng = np.random.default_rng(5)
    feats = (rng.random((300, 4), dtype=np.float32) * 100 - 50)
    coors = rng.integers(-1, 3, (300, 3)).astype(np.int32)
    for red in ("mean", "max", "sum"):
        f, c, cmap, cnt = oracle_mod.dynamic_scatter(feats, coors, red)
        gout = rng.normal(size=f.shape).astype(np.float32)
        g = oracle_mod.dynamic_scatter_backward(gout, feats, f, cmap, cnt, red)
        # autograd reference with torch index ops
        x = torch.from_numpy(feats).double().requires_grad_()
        m = torch.from_numpy(cmap).long()
        valid = m >= 0
        M = f.shape[0]
        if red == "max":
            out = torch.full((M, 4), -float("inf"), dtype=torch.float64)
            out = out.scatter_reduce(0, m[valid][:, None].expand(-1, 4), x[valid], "amax", include_self=True)
        else:
            out = torch.zeros((M, 4), dtype=torch.float64).index_add(0, m[valid], x[valid])
            if red == "mean":
                out = out / torch.from_numpy(cnt).double()[:, None]
        out.backward(torch.from_numpy(gout).double())
        assert np.allclose(g, x.grad.numpy(), atol=1e-5), red


@pytest.mark.parametrize("case", ["subm_k3", "conv_s2p1", "conv_s2p011", "conv_311", "subm_5to16"])
def test_sparse_conv_matches_dense_conv3d(oracle_mod, golden, case):
    g = golden("spconv_dense_ref.npz")
    cfg = g[case + "_cfg"]
    B, shape, ks, st, pd, subm = int(cfg[0]), list(cfg[1:4]), list(cfg[4:7]), list(cfg[7:10]), list(cfg[10:13]), bool(cfg[13])
    idx, feats, w = g[case + "_idx"], g[case + "_feats"], g[case + "_w"]
    out_idx, pairs, num = oracle_mod.get_indice_pairs(idx, B, shape, ks, st, pd, subm=subm)
    y = oracle_mod.indice_conv(feats, w, pairs, num, out_idx.shape[0])
    # the CPU reference numbers strided outputs first-come; compare after sorting by (b,z,y,x)
    key = lambda a: np.lexsort((a[:, 3], a[:, 2], a[:, 1], a[:, 0]))
    o = key(out_idx)
    assert np.array_equal(out_idx[o], g[case + "_out_idx"][key(g[case + "_out_idx"])])
    assert np.allclose(y[o], g[case + "_out"][key(g[case + "_out_idx"])], atol=2e-4, rtol=1e-4)


def test_rulebook_conventions(oracle_mod):
    """Hand-checkable case: one voxel, stride-2 k3 p1 conv -> the outputs and taps geometry.h defines."""
    idx = np.array([[0, 3, 4, 5]], np.int32)
    out_idx, pairs, num = oracle_mod.get_indice_pairs(idx, 1, [8, 8, 8], [3, 3, 3], [2, 2, 2], [1, 1, 1])
    # z=3 (odd): o in {2,1} with taps k = 3 - 2o + 1 = {0,2}; y=4 (even): o=2,k=1; x=5 (odd): o in {3,2}, k={0,2}
    got = {(tuple(out_idx[pairs[k, 1, 0]][1:]), k) for k in range(27) if num[k]}
    exp = set()
    for oz, kz in ((2, 0), (1, 2)):
        for ox, kx in ((3, 0), (2, 2)):
            exp.add(((oz, 2, ox), (kz * 3 + 1) * 3 + kx))
    assert got == exp and num.sum() == 4
    # first-come numbering: largest output coordinate first, x fastest (geometry.h:60-83)
    assert out_idx[0].tolist() == [0, 2, 2, 3] and out_idx[1].tolist() == [0, 2, 2, 2]


def test_sparse_encoder_shape_known_answer(oracle_mod):
    """reference tests/test_models/test_common_modules/test_middle_encoders.py:8-27: [4,256,128,128]
    (the level shapes only; run on a thin slab of voxels so the scalar oracle stays fast)."""
    shape = [40, 1024, 1024]
    for _ in range(3):
        shape = oracle_mod.conv_out_shape(shape, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    shape = oracle_mod.conv_out_shape(shape, [3, 1, 1], [2, 1, 1], [0, 0, 0])
    assert [128 * shape[0], shape[1], shape[2]] == [256, 128, 128]
    # IS-Fusion config: [41,1440,1440] -> [2,180,180], 256*2 = 512 channels (sparse_encoder.py:75-104)
    s = [41, 1440, 1440]
    s = oracle_mod.conv_out_shape(s, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    s = oracle_mod.conv_out_shape(s, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    s = oracle_mod.conv_out_shape(s, [3, 3, 3], [2, 2, 2], [0, 1, 1])
    s = oracle_mod.conv_out_shape(s, [3, 1, 1], [2, 1, 1], [0, 0, 0])
    assert s == [2, 180, 180]


def test_dynamic_vfe_matches_torch_composition(oracle_mod):
    """oracle DynamicVFE vs the reference's composition written with torch ops (voxel_encoder.py:453-547)."""
    rng = np.random.default_rng(2)
    vs, rg = [0.5, 0.5, 1.0], [0, 0, 0, 8, 8, 4]
    P = 600
    pts = rng.random((P, 5), dtype=np.float32) * np.array([8, 8, 4, 1, 1], np.float32)
    coors3 = oracle_mod.dynamic_voxelize(pts, vs, rg)
    b = (np.arange(P) >= P // 2).astype(np.int32)
    coors4 = np.concatenate([b[:, None], coors3], 1).astype(np.int32)
    w1 = rng.normal(0, 0.3, (64, 11)).astype(np.float32)
    w2 = rng.normal(0, 0.2, (64, 128)).astype(np.float32)
    bn = lambda: (rng.random(64, dtype=np.float32) + 0.5, rng.normal(0, 0.1, 64).astype(np.float32))
    bn1, bn2 = bn(), bn()
    vf, vc, p2v = oracle_mod.dynamic_vfe(pts, coors4, vs, rg, w1, bn1, w2, bn2)
    # torch composition
    x = torch.from_numpy(pts)
    c = torch.from_numpy(coors4).long()
    uc, inv = torch.unique(c, dim=0, sorted=True, return_inverse=True)
    assert np.array_equal(uc.numpy(), vc) and np.array_equal(inv.numpy(), p2v)
    N = uc.shape[0]
    mean = torch.zeros((N, 3)).index_add(0, inv, x[:, :3]) / torch.bincount(inv, minlength=N)[:, None]
    f_cluster = x[:, :3] - mean[inv]
    off = [vs[0] / 2 + rg[0], vs[1] / 2 + rg[1], vs[2] / 2 + rg[2]]
    f_center = torch.stack([x[:, 0] - (c[:, 3].float() * vs[0] + off[0]), x[:, 1] - (c[:, 2].float() * vs[1] + off[1]),
                            x[:, 2] - (c[:, 1].float() * vs[2] + off[2])], 1)
    f = torch.cat([x, f_cluster, f_center], 1)
    h1 = torch.relu(f @ torch.from_numpy(w1).T * torch.from_numpy(bn1[0]) + torch.from_numpy(bn1[1]))
    vmax1 = torch.full((N, 64), -float("inf")).scatter_reduce(0, inv[:, None].expand(-1, 64), h1, "amax")
    g = torch.cat([h1, vmax1[inv]], 1)
    h2 = torch.relu(g @ torch.from_numpy(w2).T * torch.from_numpy(bn2[0]) + torch.from_numpy(bn2[1]))
    ref = torch.full((N, 64), -float("inf")).scatter_reduce(0, inv[:, None].expand(-1, 64), h2, "amax")
    assert np.allclose(vf, ref.numpy(), atol=1e-4, rtol=1e-4)


# ------------------------------------------------------------------------------------------- input pre-pass (8f #3)
@pytest.mark.parametrize("case", ["test", "remove_close", "train_aug"])
@pytest.mark.parametrize("name", ["a", "b", "key_only"])
def test_input_pipeline_restatement_matches_reference(golden, name, case):
    """LoadPointsFromMultiSweeps + GlobalRotScaleTransV2 + RandomFlip3DV2 + PointsRangeFilter: the restatement equals
    the reference's own output bit for bit (float64 sensor poses, float32 augmentation)"""
    from input_common import INPUT_CONFIGS, PC_RANGE, sweep_inputs, train_aug
    from oracle import input_ops
    g = golden("input_ref.npz")
    seed, key_n, sweep_ns = INPUT_CONFIGS[name]
    key, sweeps, ts = sweep_inputs(seed, key_n, sweep_ns)
    out = input_ops.load_frame(key, sweeps, ts, PC_RANGE, drop_close=(case == "remove_close"),
                               aug=train_aug(77) if case == "train_aug" else None)
    ref = g[f"{name}.{case}.points"]
    assert out.dtype == np.float32 and out.shape == ref.shape
    assert np.array_equal(out, ref)
    # key-frame points come first with a zero time column; sweep points carry a positive lag
    n_key = int((ref[:, 4] == 0).sum())
    assert 0 < n_key <= key_n and (ref[:n_key, 4] == 0).all() and (ref[n_key:, 4] > 0).all()


@pytest.mark.parametrize("name,seed,P,nf", [("nf4", 31, 3000, 4), ("nf5", 32, 2500, 5)])
def test_hard_simple_vfe_restatement_matches_reference(golden, oracle_mod, name, seed, P, nf):
    """BASELINE configs[0]'s VFE: HardSimpleVFE (voxel_encoder.py:14-45) restatement vs the reference's output"""
    from isfusion_amd import synthetic
    v, c, n = oracle_mod.hard_voxelize(synthetic.lidar_sweeps(seed, P), [0.075, 0.075, 0.2],
                                       [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0], 10, 20000)
    ref = golden("vfe_ref.npz")[name]
    out = oracle_mod.hard_simple_vfe(v, n, nf)
    assert out.shape == ref.shape and np.abs(out - ref).max() < 1e-6


@pytest.mark.parametrize("case", ["subm_k3", "conv_s2p1", "conv_s2p011", "conv_311", "subm_5to16"])
def test_sparse_conv_backward_matches_dense_conv3d_autograd(oracle_mod, golden, case):
    """indice_conv_backward restatement (spconv_ops.h:363-456) vs torch autograd through the dense-conv3d identity
    the forward goldens rest on: y = conv3d(dense(x), W)[output sites]; dL/dx, dL/dW for a random dL/dy."""
    import torch.nn.functional as F
    g = golden("spconv_dense_ref.npz")
    cfg = g[case + "_cfg"]
    B, shape, ks, st, pd, subm = int(cfg[0]), [int(v) for v in cfg[1:4]], [int(v) for v in cfg[4:7]], \
        [int(v) for v in cfg[7:10]], [int(v) for v in cfg[10:13]], bool(cfg[13])
    idx, feats, w = g[case + "_idx"], g[case + "_feats"], g[case + "_w"]
    out_idx, pairs, num = oracle_mod.get_indice_pairs(idx, B, shape, ks, st, pd, subm=subm)
    gout = np.random.default_rng(5).normal(size=(out_idx.shape[0], w.shape[-1])).astype(np.float32)
    dx, dw = oracle_mod.indice_conv_backward(feats, w, gout, pairs, num)
    # dense identity in float64
    x = torch.from_numpy(feats).double().requires_grad_()
    wt = torch.from_numpy(w).double().requires_grad_()
    ii = torch.from_numpy(idx).long()
    dense = torch.zeros((B, *shape, feats.shape[1]), dtype=torch.float64)
    dense = dense.index_put((ii[:, 0], ii[:, 1], ii[:, 2], ii[:, 3]), x).permute(0, 4, 1, 2, 3)
    y = F.conv3d(dense, wt.permute(4, 3, 0, 1, 2), stride=st, padding=pd)
    oo = torch.from_numpy(out_idx).long()
    ys = y[oo[:, 0], :, oo[:, 1], oo[:, 2], oo[:, 3]]
    ys.backward(torch.from_numpy(gout).double())
    assert np.abs(dx - x.grad.numpy()).max() < 1e-4
    assert np.abs(dw - wt.grad.numpy()).max() < 2e-4 * max(1.0, np.abs(wt.grad.numpy()).max())


# ------------------------------------------------------------------------------------------- encoder wiring (A5-A7)
@pytest.mark.parametrize("name", ["isfusion", "conv_module"])
def test_encoder_plan_matches_reference_module_tree(golden, oracle_mod, name):
    """the exported layer plan (stage wiring, paddings, residuals, BN fold, dense layout) run by the oracle equals the
    REFERENCE's SparseEncoder / SparseBasicBlock / spconv Python layer running over the same two native ops
    (tests/golden/make_golden_encoder.py; the state dict was loaded into the reference module with strict=True)"""
    import isfusion_amd as m
    from encoder_common import ENCODER_CASES, encoder_input
    g = golden("encoder_ref.npz")
    case = ENCODER_CASES[name]
    cfg = dict(case["cfg"])
    lb = m.LidarBranch(pts_middle_encoder=cfg).randomize_weights_(case["seed"]).randomize_bn_(case["seed"] + 1).eval()
    feats, coors, B = encoder_input(case)
    bev, outs = oracle_mod.sparse_encoder_forward(lb.pts_middle_encoder.plan_to_numpy(), feats, coors, B)
    assert list(bev.shape) == g[name + ".shape"].tolist()
    assert np.abs(bev.reshape(-1)[g[name + ".idx"]] - g[name + ".val"]).max() < 1e-4
    assert int(np.count_nonzero(bev)) == int(g[name + ".nonzero"][0])


@pytest.mark.parametrize("name,seed,B,P", [("b2", 41, 2, 3000), ("b1", 43, 1, 5000)])
def test_dynamic_vfe_restatement_matches_reference_module(golden, oracle_mod, name, seed, B, P):
    """DynamicVFE (A4): the fused restatement vs the REFERENCE's DynamicVFE module + Python DynamicScatter wrapper
    running over the oracle's scatter op (tests/golden/make_golden_dynvfe.py; strict state-dict load)"""
    import isfusion_amd as m
    from isfusion_amd import synthetic
    from isfusion_amd.norm import fold_bn
    VS, RG = [0.075, 0.075, 0.2], [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]
    g = golden("dynvfe_ref.npz")
    pl = []
    for i in range(B):
        p = synthetic.lidar_sweeps(seed + i, P)
        pl.append(p[(oracle_mod.dynamic_voxelize(p, VS, RG) >= 0).all(1)])
    coors = np.concatenate([np.concatenate([np.full((p.shape[0], 1), b, np.int32),
                                            oracle_mod.dynamic_voxelize(p, VS, RG)], 1) for b, p in enumerate(pl)])
    vfe = m.LidarBranch().randomize_weights_(seed).randomize_bn_(seed + 1).eval().pts_voxel_encoder
    bn1 = [t.numpy() for t in fold_bn(vfe.vfe_layers[0].norm)]
    bn2 = [t.numpy() for t in fold_bn(vfe.vfe_layers[1].norm)]
    vf, vc, _ = oracle_mod.dynamic_vfe(np.concatenate(pl), coors, VS, RG,
                                       vfe.vfe_layers[0].linear.weight.detach().numpy(), bn1,
                                       vfe.vfe_layers[1].linear.weight.detach().numpy(), bn2)
    assert np.array_equal(vc, g[name + ".voxel_coors"])
    assert np.abs(vf[::4] - g[name + ".voxel_feats_every4"]).max() < 1e-4
    assert np.abs(vf.astype(np.float64).sum(0) - g[name + ".feat_sums"]).max() < 1e-2


def test_oracle_conv_is_bit_identical_across_thread_counts(oracle_mod):
    """the OpenMP loop of the oracle's conv (pairs of one tap in parallel, taps sequential) must not change a bit"""
    import os
    import subprocess
    import sys
    code = """
import sys, hashlib
sys.path.insert(0, %r)
import numpy as np, oracle
rng = np.random.default_rng(0)
cells = np.sort(rng.choice(2 * 9 * 40 * 40, 4000, replace=False))
idx = np.stack(np.unravel_index(cells, (2, 9, 40, 40)), 1).astype(np.int32)
x = rng.normal(size=(4000, 32)).astype(np.float32)
w = rng.normal(size=(3, 3, 3, 32, 64)).astype(np.float32)
o, p, n = oracle.get_indice_pairs(idx, 2, [9, 40, 40], [3, 3, 3], [2, 2, 2], [1, 1, 1])
y = oracle.indice_conv(x, w, p, n, o.shape[0])
print(oracle.num_threads(), hashlib.sha256(y.tobytes()).hexdigest())
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for t in ("1", "4"):
        env = dict(os.environ, OMP_NUM_THREADS=t)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True)
        n, h = r.stdout.split()
        out[t] = h
        assert int(n) in (1, int(t))          # 1 when the oracle was built without OpenMP
    assert out["1"] == out["4"]
