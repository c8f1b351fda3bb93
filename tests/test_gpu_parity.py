"""GPU parity tests (-m gpu): every HIP entry point, called through the C ABI (ctypes), against the CPU
oracle on the same seeded inputs, against the committed golden vectors, and -- at BASELINE config sizes --
through size-independent properties.  Integer / index outputs must be bit-exact; floating point within the
tolerance written next to each check (north_star: 1e-3 fp32 on BEV features)."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

VS = [0.075, 0.075, 0.2]
RG = [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]


def T(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t if dtype is None else t.to(dtype)


def lexsort4(a):
    return np.lexsort(tuple(a[:, j] for j in range(a.shape[1] - 1, -1, -1)))


# ------------------------------------------------------------------------------------------- A1 / A2
@pytest.mark.parametrize("case", ["pillar", "pillar_capped", "fine", "kitti"])
def test_voxelize_golden_reference_vectors(dev, golden, case):
    import isfusion_amd as m
    g = golden("voxelize_ref.npz")
    cfg = g[case + "_cfg"]
    vs, rg, Tm, MV = list(cfg[:3]), list(cfg[3:9]), int(cfg[9]), int(cfg[10])
    pts = T(g[case + "_points"], dev)
    coors = m.voxelization(pts, vs, rg, -1, -1)
    assert coors.dtype == torch.int32 and np.array_equal(coors.cpu().numpy(), g[case + "_dyn_coors"])
    v, c, n = m.voxelization(pts, vs, rg, Tm, MV)
    assert np.array_equal(c.cpu().numpy(), g[case + "_coors"])
    assert np.array_equal(n.cpu().numpy(), g[case + "_num"])
    assert np.array_equal(v.cpu().numpy(), g[case + "_voxels"])
    if Tm != -1:
        # the same through the device-resident count (isf_hard_voxelize_device: no host read-back, outputs not zeroed by
        # the caller -- fill them with garbage first: the kernel must write every padding slot of the voxels it emits)
        from isfusion_amd.voxelize import hard_voxelize_async
        junk = torch.full((MV * Tm * pts.shape[1] + 4 * MV,), float("nan"), device=dev)
        del junk                                   # the caching allocator hands the poisoned block to the next empty()
        v2, c2, n2 = hard_voxelize_async(pts, vs, rg, Tm, MV).result()
        assert np.array_equal(c2.cpu().numpy(), g[case + "_coors"]) and np.array_equal(n2.cpu().numpy(), g[case + "_num"])
        assert np.array_equal(v2.cpu().numpy(), g[case + "_voxels"])


def test_voxelize_vs_oracle_synthetic(dev, oracle_mod):
    import isfusion_amd as m
    from isfusion_amd import synthetic
    pts = synthetic.lidar_sweeps(5, 120000)
    pts[:500, 0] += 60.0  # some out-of-range rows
    d = m.voxelization(T(pts, dev), VS, RG, -1, -1).cpu().numpy()
    assert np.array_equal(d, oracle_mod.dynamic_voxelize(pts, VS, RG))
    for vs, Tm, MV in (([0.6, 0.6, 8.0], 12, 30000), ([0.6, 0.6, 8.0], 12, 1000), (VS, 10, 60000)):
        v, c, n = m.voxelization(T(pts, dev), vs, RG, Tm, MV)
        ov, oc, on = oracle_mod.hard_voxelize(pts, vs, RG, Tm, MV)
        assert np.array_equal(c.cpu().numpy(), oc) and np.array_equal(n.cpu().numpy(), on)
        assert np.array_equal(v.cpu().numpy(), ov)
    # module surface + empty input
    layer = m.Voxelization(voxel_size=[0.6, 0.6, 8], point_cloud_range=RG, max_num_points=12,
                           max_voxels=(30000, 60000)).eval()
    v, c, n = layer(T(pts[:1000], dev))
    assert v.shape[1:] == (12, 5) and c.shape[1] == 3 and n.max() <= 12
    v, c, n = m.voxelization(torch.zeros((0, 5), device=dev), [0.6, 0.6, 8.0], RG, 12, 100)
    assert v.shape == (0, 12, 5) and c.shape == (0, 3)
    v, c, n = layer.forward_async(torch.zeros((0, 5), device=dev)).result()
    assert v.shape == (0, 12, 5) and c.shape == (0, 3) and n.shape == (0,)
    pend = [layer.forward_async(T(pts[i * 30000:(i + 1) * 30000], dev)) for i in range(3)]   # several in flight
    for i, pd in enumerate(pend):
        ov, oc, on = oracle_mod.hard_voxelize(pts[i * 30000:(i + 1) * 30000], [0.6, 0.6, 8.0], RG, 12, 60000)
        v, c, n = pd.result()
        assert np.array_equal(c.cpu().numpy(), oc) and np.array_equal(n.cpu().numpy(), on)
        assert np.array_equal(v.cpu().numpy(), ov)


def test_hard_voxelize_batched_equals_the_per_sample_loop(dev, oracle_mod):
    """isf_hard_voxelize_batched_device: the samples of a batch in one pass == the oracle's hard voxelization of every
    sample, concatenated, with the sample index in front of the coordinates (what the reference's loop + cat + F.pad
    build, isfusion.py:148-176).  Samples of different sizes, an empty one, one whose voxels exceed max_voxels (capped per
    sample: the later samples' rows move up), pillar and fine grids; outputs poisoned before the call."""
    from isfusion_amd import synthetic
    from isfusion_amd.voxelize import hard_voxelize_batched_async
    sizes = [40000, 0, 70000, 25000, 3]
    pl = [synthetic.lidar_sweeps(50 + i, max(n, 1))[:n] for i, n in enumerate(sizes)]
    pl[2][:300, 1] += 200.0   # out-of-range rows
    for vs, Tm, MV in (([0.6, 0.6, 8.0], 20, 30000), ([0.6, 0.6, 8.0], 12, 900), ([0.3, 0.3, 4.0], 5, 4000)):
        for B in (5, 2, 1):
            junk = torch.full((B * MV * Tm * 5 + 64,), float("nan"), device=dev)
            del junk
            pend = hard_voxelize_batched_async([T(p, dev) for p in pl[:B]], vs, RG, Tm, MV)
            v, n, c = pend.result()
            exp = [oracle_mod.hard_voxelize(p, vs, RG, Tm, MV) for p in pl[:B]]
            assert pend.counts() == [len(e[2]) for e in exp], (vs, MV, B)
            ev = np.concatenate([e[0] for e in exp])
            ec = np.concatenate([np.concatenate([np.full((len(e[1]), 1), b, np.int32), e[1]], 1) for b, e in enumerate(exp)])
            en = np.concatenate([e[2] for e in exp])
            assert c.dtype == torch.int32 and np.array_equal(c.cpu().numpy(), ec), (vs, MV, B)
            assert np.array_equal(n.cpu().numpy(), en) and np.array_equal(v.cpu().numpy(), ev), (vs, MV, B)
    assert any(len(oracle_mod.hard_voxelize(p, [0.6, 0.6, 8.0], RG, 12, 900)[2]) == 900 for p in pl)   # the cap was hit


def test_dynamic_voxelize_batched(dev, oracle_mod):
    from isfusion_amd.voxelize import dynamic_voxelize_batched
    from isfusion_amd import synthetic
    pl = [synthetic.lidar_sweeps(i, 5000 + 100 * i) for i in range(3)]
    pts, coors = dynamic_voxelize_batched([T(p, dev) for p in pl], VS, RG)
    exp = np.concatenate([np.concatenate([np.full((p.shape[0], 1), b, np.int32),
                                          oracle_mod.dynamic_voxelize(p, VS, RG)], 1) for b, p in enumerate(pl)])
    assert np.array_equal(coors.cpu().numpy(), exp) and pts.shape[0] == exp.shape[0]


# ------------------------------------------------------------------------------------------- A3
def test_dynamic_scatter_golden_and_oracle(dev, golden, oracle_mod):
    from isfusion_amd.scatter_points import dynamic_point_to_voxel_forward
    g = golden("scatter_ref.npz")
    for red, key in (("mean", "ref_mean"), ("max", "ref_max")):
        f, c, cmap, cnt = dynamic_point_to_voxel_forward(T(g["feats"], dev), T(g["coors"], dev), red)
        assert np.array_equal(c.cpu().numpy(), g["ref_coors"])
        assert np.allclose(f.cpu().numpy(), g[key], atol=1e-2, rtol=1e-5)  # reference test tolerance
        of, oc, om, on = oracle_mod.dynamic_scatter(g["feats"], g["coors"], red)
        assert np.array_equal(cmap.cpu().numpy(), om) and np.array_equal(cnt.cpu().numpy(), on)
        assert np.allclose(f.cpu().numpy(), of, atol=1e-4, rtol=1e-5)
    f, c, cmap, cnt = dynamic_point_to_voxel_forward(T(g["feats"], dev), T(g["coors"], dev), "sum")
    of = oracle_mod.dynamic_scatter(g["feats"], g["coors"], "sum")[0]
    assert np.allclose(f.cpu().numpy(), of, atol=1e-3, rtol=1e-5)


def test_dynamic_scatter_module_edge_cases_and_grad(dev, oracle_mod):
    """mirrors reference tests/test_models/test_voxel_encoder/test_dynamic_scatter.py:9-94"""
    import isfusion_amd as m
    dsmean = m.DynamicScatter([0.32, 0.32, 6], [-74.88, -74.88, -2, 74.88, 74.88, 4], True)
    dsmax = m.DynamicScatter([0.32, 0.32, 6], [-74.88, -74.88, -2, 74.88, 74.88, 4], False)
    ef = torch.empty((0, 3), device=dev).requires_grad_()
    ec = torch.empty((0, 3), dtype=torch.int32, device=dev)
    fo, co = dsmean(ef, ec)
    fo.sum().backward()
    assert fo.shape == ef.shape and co.shape == ec.shape
    fo, co = dsmax(ef, ec)
    assert fo.shape == ef.shape
    of = (torch.rand((20000, 3), device=dev) * 100 - 50).requires_grad_()
    oc = torch.full((20000, 3), -1, dtype=torch.int32, device=dev)
    for ds in (dsmean, dsmax):
        fo, co = ds(of, oc)
        assert fo.shape[0] == 0
        fo.sum().backward()
        assert (of.grad == 0).all()
    # backward vs oracle, all three reductions
    rng = np.random.default_rng(3)
    feats = (rng.random((5000, 4), dtype=np.float32) * 100 - 50)
    coors = rng.integers(-1, 6, (5000, 3)).astype(np.int32)
    for red in ("mean", "max", "sum"):
        x = T(feats, dev).requires_grad_()
        f, c = m.dynamic_scatter(x, T(coors, dev), red)
        gout = rng.normal(size=tuple(f.shape)).astype(np.float32)
        f.backward(T(gout, dev))
        ofw, oc_, om, on = oracle_mod.dynamic_scatter(feats, coors, red)
        og = oracle_mod.dynamic_scatter_backward(gout, feats, ofw, om, on, red)
        assert np.allclose(x.grad.cpu().numpy(), og, atol=1e-5), red
    # batched (b,z,y,x) form == per-sample loop of the reference
    c4 = np.concatenate([np.sort(rng.integers(0, 3, (5000, 1)), 0), coors], 1).astype(np.int32)
    f, c = dsmax(T(feats, dev), T(c4, dev))
    of_, oc4 = oracle_mod.dynamic_scatter_batched(feats, c4, "max")
    assert np.array_equal(c.cpu().numpy(), oc4) and np.allclose(f.cpu().numpy(), of_, atol=1e-5)


# ------------------------------------------------------------------------------------------- A4
def _vfe_case(oracle_mod, dev, P=30000, B=2, seed=0):
    import isfusion_amd as m
    from isfusion_amd import synthetic
    from isfusion_amd.norm import fold_bn
    lb = m.LidarBranch().randomize_weights_(seed).randomize_bn_(seed + 1).eval()
    pl = [synthetic.lidar_sweeps(100 + seed + i, P) for i in range(B)]
    pl[0][:50, 2] = 100.0  # out of range points must be ignored
    pts = np.concatenate(pl)
    coors = np.concatenate([np.concatenate([np.full((p.shape[0], 1), b, np.int32),
                                            oracle_mod.dynamic_voxelize(p, VS, RG)], 1) for b, p in enumerate(pl)])
    vfe = lb.pts_voxel_encoder
    bn1 = [t.numpy() for t in fold_bn(vfe.vfe_layers[0].norm)]
    bn2 = [t.numpy() for t in fold_bn(vfe.vfe_layers[1].norm)]
    ovf, ovc, op2v = oracle_mod.dynamic_vfe(pts, coors, VS, RG, vfe.vfe_layers[0].linear.weight.detach().numpy(),
                                            bn1, vfe.vfe_layers[1].linear.weight.detach().numpy(), bn2)
    return lb.to(dev), pl, pts, coors, ovf, ovc, op2v


def test_dynamic_vfe_fused_vs_oracle(dev, oracle_mod):
    lb, pl, pts, coors, ovf, ovc, _ = _vfe_case(oracle_mod, dev)
    vf, vc = lb.pts_voxel_encoder(T(pts, dev), T(coors, dev))
    assert vc.dtype == torch.int32 and np.array_equal(vc.cpu().numpy(), ovc)  # sorted (b,z,y,x), bit exact
    err = np.abs(vf.cpu().numpy() - ovf).max()
    assert err < 1e-4, err  # fp32, different summation order only
    assert np.abs(ovf).max() > 0.5


def test_dynamic_vfe_voxels_longer_than_a_wave(dev, oracle_mod):
    """Voxels of 1 ... 700 points: the runs of the voxel-sorted records cross one, two, ten 64-record boundaries -- the
    cut-row accumulation of the layer kernels and the follow-the-run path of the mean kernel (isf_vfe.hip), and runs that
    end exactly on a boundary."""
    import isfusion_amd as m
    from isfusion_amd.norm import fold_bn
    rng = np.random.default_rng(7)
    lb = m.LidarBranch().randomize_weights_(5).randomize_bn_(6).eval()
    sizes = [1, 2, 63, 64, 65, 127, 128, 129, 700, 3, 64, 64, 1, 191, 320]
    pts = []
    for k, n in enumerate(sizes):   # one voxel each: cell (10 + 3k, 20 + k, 4 + k % 5) of the (x, y, z) grid
        lo = np.array([RG[0] + (10 + 3 * k) * VS[0], RG[1] + (20 + k) * VS[1], RG[2] + (4 + k % 5) * VS[2]])
        p = np.concatenate([lo + rng.uniform(0.05, 0.95, (n, 3)) * np.array(VS), rng.random((n, 2))], 1)
        pts.append(p)
    pts = np.concatenate(pts).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    coors = np.concatenate([np.zeros((len(pts), 1), np.int32), oracle_mod.dynamic_voxelize(pts, VS, RG)], 1)
    assert len(np.unique(coors, axis=0)) == len(sizes)
    vfe = lb.pts_voxel_encoder
    bn1 = [t.numpy() for t in fold_bn(vfe.vfe_layers[0].norm)]
    bn2 = [t.numpy() for t in fold_bn(vfe.vfe_layers[1].norm)]
    ovf, ovc, _ = oracle_mod.dynamic_vfe(pts, coors, VS, RG, vfe.vfe_layers[0].linear.weight.detach().numpy(), bn1,
                                         vfe.vfe_layers[1].linear.weight.detach().numpy(), bn2)
    lb = lb.to(dev)
    vf, vc = lb.pts_voxel_encoder(T(pts, dev), T(coors, dev))
    assert np.array_equal(vc.cpu().numpy(), ovc)
    err = np.abs(vf.cpu().numpy() - ovf).max()
    assert err < 1e-4, err
    vf2, _ = lb.pts_voxel_encoder(T(pts, dev), T(coors, dev))   # order independent => run to run identical
    assert torch.equal(vf, vf2)


def test_dynamic_vfe_composed_path_matches_fused(dev, oracle_mod):
    lb, pl, pts, coors, ovf, ovc, _ = _vfe_case(oracle_mod, dev, P=8000, B=2, seed=3)
    vfe = lb.pts_voxel_encoder
    with torch.no_grad():
        vf2, vc2 = vfe._forward_composed(T(pts, dev), T(coors, dev))
    keep = np.ones(len(ovc), bool)
    assert np.array_equal(vc2.cpu().numpy(), ovc)
    assert np.abs(vf2.cpu().numpy() - ovf).max() < 1e-4


# ------------------------------------------------------------------------------------------- A5 / A6
def _pairs_set(nbr, n_out):
    K = nbr.shape[0]
    s = set()
    for k in range(K):
        o = np.nonzero(nbr[k, :n_out] >= 0)[0]
        s.update(zip([k] * len(o), nbr[k, o].tolist(), o.tolist()))
    return s


@pytest.mark.parametrize("case", ["subm_k3", "conv_s2p1", "conv_s2p011", "conv_311", "subm_5to16"])
def test_rulebook_and_conv_golden_dense_identity(dev, golden, oracle_mod, case):
    from isfusion_amd import spconv as sp
    g = golden("spconv_dense_ref.npz")
    cfg = g[case + "_cfg"]
    B, shape, ks, st, pd, subm = int(cfg[0]), list(cfg[1:4]), list(cfg[4:7]), list(cfg[7:10]), list(cfg[10:13]), bool(cfg[13])
    idx, feats, w = g[case + "_idx"], g[case + "_feats"], g[case + "_w"]
    for perm_seed in (None, 1):  # rows in sorted order, then shuffled (exercises the perm path)
        order = np.arange(len(idx)) if perm_seed is None else np.random.default_rng(perm_seed).permutation(len(idx))
        rb = sp.build_rulebook(T(idx[order], dev), B, shape, ks, st, pd, subm)
        oidx, pairs, num = oracle_mod.get_indice_pairs(idx[order], B, shape, ks, st, pd, subm=subm)
        assert rb.num_out == len(oidx)
        out_idx = rb.out_indices.cpu().numpy()
        # same set of output sites; same set of (tap, in, out) pairs after mapping oracle out ids to ours
        assert np.array_equal(out_idx[lexsort4(out_idx)], oidx[lexsort4(oidx)])
        lut = {tuple(r): i for i, r in enumerate(out_idx.tolist())}
        remap = np.array([lut[tuple(r)] for r in oidx.tolist()])
        exp = set()
        for k in range(pairs.shape[0]):
            for s in range(num[k]):
                exp.add((k, int(pairs[k, 0, s]), int(remap[pairs[k, 1, s]])))
        assert _pairs_set(rb.nbr.cpu().numpy(), rb.num_out) == exp
        K = int(np.prod(ks))
        y = sp.sparse_conv_forward(T(feats[order], dev), sp.pack_filters(T(w, dev)), K, w.shape[-2], w.shape[-1], rb)
        y = y.cpu().numpy()
        gi = g[case + "_out_idx"]
        ref = g[case + "_out"][lexsort4(gi)]
        assert np.allclose(y[lexsort4(out_idx)], ref, atol=2e-4, rtol=1e-4), np.abs(y[lexsort4(out_idx)] - ref).max()


@pytest.mark.parametrize("cin,cout", [(64, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (128, 256),
                                      (256, 256), (16, 16), (16, 32)])
def test_sparse_conv_shapes_epilogue_vs_oracle(dev, oracle_mod, cin, cout):
    from isfusion_amd import spconv as sp
    rng = np.random.default_rng(cin * 1000 + cout)
    B, shape = 2, [9, 24, 20]
    cells = B * int(np.prod(shape))
    n = 700
    lin = np.sort(rng.choice(cells, n, replace=False))
    D, H, W = shape
    idx = np.stack([lin // (D * H * W), (lin // (H * W)) % D, (lin // W) % H, lin % W], 1).astype(np.int32)
    feats = rng.normal(0, 1, (n, cin)).astype(np.float32)
    for subm, ks, st, pd in ((True, [3, 3, 3], [1, 1, 1], [1, 1, 1]), (False, [3, 3, 3], [2, 2, 2], [1, 1, 1])):
        w = rng.normal(0, (1.0 / (6 * cin)) ** 0.5, (*ks, cin, cout)).astype(np.float32)
        scale = (rng.random(cout, dtype=np.float32) + 0.5)
        shift = rng.normal(0, 0.2, cout).astype(np.float32)
        rb = sp.build_rulebook(T(idx, dev), B, shape, ks, st, pd, subm)
        oidx, pairs, num = oracle_mod.get_indice_pairs(idx, B, shape, ks, st, pd, subm=subm)
        res = rng.normal(0, 1, (rb.num_out, cout)).astype(np.float32)
        y = sp.sparse_conv_forward(T(feats, dev), sp.pack_filters(T(w, dev)), 27, cin, cout, rb, T(scale, dev),
                                   T(shift, dev), T(res, dev), relu=True).cpu().numpy()
        oy = oracle_mod.indice_conv(feats, w, pairs, num, len(oidx))
        out_idx = rb.out_indices.cpu().numpy()
        o1, o2 = lexsort4(out_idx), lexsort4(oidx)
        oy = oracle_mod.bn_act(oy[o2], scale, shift, res[o1], relu=True)
        assert np.abs(y[o1] - oy).max() < 2e-4, (cin, cout, subm, np.abs(y[o1] - oy).max())
        assert (y >= 0).all() and y.max() > 0.5


@pytest.mark.parametrize("cin,cout", [(64, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (128, 256),
                                      (256, 256)])
def test_sparse_conv_f16x3_split_precision_vs_oracle(dev, oracle_mod, cin, cout):
    """Split-precision (f16 hi/lo, 3 MFMA passes, fp32 accumulate) kernel: fp32-class accuracy.
    Tolerance 2e-4 absolute on outputs of magnitude O(1) -- the same bound as the fp32 MFMA kernel."""
    from isfusion_amd import spconv as sp
    rng = np.random.default_rng(cin * 7 + cout)
    B, shape = 2, [9, 24, 20]
    cells = B * int(np.prod(shape))
    n = 1500
    lin = np.sort(rng.choice(cells, n, replace=False))
    D, H, W = shape
    idx = np.stack([lin // (D * H * W), (lin // (H * W)) % D, (lin // W) % H, lin % W], 1).astype(np.int32)
    feats = rng.normal(0, 1, (n, cin)).astype(np.float32)
    feats[::7] *= 100.0     # wide dynamic range inside one tensor
    feats[1::7] *= 1e-3
    for subm, ks, st, pd in ((True, [3, 3, 3], [1, 1, 1], [1, 1, 1]), (False, [3, 3, 3], [2, 2, 2], [1, 1, 1]),
                             (False, [3, 1, 1], [2, 1, 1], [0, 0, 0])):
        K = int(np.prod(ks))
        w = rng.normal(0, (1.0 / (6 * cin)) ** 0.5, (*ks, cin, cout)).astype(np.float32)
        scale = (rng.random(cout, dtype=np.float32) + 0.5)
        shift = rng.normal(0, 0.2, cout).astype(np.float32)
        rb = sp.build_rulebook(T(idx, dev), B, shape, ks, st, pd, subm)
        oidx, pairs, num = oracle_mod.get_indice_pairs(idx, B, shape, ks, st, pd, subm=subm)
        res = rng.normal(0, 1, (rb.num_out, cout)).astype(np.float32)
        p16 = sp.pack_filters_f16x3(T(w, dev))
        out_idx = rb.out_indices.cpu().numpy()
        o1, o2 = lexsort4(out_idx), lexsort4(oidx)
        raw = oracle_mod.indice_conv(feats, w, pairs, num, len(oidx))[o2]
        # (a) plain conv, no epilogue
        y = sp.sparse_conv_forward_f16x3(T(feats, dev), p16, K, cin, cout, rb).cpu().numpy()
        denom = np.abs(raw).max()
        assert np.abs(y[o1] - raw).max() < 3e-6 * max(denom, 1.0), (cin, cout, subm, np.abs(y[o1] - raw).max(), denom)
        # (b) BN fold + residual + ReLU epilogue
        y = sp.sparse_conv_forward_f16x3(T(feats, dev), p16, K, cin, cout, rb, T(scale, dev), T(shift, dev),
                                         T(res, dev), relu=True).cpu().numpy()
        oy = oracle_mod.bn_act(raw, scale, shift, res[o1], relu=True)
        assert np.abs(y[o1] - oy).max() < 3e-6 * max(denom, 1.0) + 1e-6
        assert (y >= 0).all()


def _random_geometry(rng, B, shape, n):
    cells = B * int(np.prod(shape))
    lin = np.sort(rng.choice(cells, n, replace=False))
    D, H, W = shape
    return np.stack([lin // (D * H * W), (lin // (H * W)) % D, (lin // W) % H, lin % W], 1).astype(np.int32)


def test_stage_tables_reconstruct_the_neighbour_table(dev):
    """isf_rulebook_stage_tables: per 64-row unit the distinct input rows (ascending) and, per table entry, its position
    in that list -- ulist[unit(o)][slots[k][o]] == nbr[k][o] wherever nbr >= 0, 0xFFFF elsewhere (rows past num_out too)."""
    from isfusion_amd import spconv as sp
    rng = np.random.default_rng(21)
    B, shape = 2, [9, 40, 36]
    idx = _random_geometry(rng, B, shape, 5000)
    for subm, ks, st, pd in ((True, [3, 3, 3], [1, 1, 1], [1, 1, 1]), (False, [3, 3, 3], [2, 2, 2], [1, 1, 1]),
                             (False, [3, 1, 1], [2, 1, 1], [0, 0, 0])):
        rb = sp.build_rulebook(T(idx, dev), B, shape, ks, st, pd, subm)
        slots, ulist, ucount = (t.cpu().numpy() for t in sp.stage_tables(rb))
        nbr = rb.nbr.cpu().numpy()
        slots = slots.view(np.uint16)
        K, stride = nbr.shape
        assert slots.shape == (K, stride) and ulist.shape[0] == stride // 64 == ucount.shape[0]
        for u in range(stride // 64):
            blk = nbr[:, u * 64:(u + 1) * 64]
            want = np.unique(blk[blk >= 0])
            assert ucount[u] == want.size and np.array_equal(ulist[u, :want.size], want), u
            sl = slots[:, u * 64:(u + 1) * 64]
            assert np.array_equal(sl == 0xFFFF, blk < 0)
            assert np.array_equal(ulist[u][sl[blk >= 0].astype(np.int64)], blk[blk >= 0])
        assert (nbr[:, rb.num_out:] < 0).all()


@pytest.mark.parametrize("cin,cout", [(64, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (128, 256),
                                      (256, 256)])
def test_staged_conv_bit_identical_to_gather_kernel(dev, cin, cout):
    """LDS-staged input rows (isf_sparse_conv_forward_staged) == the gather kernel, bit for bit: same products, same
    order.  Small LDS budgets exercise the fall-back gathers of list entries beyond a unit's share, 1 << 20 the clamp
    to what 160 KiB hold; 5000 rows over 8 XCD parts give full and half tiles and part ends inside a tile."""
    from isfusion_amd import spconv as sp
    rng = np.random.default_rng(cin * 11 + cout)
    B, shape = 2, [9, 40, 36]
    idx = _random_geometry(rng, B, shape, 5000)
    feats = rng.normal(0, 1, (5000, cin)).astype(np.float32)
    for subm, ks, st, pd in ((True, [3, 3, 3], [1, 1, 1], [1, 1, 1]), (False, [3, 3, 3], [2, 2, 2], [1, 1, 1]),
                             (False, [3, 1, 1], [2, 1, 1], [0, 0, 0])):
        K = int(np.prod(ks))
        w = rng.normal(0, (1.0 / (6 * cin)) ** 0.5, (*ks, cin, cout)).astype(np.float32)
        scale = T(rng.random(cout, dtype=np.float32) + 0.5, dev)
        shift = T(rng.normal(0, 0.2, cout).astype(np.float32), dev)
        rb = sp.build_rulebook(T(idx, dev), B, shape, ks, st, pd, subm)
        res = T(rng.normal(0, 1, (rb.num_out, cout)).astype(np.float32), dev)
        p16 = sp.pack_filters_f16x3(T(w, dev))
        x = T(feats, dev)
        for mode in (0, 1, 32):
            ref = sp.sparse_conv_forward_f16x3(x, p16, K, cin, cout, rb, scale, shift, res, relu=True, mode=mode)
            assert ref.abs().max() > 0.5
            for rows in (32, 96, 256, 1 << 20):
                got = sp.sparse_conv_forward_staged(x, p16, K, cin, cout, rb, scale, shift, res, relu=True,
                                                    stage_rows=rows, mode=mode)
                assert torch.equal(got, ref), (cin, cout, subm, mode, rows, (got - ref).abs().max().item())


def test_staged_conv_large_level_and_wide_tiles(dev):
    """the 8-wave / 256-row tile shape (128 output columns on a level of >= 2048 rows) and a dense cluster whose units
    list more rows than any LDS share"""
    from isfusion_amd import spconv as sp
    rng = np.random.default_rng(5)
    B, shape = 1, [12, 48, 48]
    idx = _random_geometry(rng, B, shape, 12000)      # 43 % occupancy: ~12 neighbours per voxel
    for cin, cout in ((128, 128), (64, 128), (32, 32)):
        feats = T(rng.normal(0, 1, (12000, cin)).astype(np.float32), dev)
        w = rng.normal(0, (1.0 / (12 * cin)) ** 0.5, (3, 3, 3, cin, cout)).astype(np.float32)
        rb = sp.build_rulebook(T(idx, dev), B, shape, [3, 3, 3], [1, 1, 1], [1, 1, 1], True)
        p16 = sp.pack_filters_f16x3(T(w, dev))
        ref = sp.sparse_conv_forward_f16x3(feats, p16, 27, cin, cout, rb)
        for rows in (64, 640, 1 << 20):
            got = sp.sparse_conv_forward_staged(feats, p16, 27, cin, cout, rb, stage_rows=rows)
            assert torch.equal(got, ref), (cin, cout, rows)


def test_split_format_roundtrip(dev):
    from isfusion_amd import spconv as sp
    x = torch.randn(1000, 64, device=dev) * torch.logspace(-6, 4, 64, device=dev)
    y = sp.from_split(sp.to_split(x), (1000, 64))
    # 22 significant bits while the low half is a normal f16 number; below that the error is absolute and
    # bounded by half an f16 subnormal step (2^-25)
    bound = torch.maximum(x.abs() * 2.0 ** -21, torch.full_like(x, 2.0 ** -24))
    assert ((y - x).abs() <= bound).all(), ((y - x).abs() / bound).max().item()
    big = x.abs() > 0.25  # low half normal (>= 2^-14) needs |x| >= 2^-3
    assert (((y - x).abs() / x.abs())[big]).max().item() < 2.0 ** -21


def test_spconv1_interchange_format_roundtrip(dev, oracle_mod):
    from isfusion_amd import _lib, spconv as sp
    rng = np.random.default_rng(9)
    B, shape = 1, [8, 16, 16]
    lin = np.sort(rng.choice(int(np.prod(shape)), 400, replace=False))
    D, H, W = shape
    idx = np.stack([lin * 0, (lin // (H * W)) % D, (lin // W) % H, lin % W], 1).astype(np.int32)
    rb = sp.build_rulebook(T(idx, dev), B, shape, [3, 3, 3], [1, 1, 1], [1, 1, 1], True)
    lib = _lib.load()
    pairs = torch.empty((27, 2, 400), dtype=torch.int32, device=dev)
    num = torch.empty((27,), dtype=torch.int32, device=dev)
    _lib.check(lib.isf_rulebook_to_indice_pairs(_lib.ptr(rb.nbr), rb.stride, rb.num_out, 27, 400, _lib.ptr(pairs),
                                                _lib.ptr(num), _lib.stream()))
    _, opairs, onum = oracle_mod.get_indice_pairs(idx, B, shape, [3, 3, 3], [1, 1, 1], [1, 1, 1], subm=True)
    assert np.array_equal(num.cpu().numpy(), onum)
    p = pairs.cpu().numpy()
    for k in range(27):
        a = set(zip(p[k, 0, :onum[k]].tolist(), p[k, 1, :onum[k]].tolist()))
        b = set(zip(opairs[k, 0, :onum[k]].tolist(), opairs[k, 1, :onum[k]].tolist()))
        assert a == b and (p[k, :, onum[k]:] == -1).all()
    nbr2 = torch.empty_like(rb.nbr)
    _lib.check(lib.isf_indice_pairs_to_rulebook(_lib.ptr(pairs), _lib.ptr(num), 27, 400, rb.num_out, _lib.ptr(nbr2),
                                                rb.stride, _lib.stream()))
    assert torch.equal(nbr2[:, :rb.num_out], rb.nbr[:, :rb.num_out])
    # drop-in op: raw (unpacked) filters through isf_sparse_conv_forward
    w = rng.normal(0, 0.1, (3, 3, 3, 16, 32)).astype(np.float32)
    feats = rng.normal(0, 1, (400, 16)).astype(np.float32)
    out = torch.empty((400, 32), device=dev)
    d_feats, d_w = T(feats, dev), T(w, dev)  # keep the device tensors alive across the asynchronous call
    _lib.check(lib.isf_sparse_conv_forward(_lib.ptr(d_feats), 400, 16, _lib.ptr(d_w), 27, 32,
                                           _lib.ptr(rb.nbr), rb.stride, 400, None, None, None, 0, _lib.ptr(out),
                                           _lib.stream()))
    assert np.abs(out.cpu().numpy() - oracle_mod.indice_conv(feats, w, opairs, onum, 400)).max() < 1e-4


# ------------------------------------------------------------------------------------------- A7
def test_dense_bev(dev, oracle_mod):
    import isfusion_amd as m
    rng = np.random.default_rng(4)
    B, shape, C = 3, [2, 180, 180], 256
    lin = rng.choice(B * int(np.prod(shape)), 5000, replace=False)  # unsorted rows on purpose
    D, H, W = shape
    idx = np.stack([lin // (D * H * W), (lin // (H * W)) % D, (lin // W) % H, lin % W], 1).astype(np.int32)
    feats = rng.normal(0, 1, (5000, C)).astype(np.float32)
    x = m.SparseConvTensor(T(feats, dev), T(idx, dev), shape, B)
    d = x.dense()
    assert d.shape == (B, C, D, H, W)
    ref = oracle_mod.dense_bev(feats, idx, B, shape)
    assert np.array_equal(d.reshape(B, C * D, H, W).cpu().numpy(), ref)


def _encoder_case(oracle_mod, dev, P, B, seed):
    lb, pl, pts, coors, ovf, ovc, _ = _vfe_case(oracle_mod, dev, P=P, B=B, seed=seed)
    bev, outs = oracle_mod.sparse_encoder_forward(lb.pts_middle_encoder.plan_to_numpy(), ovf, ovc, B)
    return lb, pl, ovf, ovc, bev, outs


def test_lidar_branch_bev_map_as_split_token_matrices(dev):
    """isf_encoder_options.bev_format = 1 (LidarBranch.forward(bev_split=True)): the BEV map leaves the branch as one
    split-format token matrix per 256-channel group -- what conv_fusion reads -- instead of fp32 [B, C*D, H, W]; converted
    back, it is the fp32 map bit for bit (hi + lo of every element is exact in fp32), empty cells included"""
    import isfusion_amd as m
    from isfusion_amd import synthetic
    lb = m.LidarBranch().randomize_weights_(0).randomize_bn_(1).eval().to(dev)
    for n, frames in ((3000, 1), (60000, 2), (300000, 3)):
        pl = [T(synthetic.lidar_sweeps(1400 + i, n), dev) for i in range(frames)]
        want = lb(pl)
        maps = lb(pl, bev_split=True)
        assert len(maps) == want.shape[1] // 256 and all(mp.C == 256 and mp.B == frames for mp in maps)
        got = torch.cat([mp.to_nchw() for mp in maps], 1)
        assert torch.equal(got, want), n
        assert (want == 0).any() and (want != 0).any()
    with pytest.raises(m._lib.IsfError):
        lb(pl, bev_split=True, precision=2)      # the f16-storage mode has no split rows to hand over


def test_sparse_encoder_and_lidar_branch_vs_oracle(dev, oracle_mod):
    """cfg-2-shaped (dynamic voxelize + DynamicVFE + full 21-layer SparseEncoder -> [B,512,180,180]) at a size
    the scalar oracle finishes in seconds.  Tolerance: 1e-3 absolute on BEV features (north_star)."""
    from isfusion_amd import _lib
    B = 2
    lb, pl, ovf, ovc, bev, outs = _encoder_case(oracle_mod, dev, P=5000, B=B, seed=7)
    enc = lb.pts_middle_encoder
    # (1) SparseEncoder.forward, fused C call, oracle VFE output as input -- both arithmetic paths
    for mode in (1, 0):  # precision 1 = fp32 MFMA kernels, 0 = default (f16x3 split-precision MFMA); per call
        stats = _lib.EncoderStats()
        sp = enc.forward_fused(T(ovf, dev), T(ovc, dev), B, stats=stats, precision=mode)
        assert stats.precision == (0 if mode == 1 else 1)
        got = sp.cpu().numpy()
        assert got.shape == (B, 512, 180, 180)
        err = np.abs(got - bev).max()
        assert err < 1e-3, (mode, err)
        assert err < 2e-4, (mode, err)  # what both paths actually achieve (fp32-class accuracy)
        assert np.array_equal(got != 0, bev != 0)
    assert np.abs(bev).max() > 1.0
    for i, (f, ix, shp) in enumerate(outs):  # voxel counts of every layer, bit exact
        assert stats.num_out[i] == f.shape[0], (i, stats.num_out[i], f.shape[0])
    # (2) the 3-tuple module surface (isfusion.py:111 unpacks x, _, kwargs)
    x, enc_feats, kw = enc(T(ovf, dev), T(ovc, dev), B, foo=1)
    assert torch.equal(x, sp) and kw == {"foo": 1}
    # (3) module-by-module path (SparseSequential / SparseBasicBlock / dense) agrees
    with torch.no_grad():
        x2, feats2 = enc.forward_modules(T(ovf, dev), T(ovc, dev), B)
    assert np.abs(x2.cpu().numpy() - bev).max() < 1e-3
    assert len(feats2) == 5 and feats2[-1].features.shape[0] == outs[-2][0].shape[0]
    # (4) unsorted input rows are re-ranked by the library
    perm = np.random.default_rng(0).permutation(len(ovc))
    sp3 = enc.forward_fused(T(ovf[perm], dev), T(ovc[perm], dev), B)
    assert np.abs(sp3.cpu().numpy() - bev).max() < 1e-3
    # (5) whole LiDAR branch in one call from raw points
    out = lb([T(p, dev) for p in pl], want_stats=True)
    err = np.abs(out.cpu().numpy() - bev).max()
    assert err < 1e-3, err
    assert lb.last_stats.num_in[0] == len(ovc)
    pairs0 = sum(int(n) for n in oracle_mod.get_indice_pairs(ovc, B, [41, 1440, 1440], [3, 3, 3], [1, 1, 1],
                                                              [1, 1, 1], subm=True)[2])
    assert lb.last_stats.pairs[0] == pairs0 and lb.last_stats.pairs[1] == pairs0


def test_chunk_split_of_the_256_column_layers(dev, oracle_mod):
    """round 6 (opt-in: conv mode 524288 / encoder diagnostic 536870912; measured slower): a tile of a 256-column layer is
    computed by two workgroups, each over half of the 32-channel chunks; the second to arrive adds the other's accumulator
    tile (exchanged through system-scope stores / loads) and runs the epilogue.  Per output element the sum over chunks is
    (lower half) + (upper half) instead of one running sum: not the bits of the unsplit kernel, but deterministic (a + b ==
    b + a: no dependence on the arrival order), independent of the batch a frame is computed in, equal between the tile
    shapes, and as close to the oracle."""
    import isfusion_amd as m
    from isfusion_amd import spconv as sp, synthetic
    KS = 536870912
    rng = np.random.default_rng(77)
    B, shape = 2, [9, 40, 36]
    for cin, cout in ((256, 256), (128, 256)):
        for n in (2500, 9000, 23000):     # one-group tiles (<= 96 x CUs rows) and a ragged last tile; 23000: the densest
            idx = _random_geometry(rng, B, shape, n)
            feats = rng.normal(0, 1, (n, cin)).astype(np.float32)
            w = rng.normal(0, (1.0 / (6 * cin)) ** 0.5, (3, 3, 3, cin, cout)).astype(np.float32)
            scale, shift = rng.random(cout, dtype=np.float32) + 0.5, rng.normal(0, 0.2, cout).astype(np.float32)
            res = rng.normal(0, 1, (n, cout)).astype(np.float32)
            rb = sp.build_rulebook(T(idx, dev), B, shape, [3, 3, 3], [1, 1, 1], [1, 1, 1], True)
            a = (T(feats, dev), sp.pack_filters_f16x3(T(w, dev)), 27, cin, cout, rb, T(scale, dev), T(shift, dev), T(res, dev))
            plain = sp.sparse_conv_forward_f16x3(*a, relu=True)
            got = sp.sparse_conv_forward_f16x3(*a, relu=True, mode=524288)
            for _ in range(3):
                assert torch.equal(sp.sparse_conv_forward_f16x3(*a, relu=True, mode=524288), got), (cin, cout, n)
            assert torch.equal(sp.sparse_conv_forward_f16x3(*a, relu=True, mode=524288 + 32), got), (cin, cout, n)
            assert (got - plain).abs().max().item() < 1e-4, (cin, cout, n)
            assert not torch.equal(got, plain)
            oidx, pairs, num = oracle_mod.get_indice_pairs(idx, B, shape, [3, 3, 3], [1, 1, 1], [1, 1, 1], subm=True)
            o1, o2 = lexsort4(rb.out_indices.cpu().numpy()), lexsort4(oidx)
            want = oracle_mod.bn_act(oracle_mod.indice_conv(feats, w, pairs, num, len(oidx))[o2], scale, shift, res[o1], relu=True)
            assert np.abs(got.cpu().numpy()[o1] - want).max() < 2e-4, (cin, cout, n)
    # shapes the split is not built for ignore the bit
    idx = _random_geometry(rng, B, shape, 3000)
    rb = sp.build_rulebook(T(idx, dev), B, shape, [3, 3, 3], [1, 1, 1], [1, 1, 1], True)
    f = T(rng.normal(0, 1, (3000, 128)).astype(np.float32), dev)
    p = sp.pack_filters_f16x3(T(rng.normal(0, 0.05, (3, 3, 3, 128, 128)).astype(np.float32), dev))
    assert torch.equal(sp.sparse_conv_forward_f16x3(f, p, 27, 128, 128, rb, mode=524288), sp.sparse_conv_forward_f16x3(f, p, 27, 128, 128, rb))
    # the LiDAR branch with the split on: repeatable; a frame alone == the frame in a batch; tile plans do not matter
    lb = m.LidarBranch().randomize_weights_(0).randomize_bn_(1).eval().to(dev)
    for n, frames in ((3000, 1), (60000, 2), (300000, 4)):
        pl = [T(synthetic.lidar_sweeps(1300 + i, n), dev) for i in range(frames)]
        got, ref = lb(pl, conv_diag=KS), lb(pl)
        assert torch.isfinite(got).all() and got.abs().max().item() > 0.1
        assert (got - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item()), n
        assert not torch.equal(got, ref), n
        assert torch.equal(lb(pl, conv_diag=KS), got) and torch.equal(lb(pl, conv_diag=KS + 32), got), n
        assert torch.equal(lb(pl, conv_diag=KS + 64), got), n
        assert torch.equal(lb(pl[:1], conv_diag=KS)[0], got[0]), n


def test_conv_neighbour_sharing_reproduces_full_gather_bits(dev):
    """the conv kernel takes a row's fragment from the right-hand lane's registers when the indices match instead of
    gathering it again (isf_spconv16.hip, load_A): the shared fragment is the fragment the load would have returned, so
    the whole LiDAR branch must give the bits of the gather-everything reference (diagnostic 16 of isf_encoder_options)"""
    import isfusion_amd as m
    from isfusion_amd import synthetic
    lb = m.LidarBranch().randomize_weights_(0).randomize_bn_(1).eval().to(dev)
    for n in (30000, 120000):
        pl = [T(synthetic.lidar_sweeps(500 + i, n), dev) for i in range(2)]
        got = lb(pl)
        want = lb(pl, conv_diag=16)
        assert torch.isfinite(got).all() and got.abs().max().item() > 0.5
        assert torch.equal(got, want), n


def test_conv_tile_mix_reproduces_uniform_tile_bits(dev):
    """launches that fit the chip in one round of workgroups are cut into full and half-height tiles so that every SIMD
    gets the same number of 16-row groups (isf_spconv16.h, conv16_plan); a row's products and their order do not depend
    on the tile it falls into, so the result must equal the uniform-tile kernel's (diagnostic 32) bit for bit -- from
    tiny inputs (half tiles only) to the full bench size (levels 3 / 4: two full + one half tile per CU)"""
    import isfusion_amd as m
    from isfusion_amd import synthetic
    lb = m.LidarBranch().randomize_weights_(0).randomize_bn_(1).eval().to(dev)
    for n, frames in ((3000, 1), (60000, 2), (300000, 4)):
        pl = [T(synthetic.lidar_sweeps(700 + i, n), dev) for i in range(frames)]
        got = lb(pl)
        want = lb(pl, conv_diag=32)
        assert torch.isfinite(got).all() and got.abs().max().item() > 0.1
        assert torch.equal(got, want), n
        assert torch.equal(lb(pl, conv_diag=48), want), n      # and without neighbour sharing
        # round 6: the deep SubM layers compute their rows in the order of their tap masks (conv_row_sort_impl: a tile / a
        # 16-row group of like rows walks fewer taps); every row still gets the same products in the same order, so the
        # un-sorted launch (diagnostic 33554432) gives the same bits
        assert torch.equal(lb(pl, conv_diag=33554432), want), n
        assert torch.equal(lb(pl, precision=2), lb(pl, precision=2, conv_diag=33554432)), n    # f16 storage: sorted too
        assert torch.equal(lb(pl, conv_diag=67108864), want), n     # the narrow layers sorted as well (opt-in)
        # round 6 (opt-in, diagnostic 16777216): the launches of several rounds (levels 0 / 1) hand their tiles out band by
        # band in y (conv16_band_order_kernel) -- a permutation of the same tiles
        assert torch.equal(lb(pl, conv_diag=16777216), want), n


@pytest.mark.parametrize("cin,cout", [(256, 256), (128, 256), (128, 128)])
def test_conv_assembly_multiply_section_reproduces_the_compiler_bits(dev, cin, cout):
    """round 6: the multiply section of the deep layers' 4-wave shape is hand-scheduled assembly (isf_spconv16_mult.h:
    column tiles in pairs, B fragments through a[0:31] two pairs ahead, one straight-line stream per (row group has the
    tap) case).  Mode 16 keeps hipcc's section: the same products in the same order per accumulator, so the outputs are
    equal bit for bit -- on sparse random geometry (most 16-row groups lack most taps: all three cases run), SubM and
    strided, with and without the epilogue terms, at sizes above the small-launch bound (two-group tiles)."""
    from isfusion_amd import spconv as sp
    rng = np.random.default_rng(cin * 3 + cout)
    B, shape = 2, [12, 96, 96]
    for n in (30000, 90000):
        idx = _random_geometry(rng, B, shape, n)
        x = T(rng.normal(0, 1, (n, cin)).astype(np.float32), dev)
        for subm, ks, st, pd in ((True, [3, 3, 3], [1, 1, 1], [1, 1, 1]), (False, [3, 3, 3], [2, 2, 2], [1, 1, 1])):
            K = int(np.prod(ks))
            rb = sp.build_rulebook(T(idx, dev), B, shape, ks, st, pd, subm)
            w = T(rng.normal(0, (1.0 / (9 * cin)) ** 0.5, (*ks, cin, cout)).astype(np.float32), dev)
            res = T(rng.normal(0, 1, (rb.num_out, cout)).astype(np.float32), dev)
            sc = T(rng.random(cout, dtype=np.float32) + 0.5, dev)
            sh = T(rng.normal(0, 0.2, cout).astype(np.float32), dev)
            p16 = sp.pack_filters_f16x3(w)
            got = sp.sparse_conv_forward_f16x3(x, p16, K, cin, cout, rb, sc, sh, res, relu=True)
            ref = sp.sparse_conv_forward_f16x3(x, p16, K, cin, cout, rb, sc, sh, res, relu=True, mode=16)
            assert ref.abs().max() > 0.5 and torch.equal(got, ref), (n, subm)
            assert torch.equal(sp.sparse_conv_forward_f16x3(x, p16, K, cin, cout, rb),
                               sp.sparse_conv_forward_f16x3(x, p16, K, cin, cout, rb, mode=16)), (n, subm)
            # mode 32768: the deep layers on isf_spconv_deep.hip (LDS-DMA gathers, a step's products and the next step's
            # loads in one hand-scheduled instruction stream; opt-in)
            assert torch.equal(sp.sparse_conv_forward_f16x3(x, p16, K, cin, cout, rb, sc, sh, res, relu=True, mode=32768),
                               ref), (n, subm)


def test_conv_one_column_block_variants_reproduce_the_two_block_bits(dev):
    """round 5: the 256-column layers with ONE column block per workgroup (conv mode 4096: 4 waves x 32 rows, 8192: 8 waves
    x 16 rows; encoder diagnostic 262144 / 524288) gather a row once per tap and chunk instead of once per column block;
    every accumulator sees the same products in the same order, so the outputs equal the production launch bit for bit.
    So does the staggered-issue variant (encoder diagnostic 1048576).  All three measured SLOWER."""
    import isfusion_amd as m
    from isfusion_amd import synthetic
    lb = m.LidarBranch().randomize_weights_(0).randomize_bn_(1).eval().to(dev)
    for n, frames in ((3000, 1), (60000, 2), (300000, 4)):
        pl = [T(synthetic.lidar_sweeps(900 + i, n), dev) for i in range(frames)]
        want = lb(pl)
        assert torch.isfinite(want).all() and want.abs().max().item() > 0.1
        assert torch.equal(lb(pl, conv_diag=262144), want), n
        assert torch.equal(lb(pl, conv_diag=524288), want), n
        assert torch.equal(lb(pl, conv_diag=262144 + 32), want), n
        # staggered issue phases inside the deep layers' workgroups (conv mode 65536): the same products in the same order
        assert torch.equal(lb(pl, conv_diag=1048576), want), n
        # round 4's issue phase (index reads inside the step, four separate weight-DMA pieces; conv mode 131072) against
        # the production one (indices a step ahead, one M0 set-up per weight run)
        assert torch.equal(lb(pl, conv_diag=2097152), want), n
        # the gathered rows two steps ahead (three register sets, counted waits; conv mode 262144), 4-wave deep shapes
        assert torch.equal(lb(pl, conv_diag=4194304), want), n


@pytest.mark.parametrize("cin,cout", [(64, 128), (128, 128), (256, 128), (128, 256), (256, 256)])
def test_small_launch_shapes_reproduce_the_two_group_bits_and_the_oracle(dev, oracle_mod, cin, cout):
    """round 5: a launch small enough that conv16_plan would cut it into half tiles only (<= 96 x CUs rows for the
    256-column layers, 2048 .. 256 x CUs rows for the 8-wave 128-column shape) runs on the ONE-GROUP instantiation (16 rows
    per wave).  Same rows per wave, same order of operations per output element as the two-group kernel, which mode 16
    (no neighbour sharing: a no-op for these shapes) still launches: bit-identical.  And against the oracle."""
    from isfusion_amd import spconv as sp
    rng = np.random.default_rng(1000 + cin + cout)
    B, shape = 2, [9, 40, 36]
    for n in (2500, 9000):   # 2500 rows: 40 tiles of 64 rows; 9000: a ragged last tile in every part
        idx = _random_geometry(rng, B, shape, n)
        feats = rng.normal(0, 1, (n, cin)).astype(np.float32)
        w = rng.normal(0, (1.0 / (6 * cin)) ** 0.5, (3, 3, 3, cin, cout)).astype(np.float32)
        scale = (rng.random(cout, dtype=np.float32) + 0.5)
        shift = rng.normal(0, 0.2, cout).astype(np.float32)
        res = rng.normal(0, 1, (n, cout)).astype(np.float32)
        rb = sp.build_rulebook(T(idx, dev), B, shape, [3, 3, 3], [1, 1, 1], [1, 1, 1], True)
        p16 = sp.pack_filters_f16x3(T(w, dev))
        a = (T(feats, dev), p16, 27, cin, cout, rb, T(scale, dev), T(shift, dev), T(res, dev))
        got = sp.sparse_conv_forward_f16x3(*a, relu=True)
        assert torch.equal(got, sp.sparse_conv_forward_f16x3(*a, relu=True, mode=16)), ("bits vs two-group", cin, cout, n)
        assert torch.equal(got, sp.sparse_conv_forward_f16x3(*a, relu=True, mode=32)), ("bits vs launch order", cin, cout, n)
        oidx, pairs, num = oracle_mod.get_indice_pairs(idx, B, shape, [3, 3, 3], [1, 1, 1], [1, 1, 1], subm=True)
        o1, o2 = lexsort4(rb.out_indices.cpu().numpy()), lexsort4(oidx)
        raw = oracle_mod.indice_conv(feats, w, pairs, num, len(oidx))[o2]
        want = oracle_mod.bn_act(raw, scale, shift, res[o1], relu=True)
        err = np.abs(got.cpu().numpy()[o1] - want).max()   # 2e-4 on O(1) outputs: the bound of the oracle test above
        assert err < 2e-4, ("oracle", cin, cout, n, err, np.abs(raw).max())


def test_conv_tile_order_is_a_per_part_permutation_and_keeps_the_bits(dev):
    """isf_sparse_conv_tile_order hands the tiles of a one-round launch to the workgroup slots longest first / least
    loaded CU first: per XCD part a permutation of the tiles, the first 32 slots (one per CU) holding the 32 heaviest
    tiles; the ordered launch computes the same tiles, so its output equals the launch-order kernel's bit for bit.
    A level too small (every CU holds at most one tile) or too large (several rounds) gets no table."""
    from isfusion_amd import spconv as sp
    rng = np.random.default_rng(77)
    B, shape = 2, [16, 96, 96]
    for n, cin, cout, applies in ((40000, 256, 256, True), (40000, 128, 128, True), (40000, 64, 64, True),
                                  (3000, 256, 256, False), (200000, 64, 64, False)):
        # dense blob + sparse rim: tile work varies
        cells = B * int(np.prod(shape))
        zz, yy, xx = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), np.arange(shape[2]), indexing="ij")
        blob = (np.abs(yy - 48) < 20) & (np.abs(xx - 48) < 20) & (zz < 8)
        p = np.tile(np.where(blob, 12.0, 1.0).ravel(), B)
        lin = np.sort(rng.choice(cells, n, replace=False, p=p / p.sum()))
        D, H, W = shape
        idx = np.stack([lin // (D * H * W), (lin // (H * W)) % D, (lin // W) % H, lin % W], 1).astype(np.int32)
        rb = sp.build_rulebook(T(idx, dev), B, shape, [3, 3, 3], [1, 1, 1], [1, 1, 1], True)
        order = sp.tile_order(rb, cin, cout)
        assert (order is not None) == applies, (n, cin, cout)
        if order is None:
            continue
        parts = 4 if cout == 256 else 8
        o = order.cpu().numpy().reshape(parts, -1)
        work = rb._tile_order[(cin, cout, 0)][1].cpu().numpy().reshape(parts, -1)
        tiles = o.shape[1]
        assert tiles > 32
        for q in range(parts):
            assert np.array_equal(np.sort(o[q]), np.arange(tiles)), q
            # replay: longest tile first (ties: lower index), to the least-loaded CU (ties: lower CU) with a free slot
            cap = [(tiles - c + 31) // 32 for c in range(32)]
            used, load, want = [0] * 32, [0] * 32, np.full(tiles, -1)
            for t in sorted(range(tiles), key=lambda t: (-work[q][t], t)):
                c = min((c for c in range(32) if used[c] < cap[c]), key=lambda c: (load[c], c))
                want[c + 32 * used[c]] = t
                used[c] += 1
                load[c] += work[q][t]
            assert np.array_equal(o[q], want), q
        assert work.min() >= 0 and work.max() <= 27 * 16
        x = T(rng.normal(0, 1, (n, cin)).astype(np.float32), dev)
        w = T(rng.normal(0, (1.0 / (9 * cin)) ** 0.5, (3, 3, 3, cin, cout)).astype(np.float32), dev)
        p16 = sp.pack_filters_f16x3(w)
        res = T(rng.normal(0, 1, (n, cout)).astype(np.float32), dev)
        for mode in (0, 1, 257):
            om = sp.tile_order(rb, cin, cout, mode)
            ref = sp.sparse_conv_forward_f16x3(x, p16, 27, cin, cout, rb, None, None, res, relu=True, mode=mode)
            got = sp.sparse_conv_forward_f16x3(x, p16, 27, cin, cout, rb, None, None, res, relu=True, mode=mode, order=om)
            assert ref.abs().max() > 0.5 and torch.equal(got, ref), (cin, cout, mode)


def test_lidar_branch_tile_order_reproduces_launch_order_bits(dev):
    """the encoder builds a tile-order table per level behind the neighbour table (geometry stream) and runs the deep
    levels' convolutions through it; diagnostic 64 switches the tables off -- same bits, at sizes where no level, the
    last level and the last two levels qualify"""
    import isfusion_amd as m
    from isfusion_amd import synthetic
    lb = m.LidarBranch().randomize_weights_(0).randomize_bn_(1).eval().to(dev)
    for n, frames in ((3000, 1), (60000, 2), (300000, 4)):
        pl = [T(synthetic.lidar_sweeps(900 + i, n), dev) for i in range(frames)]
        want = lb(pl, conv_diag=64)
        assert torch.isfinite(want).all() and want.abs().max().item() > 0.1
        assert torch.equal(lb(pl), want), n
        assert torch.equal(lb(pl, conv_diag=96), want), n       # uniform tiles, no order
        assert torch.equal(lb(pl, precision=2), lb(pl, precision=2, conv_diag=64)), n


@pytest.mark.parametrize("cin,cout", [(64, 64), (32, 32), (64, 32), (32, 64)])
def test_dma_gather_kernel_bit_identical_to_gather_kernel(dev, cin, cout):
    """isf_sparse_conv_forward_dma (narrow layers: the gathered rows come in by LDS-DMA, one cache line per lane quad,
    through a wave-private transit buffer; taps outer, one 32-channel chunk per step, neighbour indices in registers) ==
    isf_sparse_conv_forward_f16x3 bit for bit: same products, same order per accumulator.  SubM / strided / 3x1x1
    geometry, BN + residual + ReLU epilogue, split / single-pass / f16-storage modes, uniform tiles, a level of a
    few hundred rows (half tiles only, empty workgroups) and one of 40 k rows (several rounds of workgroups)."""
    from isfusion_amd import spconv as sp
    rng = np.random.default_rng(cin * 7 + cout)
    B, shape = 2, [12, 64, 64]
    for n in (300, 5000, 40000):
        idx = _random_geometry(rng, B, shape, n)
        x = T(rng.normal(0, 1, (n, cin)).astype(np.float32), dev)
        for subm, ks, st, pd in ((True, [3, 3, 3], [1, 1, 1], [1, 1, 1]), (False, [3, 3, 3], [2, 2, 2], [1, 1, 1]),
                                 (False, [3, 1, 1], [2, 1, 1], [0, 0, 0])):
            K = int(np.prod(ks))
            rb = sp.build_rulebook(T(idx, dev), B, shape, ks, st, pd, subm)
            w = T(rng.normal(0, (1.0 / (9 * cin)) ** 0.5, (*ks, cin, cout)).astype(np.float32), dev)
            res = T(rng.normal(0, 1, (rb.num_out, cout)).astype(np.float32), dev)
            sc = T(rng.random(cout, dtype=np.float32) + 0.5, dev)
            sh = T(rng.normal(0, 0.2, cout).astype(np.float32), dev)
            p16 = sp.pack_filters_f16x3(w)
            for mode in (0, 1, 257, 32):
                ref = sp.sparse_conv_forward_f16x3(x, p16, K, cin, cout, rb, sc, sh, res, relu=True, mode=mode)
                got = sp.sparse_conv_forward_dma(x, p16, K, cin, cout, rb, sc, sh, res, relu=True, mode=mode)
                assert ref.abs().max() > 0.5 and torch.equal(got, ref), (n, subm, ks, mode)
                if mode != 32:       # the tile order of the DMA kernel's OWN launch plan (not the f16x3 plan's)
                    od = sp.tile_order(rb, cin, cout, mode, dma=True)
                    if od is not None:
                        got = sp.sparse_conv_forward_dma(x, p16, K, cin, cout, rb, sc, sh, res, relu=True, mode=mode, order=od)
                        assert torch.equal(got, ref), (n, subm, ks, mode, "ordered")
            got = sp.sparse_conv_forward_dma(x, p16, K, cin, cout, rb)          # no epilogue terms
            assert torch.equal(got, sp.sparse_conv_forward_f16x3(x, p16, K, cin, cout, rb))
    with pytest.raises(Exception):
        sp.sparse_conv_forward_dma(T(np.zeros((10, 128), np.float32), dev), p16, 27, 128, 128, rb)   # wide: not built


@pytest.mark.parametrize("cin", [32, 64])
def test_dma_kernel_trace_instantiation_computes_the_same_bits(dev, cin):
    """isf_sparse_conv_dma_trace (the diagnostic instantiation of the narrow layers' kernel: tools/conv_trace.py --level
    0 | 1) == the production launch bit for bit, with the line-compressed and the dense table; its record is sane: one
    row per workgroup, stamps in order, steps <= 27 * chunks, the four cycle sums add up to a positive loop time"""
    from isfusion_amd import spconv as sp
    rng = np.random.default_rng(cin)
    B, shape, n = 2, [12, 64, 64], 20000
    idx = _random_geometry(rng, B, shape, n)
    rb = sp.build_rulebook(T(idx, dev), B, shape, [3, 3, 3], [1, 1, 1], [1, 1, 1], True)
    x = T(rng.normal(0, 1, (n, cin)).astype(np.float32), dev)
    w = T(rng.normal(0, (1.0 / (9 * cin)) ** 0.5, (3, 3, 3, cin, cin)).astype(np.float32), dev)
    res = T(rng.normal(0, 1, (rb.num_out, cin)).astype(np.float32), dev)
    sc = T(rng.random(cin, dtype=np.float32) + 0.5, dev)
    sh = T(rng.normal(0, 0.2, cin).astype(np.float32), dev)
    p16 = sp.pack_filters_f16x3(w)
    ref = sp.sparse_conv_forward_dma(x, p16, 27, cin, cin, rb, sc, sh, res, relu=True)
    xs, rs = sp.to_split(x), sp.to_split(res)
    for lines in (True, False):
        ys, tr = sp.sparse_conv_dma_trace(xs, p16, 27, cin, cin, rb, sc, sh, rs, True, lines=lines)
        assert torch.equal(sp.from_split(ys, (rb.num_out, cin)), ref), lines
        t = tr.cpu().numpy()
        t = t[t[:, 3] != 0]
        assert len(t) >= n // 128 and (t[:, 0] <= t[:, 1]).all() and (t[:, 1] <= t[:, 2]).all() and (t[:, 2] <= t[:, 3]).all()
        assert (t[:, 4] >= 1).all() and (t[:, 4] <= 27 * (cin // 32)).all()
        assert (t[:, 8:12].sum(1) > 0).all() and (t[:, 12] + t[:, 13] <= t[:, 10]).all()


@pytest.mark.parametrize("cin", [128, 256])
def test_cu_unit_kernel_bit_identical_to_tile_kernel(dev, cin):
    """isf_sparse_conv_forward_cu (256-column layers: one 8-wave workgroup per compute unit over units of equal matrix
    work, weight fragments global -> VGPR, gathered rows by LDS-DMA through a three-stage ring) ==
    isf_sparse_conv_forward_f16x3 bit for bit: same products, same order per accumulator.  SubM / strided / 3x1x1
    geometry, BN + residual + ReLU epilogue and none, from a level of 5 rows (one unit, one group) over a few hundred
    (fewer units than CUs) to 40 k rows (one unit per CU) and 120 k (several per CU); the device plan equals the host
    walk of the same arithmetic and covers every 16-row group exactly once."""
    from isfusion_amd import spconv as sp
    cout = 256
    rng = np.random.default_rng(cin)
    B, shape = 2, [12, 96, 96]
    for n in (5, 300, 5000, 40000, 120000):
        idx = _random_geometry(rng, B, shape, n)
        x = T(rng.normal(0, 1, (n, cin)).astype(np.float32), dev)
        for subm, ks, st, pd in ((True, [3, 3, 3], [1, 1, 1], [1, 1, 1]), (False, [3, 3, 3], [2, 2, 2], [1, 1, 1]),
                                 (False, [3, 1, 1], [2, 1, 1], [0, 0, 0])):
            if n == 120000 and not subm:
                continue
            K = int(np.prod(ks))
            rb = sp.build_rulebook(T(idx, dev), B, shape, ks, st, pd, subm)
            w = T(rng.normal(0, (1.0 / (9 * cin)) ** 0.5, (*ks, cin, cout)).astype(np.float32), dev)
            res = T(rng.normal(0, 1, (rb.num_out, cout)).astype(np.float32), dev)
            sc = T(rng.random(cout, dtype=np.float32) + 0.5, dev)
            sh = T(rng.normal(0, 0.2, cout).astype(np.float32), dev)
            p16 = sp.pack_filters_f16x3(w)
            ref = sp.sparse_conv_forward_f16x3(x, p16, K, cin, cout, rb, sc, sh, res, relu=True)
            got = sp.sparse_conv_forward_cu(x, p16, K, cin, cout, rb, sc, sh, res, relu=True)
            assert ref.abs().max() > 0.5 and torch.equal(got, ref), (n, subm, ks)
            got = sp.sparse_conv_forward_cu(x, p16, K, cin, cout, rb)          # no epilogue terms
            assert torch.equal(got, sp.sparse_conv_forward_f16x3(x, p16, K, cin, cout, rb)), (n, subm, ks)
            for variant in (4, 5, 6, 7, 8, 9, 10):      # 4 / 8 waves x prefetch depth 1 / 2; 8 = assembly multiply phase; 9 / 10 =
                                                        # two 4-wave workgroups per CU over units of <= 8 groups, depth 1 / 2
                got = sp.sparse_conv_forward_cu(x, p16, K, cin, cout, rb, sc, sh, res, relu=True, variant=variant)
                assert torch.equal(got, ref), (n, subm, ks, variant)
            # the plan: device == host walk; groups covered once; masks = taps with a neighbour per 16-row group
            units, masks = sp.cu_plan_units(rb)
            nbr = rb.nbr.view(K, rb.stride)[:, :rb.num_out].cpu().numpy()
            ng = (rb.num_out + 15) // 16
            pad = np.full((K, ng * 16), -1, np.int32)
            pad[:, :rb.num_out] = nbr
            want_masks = ((pad.reshape(K, ng, 16) >= 0).any(2) * (1 << np.arange(K))[:, None]).sum(0)
            assert np.array_equal(masks.numpy().astype(np.int64), want_masks)
            work = np.array([bin(int(v)).count("1") for v in want_masks], np.int32)
            cus = torch.cuda.get_device_properties(dev).multi_processor_count
            assert np.array_equal(units.numpy(), sp.cu_plan_host(work, cus)), (n, subm, ks)
            # the 8-group plan of the two-workgroups-per-CU shape: every group once, in order, 1..8 groups per unit
            u8 = sp.cu_plan_units(rb, variant=9)[0].numpy()
            assert u8[0, 0] == 0 and (u8[:, 1] >= 1).all() and (u8[:, 1] <= 8).all()
            assert np.array_equal(u8[1:, 0], (u8[:, 0] + u8[:, 1])[:-1]) and u8[-1, 0] + u8[-1, 1] == ng
    with pytest.raises(Exception):
        sp.sparse_conv_forward_cu(T(np.zeros((10, 64), np.float32), dev), p16, 27, 64, 64, rb)   # narrow: not built


def test_lidar_branch_cu_unit_layers_reproduce_tile_kernel_bits(dev):
    """with diagnostic 512 the encoder runs its 256-column layers (levels 3 / 4: 128 -> 256 strided, 4 x 256 -> 256 SubM,
    the 3-tap conv_out) on the one-workgroup-per-CU kernel, unit plans built per rulebook on the geometry stream; the
    default keeps them on the tile kernel -- same bits, from a 3 k-point frame to the bench size, repeated calls too"""
    import isfusion_amd as m
    from isfusion_amd import synthetic
    lb = m.LidarBranch().randomize_weights_(0).randomize_bn_(1).eval().to(dev)
    for n, frames in ((3000, 1), (60000, 2), (300000, 4)):
        pl = [T(synthetic.lidar_sweeps(970 + i, n), dev) for i in range(frames)]
        want = lb(pl)
        assert torch.isfinite(want).all() and want.abs().max().item() > 0.1
        got = lb(pl, conv_diag=512)
        assert torch.equal(got, want), n
        assert torch.equal(lb(pl, conv_diag=512), got), n
        assert torch.equal(lb(pl, conv_diag=512 + 64 + 128), want), n       # CU kernel on, tile order / DMA gathers off


@pytest.mark.parametrize("cin,cout", [(64, 64), (32, 32), (64, 32), (32, 64)])
def test_line_compressed_table_conv_bit_identical_and_round_trips(dev, cin, cout):
    """LINE-COMPRESSED neighbour table (9 line bases + a 27-bit tap mask per row instead of 27 indices): the dense table
    -> lines -> dense round trip is exact for rank-order tables (SubM / strided / 3x1x1), a permuted table raises the
    flag, and isf_sparse_conv_forward_dma_lines == isf_sparse_conv_forward_dma bit for bit in the split, single-pass and
    f16-storage modes, from a few hundred rows (half tiles) to several rounds of workgroups."""
    from isfusion_amd import spconv as sp
    rng = np.random.default_rng(cin * 5 + cout)
    B, shape = 2, [12, 64, 64]
    for n in (300, 5000, 40000):
        idx = _random_geometry(rng, B, shape, n)
        x = T(rng.normal(0, 1, (n, cin)).astype(np.float32), dev)
        for subm, ks, st, pd in ((True, [3, 3, 3], [1, 1, 1], [1, 1, 1]), (False, [3, 3, 3], [2, 2, 2], [1, 1, 1]),
                                 (False, [3, 1, 1], [2, 1, 1], [0, 0, 0])):
            K, tpl = int(np.prod(ks)), ks[2]
            rb = sp.build_rulebook(T(idx, dev), B, shape, ks, st, pd, subm)
            lines, mask, flag = sp.rulebook_lines(rb, tpl)
            assert int(flag.item()) == 0
            back = sp.lines_to_nbr(lines, mask, K, tpl)
            want = rb.nbr.view(K, rb.stride).clone()
            want[:, rb.num_out:] = -1
            assert torch.equal(back, want), (n, subm, ks)
            m = mask.cpu().numpy().astype(np.int64)[:rb.num_out]
            assert np.array_equal(np.array([bin(v).count("1") for v in m]), (want[:, :rb.num_out] >= 0).sum(0).cpu().numpy())
            w = T(rng.normal(0, (1.0 / (9 * cin)) ** 0.5, (*ks, cin, cout)).astype(np.float32), dev)
            res = T(rng.normal(0, 1, (rb.num_out, cout)).astype(np.float32), dev)
            sc = T(rng.random(cout, dtype=np.float32) + 0.5, dev)
            sh = T(rng.normal(0, 0.2, cout).astype(np.float32), dev)
            p16 = sp.pack_filters_f16x3(w)
            for mode in (0, 1, 257, 32):
                ref = sp.sparse_conv_forward_dma(x, p16, K, cin, cout, rb, sc, sh, res, relu=True, mode=mode)
                got = sp.sparse_conv_forward_dma_lines(x, p16, K, cin, cout, rb, sc, sh, res, relu=True, mode=mode,
                                                       taps_per_line=tpl)
                assert ref.abs().max() > 0.5 and torch.equal(got, ref), (n, subm, ks, mode)
            got = sp.sparse_conv_forward_dma_lines(x, p16, K, cin, cout, rb, taps_per_line=tpl)
            assert torch.equal(got, sp.sparse_conv_forward_dma(x, p16, K, cin, cout, rb)), (n, subm, ks)
    # a table whose rows are not in rank order has no line form: the converter says so
    rb = sp.build_rulebook(T(idx, dev), B, shape, [3, 3, 3], [1, 1, 1], [1, 1, 1], True)
    perm = torch.randperm(rb.num_out, device=dev, dtype=torch.int64)
    nb = rb.nbr.view(27, rb.stride).clone()
    live = nb[:, :rb.num_out] >= 0
    nb[:, :rb.num_out][live] = perm[nb[:, :rb.num_out][live].long()].int()
    rb2 = sp.Rulebook(nb.contiguous(), rb.stride, rb.num_in, rb.num_out, rb.out_indices, rb.out_shape)
    assert int(sp.rulebook_lines(rb2)[2].item()) == 1


def test_lidar_branch_line_tables_reproduce_dense_table_bits(dev):
    """the encoder builds the neighbour tables of its narrow levels (0 / 1: every reader runs the LDS-DMA kernel) directly
    in the line-compressed form; diagnostic 16384 keeps dense tables -- same bits from a 3 k-point frame to the bench size,
    in the fp32-class and f16-storage precisions, with and without the tile-order tables"""
    import isfusion_amd as m
    from isfusion_amd import synthetic
    lb = m.LidarBranch().randomize_weights_(0).randomize_bn_(1).eval().to(dev)
    for n, frames in ((3000, 1), (60000, 2), (300000, 4), (2000, 9)):     # 9 frames: more than one fused launch holds
        pl = [T(synthetic.lidar_sweeps(990 + i, n), dev) for i in range(frames)]
        want = lb(pl, conv_diag=16384)
        assert torch.isfinite(want).all() and want.abs().max().item() > 0.1
        assert torch.equal(lb(pl), want), n
        assert torch.equal(lb(pl, conv_diag=64), lb(pl, conv_diag=64 + 16384)), n
        assert torch.equal(lb(pl, precision=2), lb(pl, precision=2, conv_diag=16384)), n
        # the frames are voxelized inside the VFE's byte-map marking launch (one launch for the whole batch); diagnostic
        # 65536 = one dynamic-voxelize launch per frame + a separate marking pass
        assert torch.equal(lb(pl, conv_diag=65536), want), n
        # the per-level row counts reach the host through a pinned-memory mailbox (a one-thread kernel + a host spin on
        # the ticket); diagnostic 131072 = hipMemcpyAsync + synchronise
        assert torch.equal(lb(pl, conv_diag=131072), want), n
        st0 = lb(pl, want_stats=True) is not None and lb.last_stats
        st1 = lb(pl, want_stats=True, conv_diag=16384) is not None and lb.last_stats
        assert [st0.pairs[i] for i in range(21)] == [st1.pairs[i] for i in range(21)]      # same pair counts either way


@pytest.mark.parametrize("cin,cout", [(128, 128), (256, 256), (128, 256), (64, 128), (64, 64)])
def test_tile_table_conv_bit_identical_and_covers_every_group(dev, cin, cout):
    """equal-work TILE TABLE of a one-round launch (isf_sparse_conv_tile_table: the 16-row groups of an XCD's row range
    dealt to its compute units by equal work, a CU's run cut into full tiles + a remainder tile): every group of the
    launch appears in exactly one tile, tiles have 1..TM/16 groups, and isf_sparse_conv_forward_f16x3_tiled ==
    isf_sparse_conv_forward_f16x3 bit for bit (split, single-pass and f16-storage modes; SubM / strided / 3x1x1); launches
    of several rounds have no table."""
    from isfusion_amd import spconv as sp
    rng = np.random.default_rng(cin + 3 * cout)
    B, shape = 2, [12, 96, 96]
    for n in (200, 5000, 40000):
        idx = _random_geometry(rng, B, shape, n)
        x = T(rng.normal(0, 1, (n, cin)).astype(np.float32), dev)
        for subm, ks, st, pd in ((True, [3, 3, 3], [1, 1, 1], [1, 1, 1]), (False, [3, 3, 3], [2, 2, 2], [1, 1, 1]),
                                 (False, [3, 1, 1], [2, 1, 1], [0, 0, 0])):
            K = int(np.prod(ks))
            rb = sp.build_rulebook(T(idx, dev), B, shape, ks, st, pd, subm)
            if sp.tile_table(rb, cin, cout) is None:
                continue
            t = sp.tile_table(rb, cin, cout).view(-1, 2).cpu().numpy()
            ng = (rb.num_out + 15) // 16
            seen = np.zeros(ng + 64, np.int32)
            for g0, k in t:
                assert 0 <= k <= 16
                seen[g0:g0 + k] += 1
            assert (seen[:ng] == 1).all() and (seen[ng:] == 0).all(), (n, subm, ks)
            w = T(rng.normal(0, (1.0 / (9 * cin)) ** 0.5, (*ks, cin, cout)).astype(np.float32), dev)
            res = T(rng.normal(0, 1, (rb.num_out, cout)).astype(np.float32), dev)
            sc = T(rng.random(cout, dtype=np.float32) + 0.5, dev)
            sh = T(rng.normal(0, 0.2, cout).astype(np.float32), dev)
            p16 = sp.pack_filters_f16x3(w)
            for mode in (0, 1, 257):               # a table belongs to one mode: the workgroup shape depends on it
                table = sp.tile_table(rb, cin, cout, mode)
                assert table is not None
                ref = sp.sparse_conv_forward_f16x3(x, p16, K, cin, cout, rb, sc, sh, res, relu=True, mode=mode)
                got = sp.sparse_conv_forward_f16x3(x, p16, K, cin, cout, rb, sc, sh, res, relu=True, mode=mode, table=table)
                assert ref.abs().max() > 0.5 and torch.equal(got, ref), (n, subm, ks, mode)
    big = _random_geometry(rng, 4, [12, 128, 128], 200000)
    rb = sp.build_rulebook(T(big, dev), 4, [12, 128, 128], [3, 3, 3], [1, 1, 1], [1, 1, 1], True)
    assert sp.tile_table(rb, cin, cout) is None                      # several rounds of workgroups: no table


def test_lidar_branch_tile_tables_reproduce_uniform_tile_bits(dev):
    """with diagnostic 32768 the encoder gives every one-round launch of the tile kernel (levels 3 / 4; level 2 at small
    batch) an equal-work tile table built behind the neighbour table on the geometry stream; the default keeps uniform
    tiles + the tile-order permutation -- same bits from a 3 k-point frame to the bench size, fp32-class and f16 storage"""
    import isfusion_amd as m
    from isfusion_amd import synthetic
    lb = m.LidarBranch().randomize_weights_(0).randomize_bn_(1).eval().to(dev)
    for n, frames in ((3000, 1), (60000, 2), (300000, 4)):
        pl = [T(synthetic.lidar_sweeps(930 + i, n), dev) for i in range(frames)]
        want = lb(pl)
        assert torch.isfinite(want).all() and want.abs().max().item() > 0.1
        assert torch.equal(lb(pl, conv_diag=32768), want), n
        assert torch.equal(lb(pl, conv_diag=32768), want), n
        assert torch.equal(lb(pl, conv_diag=64), want), n                    # neither tables nor the permutation
        assert torch.equal(lb(pl, precision=2), lb(pl, precision=2, conv_diag=32768)), n


def test_lidar_branch_dma_gather_layers_reproduce_gather_kernel_bits(dev):
    """the encoder runs its narrow layers (levels 0 / 1) on the LDS-DMA gather kernel; diagnostic 128 keeps them on the
    gather kernel -- same bits, in the split, single-pass f16 and f16-storage precisions"""
    import isfusion_amd as m
    from isfusion_amd import synthetic
    lb = m.LidarBranch().randomize_weights_(0).randomize_bn_(1).eval().to(dev)
    for n, frames in ((3000, 1), (60000, 2), (300000, 4)):
        pl = [T(synthetic.lidar_sweeps(950 + i, n), dev) for i in range(frames)]
        want = lb(pl, conv_diag=128)
        assert torch.isfinite(want).all() and want.abs().max().item() > 0.1
        assert torch.equal(lb(pl), want), n
        assert torch.equal(lb(pl, conv_diag=32), lb(pl, conv_diag=32 + 128)), n      # uniform tiles
        # the voxel encoder writes its rows straight into the split format the first convolution reads (whole rows from
        # the segmented max, rows cut by a wave boundary from a fix-up pass); diagnostic 256 = fp32 rows + conversion pass
        assert torch.equal(lb(pl, conv_diag=256), want), n
        for prec in (2,):
            assert torch.equal(lb(pl, precision=prec), lb(pl, precision=prec, conv_diag=128)), (n, prec)


def test_lidar_branch_with_lds_staged_convs_reproduces_gather_bits(dev):
    """every conv of the LiDAR branch on the LDS-staged kernel (isf_encoder_options.stage_rows; staging tables built on
    the geometry stream) == the gather kernels, bit for bit: small LDS shares (fall-back gathers inside every tile), a
    generous one, a layer subset (tables built late for a level whose first conv ran unstaged), and the staging-off switch"""
    import isfusion_amd as m
    from isfusion_amd import synthetic
    lb = m.LidarBranch().randomize_weights_(0).randomize_bn_(1).eval().to(dev)
    for n, frames in ((3000, 1), (60000, 2), (300000, 4)):
        pl = [T(synthetic.lidar_sweeps(800 + i, n), dev) for i in range(frames)]
        want = lb(pl, stage_rows=-1)
        assert torch.isfinite(want).all() and want.abs().max().item() > 0.1
        assert torch.equal(lb(pl), want), n                     # library default
        for rows, mask in ((64, 0), (448, 0), (1 << 20, 0), (320, 0b101010101010101010100)):
            assert torch.equal(lb(pl, stage_rows=rows, stage_mask=mask), want), (n, rows, mask)


def test_concurrent_streams_have_independent_workspaces(dev):
    """the library's scratch memory (bump arena, side stream, event pool, byte maps) is one object per (device, stream):
    two host threads driving the LiDAR branch on two streams at the same time must each get the bits a lone call gives
    (round 1 had one arena per device: concurrent calls would have overwritten each other's workspace)"""
    import threading
    import isfusion_amd as m
    from isfusion_amd import synthetic
    lb = m.LidarBranch().randomize_weights_(0).randomize_bn_(1).eval().to(dev).freeze()
    inputs = {"a": [T(synthetic.lidar_sweeps(900 + i, 50000), dev) for i in range(2)],
              "b": [T(synthetic.lidar_sweeps(950 + i, 80000), dev) for i in range(1)]}
    want = {k: lb(v).clone() for k, v in inputs.items()}
    torch.cuda.synchronize()
    got, errors = {}, []

    def run(name):
        try:
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                for _ in range(4):                      # several calls each, so the two threads really overlap
                    out = lb(inputs[name])
                stream.synchronize()
            got[name] = out
        except Exception as e:   # noqa: BLE001
            errors.append((name, repr(e)))

    threads = [threading.Thread(target=run, args=(k,)) for k in inputs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k in inputs:
        assert torch.equal(got[k], want[k]), k


def test_conv_autograd_switch(dev):
    """a sparse conv on tensors that require grad runs the autograd Function (forward + dX / dW kernels, validated in
    tests/test_gpu_widened.py); with spconv.TRAINING_KERNELS = False the same call is a loud NotImplementedError"""
    import isfusion_amd as m
    from isfusion_amd import spconv
    conv = m.SubMConv3d(16, 16, 3, padding=1, bias=False).to(dev)
    x = m.SparseConvTensor(torch.rand(10, 16, device=dev), torch.zeros((10, 4), dtype=torch.int32, device=dev),
                           [4, 4, 4], 1)
    x.indices[:, 3] = torch.arange(10, device=dev) % 4
    x.indices[:, 2] = torch.arange(10, device=dev) // 4
    conv(x).features.sum().backward()
    assert conv.weight.grad is not None and torch.isfinite(conv.weight.grad).all()
    spconv.TRAINING_KERNELS = False
    try:
        with pytest.raises(NotImplementedError):
            conv(x)
    finally:
        spconv.TRAINING_KERNELS = True


# ------------------------------------------------------------------------------------------- full size
def test_full_size_numeric_parity_cfg2_frame(dev, oracle_mod):
    """BASELINE configs[1] at FULL size, numerically: one 300k-point bench frame through the whole LiDAR branch, HIP vs
    the C oracle (its conv loop runs on all host threads: seconds).  A failure class the small cases cannot see lives
    here (tiles in every wave slot, the big-level workgroup shapes, prefetch rings at depth).  1e-3 on BEV features
    (north_star), identical non-zero mask, per-layer voxel counts and pair counts exact; the frame is also run inside a
    batch of 4 (the benchmark's shape) and must reproduce its single-frame bits."""
    import isfusion_amd as m
    from isfusion_amd import synthetic
    from isfusion_amd.norm import fold_bn
    P = 300000
    pts = synthetic.lidar_sweeps(1234 + 2000, P)            # bench.py frame 0
    lb = m.LidarBranch().randomize_weights_(0).randomize_bn_(1).eval()
    vfe = lb.pts_voxel_encoder
    coors = np.concatenate([np.zeros((P, 1), np.int32), oracle_mod.dynamic_voxelize(pts, VS, RG)], 1)
    bn1 = [t.numpy() for t in fold_bn(vfe.vfe_layers[0].norm)]
    bn2 = [t.numpy() for t in fold_bn(vfe.vfe_layers[1].norm)]
    ovf, ovc, _ = oracle_mod.dynamic_vfe(pts, coors, VS, RG, vfe.vfe_layers[0].linear.weight.detach().numpy(), bn1,
                                         vfe.vfe_layers[1].linear.weight.detach().numpy(), bn2)
    obev, outs = oracle_mod.sparse_encoder_forward(lb.pts_middle_encoder.plan_to_numpy(), ovf, ovc, 1)
    lb = lb.to(dev)
    out = lb([T(pts, dev)], want_stats=True)
    st = lb.last_stats
    got = out.cpu().numpy()
    assert got.shape == obev.shape == (1, 512, 180, 180)
    assert st.num_in[0] == len(ovc)
    err = np.abs(got - obev).max()
    assert err < 1e-3, err
    # same occupied BEV cells; element-wise the post-ReLU zero pattern may differ only where both values are ~0
    # (a pre-activation within rounding distance of zero)
    assert np.array_equal((got != 0).any(1), (obev != 0).any(1)), "occupied BEV cells differ from the oracle"
    flips = (got != 0) != (obev != 0)
    assert np.maximum(np.abs(got), np.abs(obev))[flips].max(initial=0.0) < 1e-4 and flips.mean() < 1e-4
    assert np.abs(obev).max() > 0.5
    others = [T(synthetic.lidar_sweeps(1234 + 2000 + i, P), dev) for i in (1, 2, 3)]
    out4 = lb([others[0], T(pts, dev), others[1], others[2]])
    assert torch.equal(out4[1], out[0]), "the frame's bits depend on its batch"


def test_full_size_properties_cfg2(dev):
    """BASELINE configs[1]: B=4 x 300k points.  No oracle at this size: size-independent properties."""
    import isfusion_amd as m
    from isfusion_amd import synthetic
    from isfusion_amd.voxelize import dynamic_voxelize_batched
    B, P = 4, 300000
    pl = [T(synthetic.lidar_sweeps(1234 + 2000 + i, P), dev) for i in range(B)]
    lb = m.LidarBranch().randomize_weights_(0).randomize_bn_(1).eval().to(dev)
    out = lb(pl, want_stats=True)
    st = lb.last_stats
    assert out.shape == (B, 512, 180, 180) and torch.isfinite(out).all()
    # voxel count == number of distinct in-range coordinates (torch.unique as independent counter)
    pts, coors = dynamic_voxelize_batched(pl, VS, RG)
    uniq = torch.unique(coors[(coors[:, 1:] >= 0).all(1)], dim=0)
    assert st.num_in[0] == uniq.shape[0]
    # SubM layers keep the active set; strided layers: N_out <= 8 * N_in and pairs >= N_out
    tab = lb.conv_layer_table()
    for i, (kind, cin, cout, K) in enumerate(tab):
        if kind == "subm":
            assert st.num_out[i] == st.num_in[i] and st.pairs[i] >= st.num_in[i]
        else:
            assert 0 < st.num_out[i] <= 8 * st.num_in[i] and st.pairs[i] >= st.num_out[i]
    # determinism: same inputs -> bit-identical BEV (no atomics in the float path except the exact
    # fixed-point centroid sums and max reductions)
    out2 = lb(pl)
    assert torch.equal(out, out2)
    # frames are independent: running sample 2 alone reproduces its slice of the batch
    out_single = lb([pl[2]])
    assert torch.equal(out_single[0], out[2])
    # active BEV cells == cells of the last level's voxels
    nz_cells = (out != 0).any(dim=1).sum().item()
    assert 0 < nz_cells <= st.num_out[len(tab) - 1]
    # hard (pillar) voxelization at full size: voxels are distinct, counts within bounds, first points are
    # increasing in first-appearance order
    v, c, n = m.voxelization(pl[0], [0.6, 0.6, 8.0], RG, 12, 30000)
    assert torch.unique(c, dim=0).shape[0] == c.shape[0] and n.min() >= 1 and n.max() <= 12
    first = v[:, 0, :]
    # every stored first point really lies in its voxel
    cx = torch.floor((first[:, 0] - RG[0]) / 0.6).int()
    cy = torch.floor((first[:, 1] - RG[1]) / 0.6).int()
    assert torch.equal(cx, c[:, 2]) and torch.equal(cy, c[:, 1])
