"""GPU parity tests (-m gpu) of the HSF / IGF rows (A8, A10-A14): each HIP entry point through
the C ABI against the CPU restatement (oracle/fusion_ops.py) on seeded inputs, and the whole ISFusionEncoder against
the golden vectors the reference's own Python produced.  Index outputs bit-exact; fp32 tolerances written per check
(north_star: 1e-3 on BEV features)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from fusion_common import CONFIGS, build_modules, check_sample, state_dicts, torch_inputs

pytestmark = pytest.mark.gpu


def rnd(shape, seed, scale=1.0, dev="cpu"):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


# ------------------------------------------------------------------------------------------------ linear
@pytest.mark.parametrize("M,K,N", [(1, 32, 16), (63, 64, 128), (200, 128, 256), (4097, 128, 384), (777, 256, 768),
                                   (32400, 128, 128)])
def test_linear_plain_matches_fp32(dev, M, K, N):
    from isfusion_amd import fusion_ops as ops
    x, w, b = rnd((M, K), 1), rnd((N, K), 2, K ** -0.5), rnd((N,), 3, 0.1)
    y = ops.linear(x.to(dev), ops.PackedLinear(w.to(dev), b.to(dev)))
    ref = F.linear(x.double(), w.double(), b.double())
    err = (y.cpu().double() - ref).abs().max().item()
    # f16x3 split products: relative error ~2^-22 per product, fp32 accumulation over K
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("N", [128, 256])
def test_linear_fused_epilogue(dev, act, N):
    from isfusion_amd import fusion_ops as ops
    M, K = 1000, 128
    x, w, b = rnd((M, K), 4), rnd((N, K), 5, K ** -0.5), rnd((N,), 6, 0.1)
    res, tab = rnd((M, N), 7), rnd((36, N), 8)
    idx = torch.randint(0, 36, (M,), generator=torch.Generator().manual_seed(9)).int()
    ln = torch.nn.LayerNorm(N)
    ln.weight.data, ln.bias.data = rnd((N,), 10, 0.2) + 1, rnd((N,), 11, 0.1)
    y = ops.linear(x.to(dev), ops.PackedLinear(w.to(dev), b.to(dev)), table=tab.to(dev), index=idx.to(dev), act=act,
                   residual=res.to(dev), ln=ln.to(dev))
    z = F.linear(x, w, b) + tab[idx.long()]
    z = [z, F.relu(z), F.gelu(z)][act]
    ref = F.layer_norm(z + res, (N,), ln.weight.cpu(), ln.bias.cpu(), ln.eps)
    assert (y.cpu() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("B,H,W,K,N", [(2, 18, 18, 256, 256), (1, 180, 180, 128, 128), (3, 6, 10, 64, 384)])
def test_linear_channels_first_io(dev, B, H, W, K, N):
    """GEMM input / residual read from, and output written to, [B, C, H, W] maps (no transpose passes)"""
    from isfusion_amd import fusion_ops as ops
    x, w, b = rnd((B, K, H, W), 12), rnd((N, K), 13, K ** -0.5), rnd((N,), 14, 0.1)
    res = rnd((B, N, H, W), 15)
    ln = torch.nn.LayerNorm(N) if N <= 256 else None
    pl = ops.PackedLinear(w.to(dev), b.to(dev))
    y = ops.linear(x.to(dev), pl, residual=res.to(dev), ln=ln.to(dev) if ln else None, out_nchw=(B, H, W)).cpu()
    z = F.linear(x.permute(0, 2, 3, 1), w, b) + res.permute(0, 2, 3, 1)
    if ln:
        z = F.layer_norm(z, (N,), ln.weight.cpu(), ln.bias.cpu(), ln.eps)
    assert y.shape == (B, N, H, W)
    assert (y - z.permute(0, 3, 1, 2)).abs().max().item() < 2e-5
    y2 = ops.linear(x.to(dev), pl).cpu()                       # channels-first in, token-major out
    assert (y2 - F.linear(x.permute(0, 2, 3, 1), w, b).reshape(-1, N)).abs().max().item() < 2e-5


@pytest.mark.parametrize("M,K,N,flags", [(16200, 256, 768, "table"), (16200, 256, 1024, "relu"), (16200, 256, 256, "gelu"),
                                         (5000, 128, 384, ""), (64800, 128, 256, "relu"), (16137, 256, 768, "")])
def test_linear_column_split_launches_match_fp64(dev, M, K, N, flags):
    """launches with several column chunks and fewer than 1024 row blocks deal (row block, column chunk) pairs to workgroups
    (XCD-aware ids): every output element against float64, incl. a row count that leaves the last group of eight row
    blocks partial"""
    from isfusion_amd import fusion_ops as ops
    x, w, b = rnd((M, K), 71), rnd((N, K), 72, K ** -0.5), rnd((N,), 73, 0.1)
    tab = rnd((36, N), 74)
    idx = torch.randint(0, 36, (M,), generator=torch.Generator().manual_seed(75)).int()
    act = 1 if "relu" in flags else 2 if "gelu" in flags else 0
    kw = dict(act=act)
    if "table" in flags:
        kw["table"], kw["index"] = tab.to(dev), idx.to(dev)
    y = ops.linear(x.to(dev), ops.PackedLinear(w.to(dev), b.to(dev)), **kw).cpu().double()
    z = F.linear(x.double(), w.double(), b.double())
    if "table" in flags:
        z = z + tab[idx.long()].double()
    z = [z, F.relu(z), F.gelu(z)][act]
    assert (y - z).abs().max().item() < 2e-5 * max(1.0, z.abs().max().item())


def test_linear_rejects_bad_shapes(dev):
    from isfusion_amd import fusion_ops as ops
    from isfusion_amd._lib import IsfError
    with pytest.raises(IsfError):
        ops.PackedLinear(torch.zeros(10, 32, device=dev))            # out % 16
    pl = ops.PackedLinear(torch.zeros(512, 128, device=dev))
    ln = torch.nn.LayerNorm(512).to(dev)
    with pytest.raises(IsfError):
        ops.linear(torch.zeros(4, 128, device=dev), pl, ln=ln)       # LN epilogue needs N <= 256


# --------------------------------------------------------------------------------------- window attention
@pytest.mark.parametrize("S,B", [(90, 2), (45, 1), (36, 2), (31, 1), (20, 2), (13, 1), (7, 2), (6, 1), (5, 1), (96, 1), (3, 2),
                                  (54, 1)])
@pytest.mark.parametrize("shift", [0, 1])
def test_window_attention_d256_on_the_matrix_cores_matches_fp64(dev, S, B, shift):
    """the d = 256 level (head_dim 32, 8 heads) of HSF Grid-to-Region: window_attention_mfma_kernel (one wave per (window,
    head), f16x3 split MFMA, K = two 16-wide steps) against float64 per-window softmax attention on EVERY window of twelve
    grids -- whole, partial (edge), shifted and single-window grids, grids smaller than a window"""
    from isfusion_amd import fusion_ops as ops
    from oracle import fusion_ops as orc
    d, hd = 256, 32
    qkv = rnd((B * S * S, 3 * d), 40 + shift + S)
    out = ops.window_attention(qkv.to(dev), B, S, d, 8, 6, shift).cpu().double()
    wid, _, _ = orc.window_geometry(S, 6, shift)
    wid = wid.reshape(-1)
    q, k, v = (qkv[:, i * d:(i + 1) * d].double().view(B, S * S, 8, hd) for i in range(3))
    worst = 0.0
    for w in torch.unique(wid):
        sel = torch.nonzero(wid == w).squeeze(1)
        a = torch.einsum("bihd,bjhd->bhij", q[:, sel], k[:, sel]) * hd ** -0.5
        ref = torch.einsum("bhij,bjhd->bihd", a.softmax(-1), v[:, sel])
        worst = max(worst, (out.view(B, S * S, 8, hd)[:, sel] - ref).abs().max().item())
    assert worst < 2e-6, worst


@pytest.mark.parametrize("S,d,B", [(36, 128, 2), (18, 256, 2), (180, 128, 1), (90, 256, 1), (20, 128, 1)])
@pytest.mark.parametrize("shift", [0, 1])
def test_window_attention_matches_restatement(dev, S, d, B, shift):
    from isfusion_amd import fusion_ops as ops
    from oracle import fusion_ops as orc
    qkv = rnd((B * S * S, 3 * d), 20 + shift)
    out = ops.window_attention(qkv.to(dev), B, S, d, 8, 6, shift).cpu()
    wid, _, _ = orc.window_geometry(S, 6, shift)
    wid = wid.reshape(-1)
    hd = d // 8
    q, k, v = (qkv[:, i * d:(i + 1) * d].view(B, S * S, 8, hd) for i in range(3))
    ref = torch.zeros(B, S * S, 8, hd)
    # masked dense attention in chunks of windows (exact same arithmetic as per-window softmax)
    for w in torch.unique(wid)[:: max(1, len(torch.unique(wid)) // 60)]:   # a spread of windows incl. partial ones
        sel = torch.nonzero(wid == w).squeeze(1)
        a = torch.einsum("bihd,bjhd->bhij", q[:, sel], k[:, sel]) * hd ** -0.5
        ref[:, sel] = torch.einsum("bhij,bjhd->bihd", a.softmax(-1), v[:, sel])
        got = out.view(B, S * S, 8, hd)[:, sel]
        assert (got - ref[:, sel]).abs().max().item() < 1e-5


# --------------------------------------------------------------------------------------- small-key attention
@pytest.mark.parametrize("B,Lq,Lk,E", [(2, 200, 200, 128), (1, 32400, 200, 128), (3, 7, 1, 128), (2, 300, 257, 256)])
def test_attention_matches_fp32(dev, B, Lq, Lk, E):
    from isfusion_amd import fusion_ops as ops
    q, k, v = rnd((B * Lq, E), 30), rnd((B * Lk, E), 31), rnd((B * Lk, E), 32)
    out = ops.attention(q.to(dev), k.to(dev), v.to(dev), B, Lq, Lk, E, 8).cpu()
    hd = E // 8
    qq, kk, vv = q.view(B, Lq, 8, hd), k.view(B, Lk, 8, hd), v.view(B, Lk, 8, hd)
    a = torch.einsum("bihd,bjhd->bhij", qq.double(), kk.double()) * hd ** -0.5
    ref = torch.einsum("bhij,bjhd->bihd", a.softmax(-1), vv.double()).reshape(B * Lq, E)
    assert (out.double() - ref).abs().max().item() < 1e-5


# --------------------------------------------------------------------------------------- channel attention
@pytest.mark.parametrize("B,C,R", [(1, 128, 180), (2, 16, 36), (1, 3, 4), (1, 4, 192)])
def test_channel_attention_matches_fp64(dev, B, C, R):
    from isfusion_amd import fusion_ops as ops
    qs, qi = rnd((B, C, R, R), 40, 0.3), rnd((B, C, R, R), 41, 0.3)
    out = ops.channel_attention(qs.to(dev), qi.to(dev)).cpu()
    a = torch.matmul(qs.double(), qi.double().transpose(2, 3)).softmax(-1)
    ref = qs.double() + torch.matmul(a, qi.double())
    # fp32 MFMA: K = R products of O(0.1) values, softmax weights <= 1
    assert (out.double() - ref).abs().max().item() < 2e-5


# ------------------------------------------------------------------------------------------------------ A8
@pytest.mark.parametrize("name", ["small", "full"])
def test_p2g_matches_restatement_and_golden(dev, golden, name):
    from isfusion_amd import fusion_ops as ops
    from oracle import fusion_ops as orc
    cfg = CONFIGS[name]
    t = torch_inputs(cfg)
    out = ops.p2g_sample(t["pillars"].to(dev), t["pillar_coors"].to(dev), t["img_feats"][1].to(dev), t["lidar2img"],
                         t["img_aug_matrix"], t["lidar_aug_matrix"], t["input_shape"], cfg["B"], cfg["bev"]).cpu()
    check_sample(golden("fusion_ref.npz"), name + ".img_bev", out, 1e-3)
    if name == "small":
        ref = orc.p2g_sample(t["pillars"][..., :3], t["pillar_coors"], t["img_feats"][1], t["lidar2img"],
                             t["img_aug_matrix"], t["lidar_aug_matrix"], t["input_shape"], cfg["B"], cfg["bev"])
        # the only difference is the rounding of the folded camera matrices: sub-pixel shifts of ~1e-5 px times
        # O(1) feature gradients, summed over <= 72 taps
        assert (out - ref).abs().max().item() < 1e-3
        assert torch.equal(out == 0, ref == 0) or ((out == 0) ^ (ref == 0)).sum() < 10


def test_p2g_empty_and_zero_padded_slots(dev):
    from isfusion_amd import fusion_ops as ops
    cfg = CONFIGS["small"]
    t = torch_inputs(cfg)
    out = ops.p2g_sample(t["pillars"][:0].to(dev), t["pillar_coors"][:0].to(dev), t["img_feats"][1].to(dev),
                         t["lidar2img"], t["img_aug_matrix"], t["lidar_aug_matrix"], t["input_shape"], cfg["B"],
                         cfg["bev"])
    assert out.shape == (cfg["B"], 256, cfg["bev"], cfg["bev"]) and float(out.abs().max()) == 0.0


# ----------------------------------------------------------------------------------------------------- A12
@pytest.mark.parametrize("B,H,k,seed", [(1, 180, 200, 0), (4, 180, 200, 1), (2, 36, 20, 2), (1, 16, 1024, 3)])
def test_instance_topk_bit_exact(dev, B, H, k, seed):
    from isfusion_amd import fusion_ops as ops
    from oracle import fusion_ops as orc
    hm = rnd((B, 10, H, H), 50 + seed, 2.0)
    top, raw, masked = ops.instance_topk(hm.to(dev), k, return_masked=True)
    flat, rtop, rraw = orc.instance_topk(hm, k)
    got = masked.cpu()
    # sigmoid may differ by an ulp between the device and host libm; selection must agree wherever the reference's
    # own scores are separated by more than that
    assert (got - flat).abs().max().item() < 1e-6
    assert torch.equal((got > 0), (flat > 0))
    if torch.equal(raw.cpu(), rraw):
        assert torch.equal(top.cpu(), rtop)
    else:
        sv = flat.gather(1, rraw)
        gv = flat.gather(1, raw.cpu())
        assert (sv - gv).abs().max().item() < 1e-6, "selected scores differ beyond sigmoid rounding"


def test_instance_topk_edge_cases(dev):
    from isfusion_amd import fusion_ops as ops
    hm = torch.full((2, 10, 8, 8), -20.0)
    hm[0, 3, 4, 4] = 5.0
    hm[0, 8, 0, 0] = 4.0
    hm[0, 2, 0, 0] = 9.0           # border cell of a 3x3 class: suppressed
    hm[1] = -200.0                 # sigmoid underflows to 0 everywhere: fewer than k positive maxima
    hm[1, 9, 7, 7] = 1.0
    top, raw, masked = ops.instance_topk(hm.to(dev), 3, return_masked=True)
    assert raw[0, :2].tolist() == [3 * 64 + 36, 8 * 64] and top[0, :2].tolist() == [36, 0]
    assert raw[1, 0].item() == 9 * 64 + 63
    assert len(set(raw[1].tolist())) == 3                       # remaining slots: distinct zero-score cells
    assert masked[0, 2 * 64].item() == 0.0
    # equal scores: ascending flat index
    hm2 = torch.full((1, 10, 8, 8), -5.0)
    hm2[0, 8] = 2.0
    _, raw2, _ = ops.instance_topk(hm2.to(dev), 5, return_masked=True)
    assert raw2[0].tolist() == [8 * 64 + i for i in range(5)]


# ----------------------------------------------------------------------------------------------------- A13
@pytest.mark.parametrize("B,Q,H", [(2, 200, 180), (1, 33, 36)])
def test_msda_matches_restatement(dev, B, Q, H):
    from isfusion_amd import fusion_ops as ops
    from oracle import fusion_ops as orc
    value = rnd((B, H * H, 8, 16), 60)
    off = rnd((B * Q, 8 * 16 * 2), 61, 3.0)
    logits = rnd((B * Q, 8 * 16), 62)
    ref_pts = torch.rand((B * Q, 2), generator=torch.Generator().manual_seed(63)) * 1.2 - 0.1   # some outside
    out = ops.msda(value.to(dev), off.to(dev), logits.to(dev), ref_pts.to(dev), B, Q, 8, 16, 16, H, H).cpu()
    loc = ref_pts.view(B, Q, 1, 1, 1, 2) + off.view(B, Q, 8, 1, 16, 2) / torch.tensor([H, H], dtype=torch.float32)
    aw = logits.view(B, Q, 8, 16).softmax(-1).view(B, Q, 8, 1, 16)
    ref = orc.msda_core(value, torch.tensor([[H, H]]), loc, aw).reshape(B * Q, 128)
    # pixel coordinates up to H = 180 carry an fp32 ulp of 1.5e-5: bilinear weights differ by that much between
    # two correctly rounded evaluation orders, times O(1) values
    assert (out - ref).abs().max().item() < 5e-5


# ------------------------------------------------------------------------------- A9 / A15 dense convolutions
@pytest.mark.parametrize("B,cin,cout,H,W,stride", [(2, 128, 128, 36, 36, 1), (1, 768, 128, 180, 180, 1),
                                                  (2, 128, 256, 36, 36, 2), (1, 64, 64, 20, 30, 1),
                                                  (1, 256, 256, 90, 90, 1), (1, 128, 64, 9, 7, 2)])
def test_dense_conv_bn_relu_matches_torch(dev, B, cin, cout, H, W, stride):
    from isfusion_amd.dense_conv import PackedConvBN, SplitMap
    conv = torch.nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
    bn = torch.nn.BatchNorm2d(cout, eps=1e-3).eval()
    conv.weight.data = rnd(conv.weight.shape, 90, (9 * cin) ** -0.5)
    bn.weight.data, bn.bias.data = rnd((cout,), 91, 0.2) + 1, rnd((cout,), 92, 0.1)
    bn.running_mean.data, bn.running_var.data = rnd((cout,), 93, 0.1), rnd((cout,), 94, 0.1).abs() + 0.8
    x = rnd((B, cin, H, W), 95)
    with torch.no_grad():
        ref = F.relu(bn(conv(x.double().float())))
        reft = F.relu(bn(conv(x.permute(0, 1, 3, 2)))).permute(0, 1, 3, 2)
    pc = PackedConvBN(conv.to(dev), bn.to(dev), relu=True)
    xd = x.to(dev)
    maps = [SplitMap.from_nchw(xd, off, min(256, cin - off)) for off in range(0, cin, 256)]
    got = pc(maps).to_nchw().cpu()
    assert got.shape == ref.shape
    # fp32-class arithmetic; MIOpen / oneDNN fp32 differ from each other by this much on K = 9*cin sums
    assert (got - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())
    if stride == 1:
        gott = pc(maps, transpose=True).to_nchw().cpu()
        assert (gott - reft).abs().max().item() < 3e-5 * max(1.0, reft.abs().max().item())


def test_split_map_roundtrip_and_slices(dev):
    from isfusion_amd.dense_conv import SplitMap
    x = rnd((2, 320, 13, 11), 96, 3.0).to(dev)
    a = SplitMap.from_nchw(x, 0, 256).to_nchw()
    b = SplitMap.from_nchw(x, 256, 64).to_nchw()
    got = torch.cat([a, b], 1)
    assert (got - x).abs().max().item() <= 3.0 * 8 * 2.0 ** -21   # hi + lo f16 halves keep 22 bits


def test_secondv2_hip_matches_stock(dev):
    cfg = CONFIGS["small"]
    _, bb = build_modules(cfg, dev)
    x = rnd((2, 128, 36, 36), 97, 0.5).relu().to(dev)
    with torch.no_grad():
        nxt, _, f0 = bb([x], "stage1")
        type(bb).dense_conv = "stock"
        try:
            snxt, _, sf0 = bb([x], "stage1")
            s1 = bb([snxt], "stage2")[2]
        finally:
            type(bb).dense_conv = "hip"
        h1 = bb([snxt], "stage2")[2]
    assert (f0 - sf0).abs().max().item() < 1e-4 and (nxt - snxt).abs().max().item() < 1e-4
    assert (h1 - s1).abs().max().item() < 1e-4


# -------------------------------------------------------------------------------------- module-level parity
@pytest.mark.parametrize("name", ["small", "full"])
def test_grid_to_region_matches_restatement(dev, name):
    """A10/A11: SSTInputLayerV2 + SSTv2 on a dense grid (both levels)"""
    from oracle import fusion_ops as orc
    cfg = CONFIGS[name]
    enc, _ = build_modules(cfg, dev)
    sd, _ = state_dicts(cfg)
    B, S = cfg["B"], cfg["bev"]
    x0 = rnd((B, 128, S, S), 70, 0.5)
    got = enc.grid2region(0, x0.to(dev)).cpu()
    ref = orc.sstv2_forward(x0, sd, "grid2region_att.0")
    assert (got - ref).abs().max().item() < 1e-4
    x1 = rnd((B, 256, S // 2, S // 2), 71, 0.5)
    got = enc.grid2region(1, x1.to(dev)).cpu()
    ref = orc.sstv2_forward(x1, sd, "grid2region_att.1")
    assert (got - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("name,dense", [("small", "hip"), ("full", "hip"), ("small", "stock")])
def test_fusion_encoder_matches_reference_golden(dev, golden, name, dense):
    """whole ISFusionEncoder.forward + SECONDV2 stages against the reference's own outputs; the 3x3 dense convs on the
    f16x3 MFMA kernel (default) or on stock PyTorch-ROCm"""
    g = golden("fusion_ref.npz")
    cfg = CONFIGS[name]
    enc, bb = build_modules(cfg, dev)
    enc.dense_conv = dense
    bb.dense_conv = dense          # instance attribute shadows the class default
    t = torch_inputs(cfg, dev)
    stages = {}
    fuse = enc.fuse
    enc.fuse = lambda a, b: stages.setdefault("bev_feats", fuse(a, b))
    hs = []
    feats, hm = enc(t["img_feats"], t["lidar_feats"], cfg["B"],
                    pts_metas=dict(pillars=t["pillars"], pillar_coors=t["pillar_coors"]),
                    img_metas=[dict(input_shape=t["input_shape"])], pts_backbone=bb, lidar2img=t["lidar2img"],
                    img_aug_matrix=t["img_aug_matrix"], lidar_aug_matrix=t["lidar_aug_matrix"])
    for h in hs:
        h.remove()
    assert [tuple(f.shape) for f in feats] == [(cfg["B"], 128, cfg["bev"], cfg["bev"]),
                                               (cfg["B"], 256, cfg["bev"] // 2, cfg["bev"] // 2)]
    check_sample(g, name + ".bev_feats", stages["bev_feats"], 1e-3)
    check_sample(g, name + ".hm", hm, 1e-3)
    assert np.array_equal(enc.last_top_idx.cpu().numpy(), g[name + ".top_idx"]), "instance indices differ"
    check_sample(g, name + ".feat0", feats[0], 1e-3)   # north_star tolerance on BEV features
    check_sample(g, name + ".feat1", feats[1], 1e-3)


def test_fusion_encoder_batch4_matches_restatement(dev):
    """B = 4 at the nuScenes grid size against the CPU restatement (instance fusion stage, the widest data flow)"""
    from oracle import fusion_ops as orc
    cfg = dict(CONFIGS["full"], B=2, seed=21, num_pillars=3000)
    enc, bb = build_modules(cfg, dev)
    sd, _ = state_dicts(cfg)
    B, S = cfg["B"], cfg["bev"]
    bev_feats, scene = rnd((B, 128, S, S), 80, 0.5).relu(), rnd((B, 128, S, S), 81, 0.5)
    got, hm = enc.instance_fusion(bev_feats.to(dev), scene.to(dev), B)
    ref, rhm, rtop = orc.instance_fusion(bev_feats, scene, sd, B, S, cfg["instance_num"])
    assert (hm.cpu() - rhm).abs().max().item() < 1e-4
    assert torch.equal(enc.last_top_idx.cpu(), rtop)
    assert (got.cpu() - ref).abs().max().item() < 1e-3


# ------------------------------------------------------------------------------------------- detection head (8f #1)
@pytest.mark.parametrize("B,Lq,Lk,E", [(2, 200, 32400, 128), (1, 20, 1296, 128), (1, 7, 513, 256)])
def test_attention_many_keys_matches_fp64(dev, B, Lq, Lk, E):
    """keys split into 512-key chunks + merge (the head's 200 x 32400 cross attention)"""
    from isfusion_amd import fusion_ops as ops
    q, k, v = rnd((B * Lq, E), 33), rnd((B * Lk, E), 34), rnd((B * Lk, E), 35)
    out = ops.attention(q.to(dev), k.to(dev), v.to(dev), B, Lq, Lk, E, 8).cpu()
    hd = E // 8
    qq, kk, vv = q.view(B, Lq, 8, hd), k.view(B, Lk, 8, hd), v.view(B, Lk, 8, hd)
    a = torch.einsum("bihd,bjhd->bhij", qq.double(), kk.double()) * hd ** -0.5
    ref = torch.einsum("bhij,bjhd->bihd", a.softmax(-1), vv.double()).reshape(B * Lq, E)
    assert (out.double() - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize("name,dense", [("small", "hip"), ("full", "hip"), ("small", "stock")])
def test_transfusion_head_matches_reference_golden(dev, golden, name, dense):
    from fusion_common import HEAD_CONFIGS, HEAD_SEED, head_input, head_kwargs
    from isfusion_amd.fusion_modules import seeded_state_dict
    from isfusion_amd.transfusion_head import TransFusionHeadV2
    g = golden("head_ref.npz")
    cfg = HEAD_CONFIGS[name]
    head = TransFusionHeadV2(dense_conv=dense, **head_kwargs(cfg)).eval()
    head.load_state_dict(seeded_state_dict(head, HEAD_SEED))
    head = head.to(dev)
    out = head.forward_single(head_input(cfg).to(dev))[0]
    assert np.array_equal(head.query_labels.cpu().numpy(), g[name + ".labels"]), "proposal classes / order differ"
    for k in ("center", "height", "dim", "rot", "vel", "heatmap", "query_heatmap_score"):
        got = out[k].cpu().numpy()
        assert got.shape == g[f"{name}.{k}"].shape
        assert np.abs(got - g[f"{name}.{k}"]).max() < 1e-3, k      # north_star tolerance (fp32, 1e-3)
    dh = out["dense_heatmap"].reshape(-1).cpu().numpy()
    assert np.abs(dh[g[name + ".dense_heatmap.idx"]] - g[name + ".dense_heatmap.val"]).max() < 1e-3


# ---------------------------------------------------------------------------------- op / module boundary (8b)
@pytest.mark.parametrize("B,Q,M,D,shapes,P", [(2, 50, 8, 16, [(20, 30), (10, 15)], 4), (1, 200, 8, 16, [(180, 180)], 16),
                                              (1, 7, 4, 32, [(9, 5), (5, 3), (3, 2)], 3)])
def test_ms_deform_attn_with_the_mmcv_signature(dev, B, Q, M, D, shapes, P):
    """isf_ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight): several levels,
    any head_dim / point count, locations partly outside the maps -- vs the restatement of the reference kernel"""
    from isfusion_amd import fusion_ops as ops
    from oracle import fusion_ops as orc
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    value = rnd((B, S, M, D), 160)
    g = torch.Generator().manual_seed(161)
    loc = torch.rand((B, Q, M, L, P, 2), generator=g) * 1.3 - 0.15
    aw = torch.rand((B, Q, M, L * P), generator=g).softmax(-1).view(B, Q, M, L, P)
    ss = torch.tensor(shapes, dtype=torch.long)
    ls = torch.cat([ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]])
    out = ops.ms_deform_attn(value.to(dev), ss.to(dev), ls.to(dev), loc.to(dev), aw.to(dev)).cpu()
    ref = orc.msda_core(value, ss, loc, aw)
    assert out.shape == ref.shape == (B, Q, M * D)
    assert (out - ref).abs().max().item() < 5e-5
    # isf_ms_deform_attn_backward (the op's own contract: caller-zeroed gradients) vs autograd through the restatement
    v64, l64, a64 = (t.double().requires_grad_() for t in (value, loc, aw))
    gout = rnd((B, Q, M * D), 162)
    orc.msda_core(v64, ss, l64, a64).backward(gout.double())
    vd, ld, ad = (t.to(dev).requires_grad_() for t in (value, loc, aw))
    ops.ms_deform_attn(vd, ss.to(dev), ls.to(dev), ld, ad).backward(gout.to(dev))
    for got, want in ((vd.grad, v64.grad), (ld.grad, l64.grad), (ad.grad, a64.grad)):
        assert (got.cpu().double() - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("name,B,Q,H", [("small", 1, 20, 12), ("edge", 2, 16, 7)])
def test_ms_deform_attn_backward_op_vs_reference_autograd_golden(dev, golden, name, B, Q, H):
    """isf_ms_deform_attn_backward called as the reference calls ext_module.ms_deform_attn_backward
    (multi_scale_deformable_attn_function.py:150-160) vs tests/golden/msda_grad_ref.npz = autograd through the REFERENCE's
    own pure-torch core (ms_deform_attn_core_pytorch), in the op's own layouts"""
    from fusion_common import msda_grad_inputs
    from isfusion_amd import fusion_ops as ops
    g = golden("msda_grad_ref.npz")
    value, loc, aw, gout = msda_grad_inputs(B, Q, H)
    ss = torch.tensor([[H, H]], dtype=torch.long)
    ls = torch.zeros((1,), dtype=torch.long)
    vd, ld, ad = (t.float().to(dev).requires_grad_() for t in (value, loc, aw))
    out = ops.ms_deform_attn(vd, ss.to(dev), ls.to(dev), ld, ad)
    assert np.abs(out.detach().cpu().numpy() - g[name + ".out"]).max() < 5e-5
    out.backward(gout.float().to(dev))
    for got, key in ((vd.grad, "grad_value"), (ld.grad, "grad_loc"), (ad.grad, "grad_weight")):
        want = g[f"{name}.{key}"]
        assert got.shape == want.shape
        assert np.abs(got.cpu().numpy() - want).max() < 1e-4 * max(1.0, np.abs(want).max()), key


@pytest.mark.parametrize("n,groups,seed", [(1, 1, 0), (5000, 37, 1), (129600, 3844, 2), (777, 5, 3)])
def test_ingroup_indices_first_come_rank(dev, n, groups, seed):
    """isf_ingroup_indices == TorchEx ingroup_indices.forward up to the order inside a group (the reference numbers
    with atomicAdd): every group's numbers are exactly 0..count-1, and here they follow input order"""
    from isfusion_amd import fusion_ops as ops
    rng = np.random.default_rng(seed)
    g = rng.integers(0, groups, n).astype(np.int64)
    if n > 10:
        g[:3] = groups + 5          # a group id above the others; ids need not be dense
    out = ops.ingroup_indices(torch.from_numpy(g).to(dev)).cpu().numpy()
    want = np.zeros(n, np.int64)
    seen = {}
    for i, v in enumerate(g.tolist()):
        want[i] = seen.get(v, 0)
        seen[v] = want[i] + 1
    assert np.array_equal(out, want)
    assert ops.ingroup_indices(torch.zeros((0,), dtype=torch.long, device=dev)).numel() == 0


def test_sst_modules_forward_like_the_reference(dev):
    """SSTInputLayerV2.forward(voxel_feats, voxel_coors, batch_size) -> SSTv2.forward(voxel_info) -> [BEV map]
    (sst_input_layer_v2.py:63, sst_v2.py:65; the call sequence of fusion_encoder.py:1179-1181) on the dense token grid,
    vs the restatement; a sparse token set raises instead of computing something else"""
    from isfusion_amd import _lib
    from oracle import fusion_ops as orc
    cfg = CONFIGS["small"]
    enc, _ = build_modules(cfg, dev)
    sd, _ = state_dicts(cfg)
    B, S = cfg["B"], cfg["bev"]
    x0 = rnd((B, 128, S, S), 170, 0.5)
    feats = x0.permute(0, 2, 3, 1).reshape(B * S * S, 128).to(dev)
    bb, yy, xx = torch.meshgrid(torch.arange(B), torch.arange(S), torch.arange(S), indexing="ij")
    coors = torch.stack([bb, torch.zeros_like(bb), yy, xx], -1).reshape(-1, 4).to(dev)
    info = enc.get_regions[0](feats, coors, B)
    assert info["voxel_coors"].dtype == torch.int64 and info["dense_grid"] == (B, S, S)
    out = enc.grid2region_att[0](info)
    assert isinstance(out, list) and tuple(out[0].shape) == (B, 128, S, S)
    ref = orc.sstv2_forward(x0, sd, "grid2region_att.0")
    assert (out[0].cpu() - ref).abs().max().item() < 1e-4
    with pytest.raises(_lib.IsfError):
        enc.get_regions[0](feats[:-5], coors[:-5], B)                      # not every cell is a token
    with pytest.raises(_lib.IsfError):
        enc.get_regions[0](feats, coors.flip(0), B)                        # not in (b, y, x) order


def test_igf_modules_forward_like_the_reference(dev):
    """MSDeformAttn.forward / InsContextAtt.forward / Instane2SceneAtt.forward with the reference's argument lists
    (fusion_encoder.py:560, :795, :480) agree with the restatements"""
    from isfusion_amd import fusion_ops as ops
    from oracle import fusion_ops as orc
    cfg = CONFIGS["small"]
    enc, _ = build_modules(cfg, dev)
    sd, _ = state_dicts(cfg)
    B, S, E, Q = cfg["B"], cfg["bev"], 128, cfg["instance_num"]
    scene = rnd((B, E, S, S), 180, 0.5)
    x_ins = rnd((B, E, Q), 181, 0.5)
    qpos = torch.rand((B, Q, 2), generator=torch.Generator().manual_seed(182)) * S
    bev_pos = orc.bev_pos_grid(S)
    got = enc.instance_att(x_ins.to(dev), qpos.to(dev), bev_pos.to(dev), scene_feats=scene.to(dev)).cpu()
    ref = orc.ins_context_att(x_ins, qpos, bev_pos, scene, sd, "instance_att", S)
    assert got.shape == ref.shape and (got - ref).abs().max().item() < 2e-4
    # MSDeformAttn alone, two levels (more than the path uses)
    msda = enc.instance_att.layers[0].cross_attn
    shapes = torch.tensor([[S, S]], dtype=torch.long)
    src = rnd((B, S * S, E), 183, 0.5)
    qry = rnd((B, Q, E), 184, 0.5)
    refp = torch.rand((B, Q, 1, 2), generator=torch.Generator().manual_seed(185))
    out, loc, aw = msda(qry.to(dev), refp.to(dev), src.to(dev), shapes.to(dev), torch.zeros((1,), dtype=torch.long, device=dev))
    pre = "instance_att.layers.0.cross_attn."
    lin = lambda t, n: t @ sd[pre + n + ".weight"].t() + sd[pre + n + ".bias"]
    value = lin(src, "value_proj").view(B, S * S, 8, 16)
    off = lin(qry, "sampling_offsets").view(B, Q, 8, 1, msda.n_points, 2)
    aw_ref = lin(qry, "attention_weights").view(B, Q, 8, msda.n_points).softmax(-1).view(B, Q, 8, 1, msda.n_points)
    loc_ref = refp[:, :, None, :, None, :] + off / torch.tensor([S, S], dtype=torch.float32)
    want = lin(orc.msda_core(value, shapes, loc_ref, aw_ref), "output_proj")
    assert (loc.cpu() - loc_ref).abs().max().item() < 1e-4 and (aw.cpu() - aw_ref).abs().max().item() < 1e-5
    assert (out.cpu() - want).abs().max().item() < 2e-4
    # Instane2SceneAtt with the reference's flattened query
    query = rnd((B, E, S, S), 186, 0.5)
    got = enc.instance_to_scene_att(query.flatten(2).to(dev), x_ins.to(dev), scene.to(dev), B, S).cpu()
    ref = orc.instance_to_scene(query.flatten(2), x_ins, scene, sd, "instance_to_scene_att", B, S)
    assert got.shape == ref.shape and (got - ref).abs().max().item() < 1e-3


# ---------------------------------------------------------------------------------- fused window block (A10/A11)
@pytest.mark.parametrize("shift", [0, 1])
@pytest.mark.parametrize("S,d,B", [(12, 128, 2), (9, 128, 1), (16, 128, 2), (13, 128, 1), (180, 128, 1), (37, 128, 3)])
def test_window_block_kernel_matches_unfused_layers(dev, S, d, B, shift):
    """isf_window_block_forward (qkv projection + position table + attention + out-projection + residual + LayerNorm on
    the matrix cores, one kernel) vs the three-launch form it replaces and vs float64 torch: full and partial edge
    windows (grids that are not a multiple of 6, shifted windows); d = 128, the level it is built for"""
    from isfusion_amd import fusion_ops as ops
    from isfusion_amd.fusion_modules import EncoderLayer, seeded_state_dict
    layer = EncoderLayer(d, 8, d).eval()
    layer.load_state_dict(seeded_state_dict(layer, 500 + d + shift))
    x = rnd((B * S * S, d), 501 + S, 0.7)
    # ---- float64 reference of the attention half (sst_basic_block_v2.py:41-75, :104-116)
    index, pos = ops._window_tables(S, 6, shift, d, 1000.0, torch.device("cpu"))
    attn = layer.win_attn.self_attn
    w, b = attn.in_proj_weight.detach().double(), attn.in_proj_bias.detach().double()
    xd = x.double()
    xp = xd + pos.double()[index.long().repeat(B)]
    q, k, v = xp @ w[:d].t() + b[:d], xp @ w[d:2 * d].t() + b[d:2 * d], xd @ w[2 * d:].t() + b[2 * d:]
    off = 3 if shift else 0
    yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
    wid = ((yy + off) // 6) * 64 + (xx + off) // 6
    wid = (torch.arange(B)[:, None, None] * 4096 + wid[None]).reshape(-1)
    hd = d // 8
    att = torch.zeros_like(q)
    for g in torch.unique(wid):
        m = (wid == g).nonzero().squeeze(1)
        qh = q[m].view(-1, 8, hd).transpose(0, 1)
        kh = k[m].view(-1, 8, hd).transpose(0, 1)
        vh = v[m].view(-1, 8, hd).transpose(0, 1)
        p = (qh @ kh.transpose(1, 2) / hd ** 0.5).softmax(-1)
        att[m] = (p @ vh).transpose(0, 1).reshape(-1, d)
    ref = torch.nn.functional.layer_norm(xd + att @ attn.out_proj.weight.detach().double().t() + attn.out_proj.bias.detach().double(),
                                         (d,), layer.norm1.weight.detach().double(), layer.norm1.bias.detach().double(), layer.norm1.eps)
    # ---- HIP: fused and unfused
    layer = layer.to(dev)
    p_ = ops._encoder_layer_cache(layer, S, 6, shift, 1000.0, dev, B)
    xg = x.to(dev)
    fused = ops.window_block(xg, p_["block"], p_["in_bias"], p_["table"], p_["out_bias"], layer.norm1, B, S, d, 8, 6, shift)
    qkv = ops.linear(xg, p_["qkv"], table=p_["table"], index=p_["index"])
    unfused = ops.linear(ops.window_attention(qkv, B, S, d, 8, 6, shift), p_["out"], residual=xg, ln=layer.norm1)
    assert torch.isfinite(fused).all()
    assert (fused.cpu().double() - ref).abs().max().item() < 1e-4
    assert (unfused.cpu().double() - ref).abs().max().item() < 1e-4


def test_p2g_split_output_is_the_fp32_canvas_in_split_rows(dev):
    """isf_p2g_forward_split: Point-to-Grid writing ONE split-format token matrix (what conv_fusion reads) instead of the
    fp32 [B, C, bev, bev] canvas: converted back, the same values bit for bit (hi + lo is exact), zeros where no pillar"""
    from isfusion_amd import fusion_ops as ops
    cfg = CONFIGS["small"]
    t = torch_inputs(cfg, dev)
    args = (t["pillars"], t["pillar_coors"], t["img_feats"][1], t["lidar2img"], t["img_aug_matrix"], t["lidar_aug_matrix"],
            t["input_shape"], cfg["B"], cfg["bev"])
    want = ops.p2g_sample(*args)
    got = ops.p2g_sample(*args, split=True)
    from isfusion_amd.dense_conv import SplitMap
    assert (got.B, got.C, got.H, got.W) == tuple(want.shape)
    assert torch.equal(got.data, SplitMap.from_nchw(want).data)     # the very rows isf_nchw_to_split makes of the canvas
    assert (got.to_nchw() - want).abs().max().item() <= 2.0 ** -21 * want.abs().max().item()   # hi + lo: 22 significant bits
    assert (want != 0).any() and (want == 0).any()
