"""CPU tests: the restatement of the HSF / IGF rows (oracle/fusion_ops.py) against the golden vectors the
reference's own Python produced (tests/golden/make_golden_fusion.py -> fusion_ref.npz)."""
import numpy as np
import pytest
import torch

from fusion_common import CONFIGS, check_sample, state_dicts, torch_inputs


def oracle_forward(cfg):
    """ISFusionEncoder.forward + SECONDV2 stages through the restatement -> dict of stage outputs"""
    from oracle import fusion_ops as orc
    sd, sdb = state_dicts(cfg)
    t = torch_inputs(cfg)
    B, bev = cfg["B"], cfg["bev"]
    out = {}
    with torch.no_grad():
        out["img_bev"] = orc.p2g_sample(t["pillars"][..., :3], t["pillar_coors"], t["img_feats"][1], t["lidar2img"],
                                        t["img_aug_matrix"], t["lidar_aug_matrix"], t["input_shape"], B, bev)
        out["bev_feats"] = orc.conv_module(torch.cat([out["img_bev"], t["lidar_feats"]], 1), sd, "conv_fusion")
        out["g2r0"] = orc.sstv2_forward(out["bev_feats"], sd, "grid2region_att.0")
        out["ins_fusion"], out["hm"], out["top_idx"] = orc.instance_fusion(out["bev_feats"], out["g2r0"], sd, B, bev,
                                                                            cfg["instance_num"])
        nxt, out["feat0"] = orc.secondv2_stage(out["ins_fusion"], sdb, "bb", "stage1")
        out["g2r1"] = orc.sstv2_forward(nxt, sd, "grid2region_att.1")
        _, out["feat1"] = orc.secondv2_stage(out["g2r1"], sdb, "bb", "stage2")
    return out


@pytest.mark.parametrize("name", ["small", "full"])
def test_restatement_matches_reference_outputs(golden, name):
    g = golden("fusion_ref.npz")
    out = oracle_forward(CONFIGS[name])
    assert np.array_equal(out["top_idx"].numpy(), g[name + ".top_idx"])
    for key, tol in (("img_bev", 1e-5), ("bev_feats", 1e-4), ("g2r0", 2e-4), ("hm", 1e-4), ("ins_fusion", 3e-4),
                     ("feat0", 5e-4), ("g2r1", 5e-4), ("feat1", 1e-3)):
        check_sample(g, f"{name}.{key}", out[key], tol)


def test_window_geometry_matches_reference_window_count():
    """sst_ops.get_window_coors: shift 0 -> ceil(S/6) windows per side hold cells, shift 1 -> one more"""
    from oracle import fusion_ops as orc
    for S, n0, n1 in ((180, 30, 31), (90, 15, 16), (36, 6, 7)):
        w0, _, _ = orc.window_geometry(S, 6, 0)
        w1, _, _ = orc.window_geometry(S, 6, 1)
        assert len(torch.unique(w0)) == n0 * n0 and len(torch.unique(w1)) == n1 * n1


def test_instance_topk_edge_cases():
    from oracle import fusion_ops as orc
    hm = torch.full((1, 10, 8, 8), -20.0)
    hm[0, 3, 4, 4] = 5.0
    hm[0, 8, 0, 0] = 4.0   # class 8 uses a 1x1 pool: border cells are candidates there
    hm[0, 2, 0, 0] = 9.0   # 3x3 classes never keep border cells
    flat, top, raw = orc.instance_topk(hm, 2)
    assert raw[0].tolist() == [3 * 64 + 36, 8 * 64] and top[0].tolist() == [36, 0]
    assert flat[0, 2 * 64] == 0


# ------------------------------------------------------------------------------------------- detection head
HEAD_KEYS = ("center", "height", "dim", "rot", "vel", "heatmap", "query_heatmap_score")


@pytest.mark.parametrize("name", ["small", "full"])
def test_head_restatement_matches_reference_outputs(golden, name):
    """TransFusionHeadV2.forward_single: restatement vs the reference's own outputs"""
    from fusion_common import HEAD_CONFIGS, HEAD_SEED, head_input, head_kwargs
    from isfusion_amd.fusion_modules import seeded_state_dict
    from isfusion_amd.transfusion_head import TransFusionHeadV2
    from oracle import fusion_ops as orc
    g = golden("head_ref.npz")
    cfg = HEAD_CONFIGS[name]
    sd = {k: v.float() for k, v in seeded_state_dict(TransFusionHeadV2(**head_kwargs(cfg)), HEAD_SEED).items()}
    with torch.no_grad():
        o = orc.transfusion_head_forward(head_input(cfg), sd, cfg["num_proposals"])
    assert np.array_equal(o["top_idx"].numpy(), g[name + ".top_idx"])
    for k in HEAD_KEYS:
        assert np.abs(o[k].numpy() - g[f"{name}.{k}"]).max() < 2e-4, k
    dh = o["dense_heatmap"].reshape(-1).numpy()
    assert np.abs(dh[g[name + ".dense_heatmap.idx"]] - g[name + ".dense_heatmap.val"]).max() < 1e-4


def _golden_head_outputs(g, name):
    out = {k: torch.from_numpy(g[f"{name}.{k}"]) for k in HEAD_KEYS}
    return out, torch.from_numpy(g[name + ".labels"])


@pytest.mark.parametrize("coder", ["shipped", "tight"])
@pytest.mark.parametrize("name", ["small", "full"])
def test_decode_boxes_restatement_matches_reference_get_bboxes(golden, name, coder):
    """TransFusionHeadV2.get_bboxes (nms_type=None) + TransFusionBBoxCoder.decode: restatement vs the reference's own
    boxes on the reference's head outputs; the 'tight' coder makes the centre-range / score filter bite"""
    from fusion_common import HEAD_CODERS, HEAD_CONFIGS
    from oracle import fusion_ops as orc
    g = golden("head_ref.npz")
    cfg, c = HEAD_CONFIGS[name], HEAD_CODERS[coder]
    out, labels = _golden_head_outputs(g, name)
    dec = orc.decode_boxes(out, labels, cfg["num_proposals"], c["out_size_factor"], c["voxel_size"], c["pc_range"],
                           c["post_center_range"], c["score_threshold"])
    assert len(dec) == cfg["B"]
    kept = 0
    for i, (boxes, scores, labs) in enumerate(dec):
        ref = g[f"{name}.{coder}.{i}.boxes"]
        assert boxes.shape == ref.shape and boxes.shape[1] == 9
        assert np.array_equal(labs.numpy(), g[f"{name}.{coder}.{i}.box_labels"])
        assert np.array_equal(scores.numpy(), g[f"{name}.{coder}.{i}.scores"])
        if ref.size:
            assert np.abs(boxes.numpy() - ref).max() < 1e-5
        kept += ref.shape[0]
    if coder == "tight":
        assert 0 < kept < cfg["B"] * cfg["num_proposals"], "the filter must drop some and keep some"
    # the inputs are left untouched (the reference decodes in place)
    assert np.array_equal(out["center"].numpy(), g[name + ".center"])


@pytest.mark.parametrize("name,B,Q,H", [("small", 1, 20, 12), ("edge", 2, 16, 7)])
def test_msda_backward_restatement_matches_reference_autograd(golden, name, B, Q, H):
    """gradients of the multi-scale deformable attention: autograd through the restatement (CUDA
    kernel's bilinear rule) vs autograd through the reference's pure-torch implementation (float64 goldens)"""
    from fusion_common import msda_grad_inputs
    from oracle import fusion_ops as orc
    g = golden("msda_grad_ref.npz")
    value, loc, aw, gout = msda_grad_inputs(B, Q, H)
    v, l, a = value.clone().requires_grad_(), loc.clone().requires_grad_(), aw.clone().requires_grad_()
    out = orc.msda_core(v, torch.tensor([[H, H]]), l, a).reshape(B, Q, -1)
    out.backward(gout)
    assert np.abs(out.detach().numpy() - g[name + ".out"]).max() < 1e-5
    for got, key in ((v.grad, "grad_value"), (l.grad, "grad_loc"), (a.grad, "grad_weight")):
        ref = g[f"{name}.{key}"]
        assert np.abs(got.numpy() - ref).max() < 1e-4 * max(1.0, np.abs(ref).max()), key


def test_extract_pts_feat_oracle_composition_matches_reference_detector(golden, oracle_mod):
    """the composition of the CPU oracles (what the GPU end-to-end test compares against) vs the REFERENCE's own
    ISFusionDetector.extract_pts_feat run on the CPU: reference detector code + reference sub-modules over the oracle's
    compiled-op restatements (tests/golden/make_golden_detector.py) -- pins the glue between the pinned pieces"""
    from detector_common import build_path, detector_inputs, oracle_extract_pts_feat
    g = golden("detector_ref.npz")
    net = build_path()
    pts, inp, kw, _ = detector_inputs()
    f0, f1, _, _ = oracle_extract_pts_feat(net, pts, inp, kw)
    net.pts_neck.dense_conv = "stock"      # CPU tensors: the torch modules (the default path is the HIP linear kernel)
    with torch.no_grad():
        out = net.pts_neck([f0, f1])[0]
    assert list(out.shape) == g["shape"].tolist()
    flat = out.numpy().reshape(-1)
    assert np.abs(flat[g["idx"]] - g["val"]).max() < 1e-4
    assert abs(np.abs(flat).mean() - g["mean_abs"][0]) < 1e-5
