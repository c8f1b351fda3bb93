"""Inputs, weights and the CPU oracle composition of ISFusionDetector.extract_pts_feat shared by
tests/golden/make_golden_detector.py, tests/test_oracle_fusion.py and the GPU end-to-end tests."""
import numpy as np
import torch

B = 2


def build_path():
    """ISFusionPtsPath with the seeded weights of tests/test_gpu_e2e.py (on the CPU)"""
    from isfusion_amd.detector import ISFusionPtsPath
    from isfusion_amd.fusion_modules import seeded_state_dict
    torch.manual_seed(0)
    net = ISFusionPtsPath().eval()
    net._lidar.randomize_weights_(0).randomize_bn_(1)
    net.fusion_encoder.load_state_dict(seeded_state_dict(net.fusion_encoder, 100))
    net.pts_backbone.load_state_dict(seeded_state_dict(net.pts_backbone, 200))
    net.pts_neck.load_state_dict(seeded_state_dict(net.pts_neck, 250))
    net.pts_bbox_head.load_state_dict(seeded_state_dict(net.pts_bbox_head, 300))
    return net


def detector_inputs(filter_range=True):
    """raw sweeps (range-filtered as the reference pipeline's PointsRangeFilter leaves them), camera features,
    matrices, metas"""
    import oracle
    from isfusion_amd import synthetic
    from oracle import input_ops
    vs, rg = [0.075, 0.075, 0.2], [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]
    pts = []
    for i in range(B):
        p = synthetic.lidar_sweeps(4321 + i, 6000)
        if filter_range:
            p = input_ops.range_filter(p, rg)
            p = p[(oracle.dynamic_voxelize(p, vs, rg) >= 0).all(1)]
        pts.append(np.ascontiguousarray(p))
    inp = synthetic.fusion_inputs(77, B)
    kw = dict(lidar2img=torch.from_numpy(inp["lidar2img"]), img_aug_matrix=torch.from_numpy(inp["img_aug_matrix"]),
              lidar_aug_matrix=torch.from_numpy(inp["lidar_aug_matrix"]))
    metas = [dict(input_shape=inp["input_shape"]) for _ in range(B)]
    return pts, inp, kw, metas


def oracle_extract_pts_feat(net, pts, inp, kw):
    """the composition of the CPU oracles for isfusion.py:103-119 (no neck) -> f0 [B,128,180,180], f1 [B,256,90,90],
    instance heat-map, mined instance cells"""
    import oracle
    from isfusion_amd.norm import fold_bn
    from oracle import fusion_ops as orc
    vs, rg = net.voxel_size, net.pc_range
    coors = np.concatenate([np.concatenate([np.full((p.shape[0], 1), b, np.int32),
                                            oracle.dynamic_voxelize(p, vs, rg)], 1) for b, p in enumerate(pts)])
    vfe = net.pts_voxel_encoder
    bn1 = [t.cpu().numpy() for t in fold_bn(vfe.vfe_layers[0].norm)]
    bn2 = [t.cpu().numpy() for t in fold_bn(vfe.vfe_layers[1].norm)]
    vf, vc, _ = oracle.dynamic_vfe(np.concatenate(pts), coors, vs, rg,
                                   vfe.vfe_layers[0].linear.weight.detach().cpu().numpy(), bn1,
                                   vfe.vfe_layers[1].linear.weight.detach().cpu().numpy(), bn2)
    bev, _ = oracle.sparse_encoder_forward(net.pts_middle_encoder.plan_to_numpy(), vf, vc, B)
    pil, pco = [], []
    for b, p in enumerate(pts):
        v, c, n = oracle.hard_voxelize(p, net.pillar_size, rg, 12, 60000)
        pil.append(v)
        pco.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
    pillars, pcoors = torch.from_numpy(np.concatenate(pil)), torch.from_numpy(np.concatenate(pco))
    sd = {k: v.float().cpu() for k, v in net.fusion_encoder.state_dict().items()}
    sdb = {"bb." + k: v.float().cpu() for k, v in net.pts_backbone.state_dict().items()}
    with torch.no_grad():
        img_bev = orc.p2g_sample(pillars[..., :3], pcoors, torch.from_numpy(inp["img_feats"][1]), kw["lidar2img"],
                                 kw["img_aug_matrix"], kw["lidar_aug_matrix"], inp["input_shape"], B, 180)
        bev_feats = orc.conv_module(torch.cat([img_bev, torch.from_numpy(bev)], 1), sd, "conv_fusion")
        g0 = orc.sstv2_forward(bev_feats, sd, "grid2region_att.0")
        ret, rhm, rtop = orc.instance_fusion(bev_feats, g0, sd, B, 180, 200)
        nxt, f0 = orc.secondv2_stage(ret, sdb, "bb", "stage1")
        g1 = orc.sstv2_forward(nxt, sd, "grid2region_att.1")
        _, f1 = orc.secondv2_stage(g1, sdb, "bb", "stage2")
    return f0, f1, rhm, rtop
