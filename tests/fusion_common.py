"""Shared setup of the HSF / IGF tests: the configurations the golden file was generated for, seeded weights,
seeded inputs (all regenerated here; tests/golden/fusion_ref.npz holds only the reference's outputs)."""
import numpy as np
import torch

from isfusion_amd import synthetic
from isfusion_amd.fusion_encoder import ISFusionEncoder
from isfusion_amd.fusion_modules import SECONDV2, seeded_state_dict

CONFIGS = {
    "small": dict(seed=11, B=2, bev=36, image_hw=(96, 160), num_pillars=300, instance_num=20),
    "full": dict(seed=12, B=1, bev=180, image_hw=(384, 1056), num_pillars=6000, instance_num=200),
}
BACKBONE_KW = dict(in_channels=128, out_channels=[128, 256], layer_nums=[5, 5], layer_strides=[1, 2],
                   norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01), conv_cfg=dict(type="Conv2d", bias=False))
ENC_SEED, BB_SEED = 100, 200


def encoder_kwargs(cfg):
    bev = cfg["bev"]
    return dict(num_points_in_pillar=10, embed_dims=256, bev_size=bev, num_views=6,
                region_shape=[(6, 6, 1), (6, 6, 1)], grid_size=[[bev, bev, 1], [bev // 2, bev // 2, 1]],
                region_drop_info=[{0: {"max_tokens": 36, "drop_range": (0, 100000)}},
                                  {0: {"max_tokens": 36, "drop_range": (0, 100000)}}],
                instance_num=cfg["instance_num"])


def torch_inputs(cfg, device="cpu"):
    inp = synthetic.fusion_inputs(cfg["seed"], cfg["B"], bev_size=cfg["bev"], num_pillars=cfg["num_pillars"],
                                  image_hw=cfg["image_hw"])
    t = dict(input_shape=inp["input_shape"])
    t["img_feats"] = tuple(torch.from_numpy(a).to(device) for a in inp["img_feats"])
    for k in ("lidar_feats", "pillars", "pillar_coors", "lidar2img", "img_aug_matrix", "lidar_aug_matrix"):
        t[k] = torch.from_numpy(inp[k]).to(device)
    return t


def build_modules(cfg, device="cpu"):
    torch.manual_seed(0)
    enc = ISFusionEncoder(**encoder_kwargs(cfg)).eval()
    bb = SECONDV2(**BACKBONE_KW).eval()
    enc.load_state_dict(seeded_state_dict(enc, ENC_SEED))
    bb.load_state_dict(seeded_state_dict(bb, BB_SEED))
    return enc.to(device), bb.to(device)


def state_dicts(cfg):
    enc, bb = build_modules(cfg)
    sd = {k: v.float() for k, v in enc.state_dict().items()}
    sdb = {"bb." + k: v.float() for k, v in bb.state_dict().items()}
    return sd, sdb


def check_sample(g, key, x, atol, rtol=0.0):
    """compare tensor x with the golden sample `key` (.idx/.val/.stat)"""
    flat = x.detach().float().cpu().contiguous().view(-1).numpy()
    idx, val, stat = g[key + ".idx"], g[key + ".val"], g[key + ".stat"]
    assert flat.size == int(stat[2]), (key, flat.size, stat[2])
    err = np.abs(flat[idx] - val)
    tol = atol + rtol * np.abs(val)
    assert (err <= tol).all(), f"{key}: max err {err.max():.3e} (tol {atol}), at value {val[err.argmax()]:.4f}"
    assert abs(flat.mean() - stat[0]) <= atol and abs(np.abs(flat).mean() - stat[1]) <= atol, key
    return float(err.max())


# ------------------------------------------------------------------------------------------- detection head
HEAD_SEED = 300
HEAD_CONFIGS = {
    "small": dict(seed=41, B=2, X=36, num_proposals=20),
    "full": dict(seed=42, B=1, X=180, num_proposals=200),
}
# TransFusionBBoxCoder arguments for the get_bboxes goldens: the shipped ones (isfusion_0075voxel.py:130-138) and a
# variant whose centre range / score threshold actually filter the seeded proposals
HEAD_CODERS = {
    "shipped": dict(pc_range=[-54.0, -54.0], voxel_size=[0.075, 0.075], out_size_factor=8,
                    post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], score_threshold=0.0, code_size=10),
    "tight": dict(pc_range=[-54.0, -54.0], voxel_size=[0.075, 0.075], out_size_factor=8,
                  post_center_range=[-50.0, -52.0, -1.5, -35.0, 30.0, 0.8], score_threshold=0.33, code_size=10),
}


def head_kwargs(cfg):
    return dict(num_proposals=cfg["num_proposals"], auxiliary=True, in_channels=512, hidden_channel=128, num_classes=10,
                num_decoder_layers=1, num_heads=8, nms_kernel_size=3, ffn_channel=256,
                common_heads=dict(center=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2)),
                test_cfg=dict(dataset="nuScenes", grid_size=[cfg["X"] * 8, cfg["X"] * 8, 40], out_size_factor=8))


def head_input(cfg):
    g = torch.Generator().manual_seed(cfg["seed"])
    return torch.randn((cfg["B"], 512, cfg["X"], cfg["X"]), generator=g) * 0.5


# ------------------------------------------------------------------------------------------- MSDA gradients (8f #2)
def msda_grad_inputs(B, Q, H, heads=8, hd=16, P=16, raw=False):
    """value [B, H*H, heads, hd], sampling locations [B, Q, heads, 1, P, 2] (some outside [0, 1]), attention weights
    [B, Q, heads, 1, P] (softmax of seeded logits), grad_out [B, Q, heads*hd] -- float32, seeded."""
    g = torch.Generator().manual_seed(1000 + 31 * B + 7 * Q + H)
    value = torch.randn((B, H * H, heads, hd), generator=g)
    ref = torch.rand((B, Q, 1, 1, 1, 2), generator=g) * 1.2 - 0.1
    off = torch.randn((B, Q, heads, 1, P, 2), generator=g) * 2.0
    loc = ref + off / H
    logits = torch.randn((B, Q, heads, P), generator=g)
    aw = logits.softmax(-1).view(B, Q, heads, 1, P)
    gout = torch.randn((B, Q, heads * hd), generator=g)
    if raw:     # the kernel's parameterisation: reference points, pixel offsets, pre-softmax logits
        return value, loc, aw, gout, ref.reshape(B * Q, 2), off.reshape(B * Q, heads * P * 2), \
            logits.reshape(B * Q, heads * P)
    return value, loc, aw, gout
