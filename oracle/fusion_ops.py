"""CPU restatement (torch, fp32 unless noted) of the HSF / IGF rows of SURVEY.md section 8 -- TEST INFRASTRUCTURE ONLY.

Each function cites the reference file:line it follows (paths relative to the reference root).  Pinned by
golden vectors produced by the reference's own Python code (tests/golden/make_golden_fusion.py imports
mmdet3d/models/middle_encoders/fusion_encoder.py, models/sst/*, models/backbones/{sst_v2,second}.py through
tests/golden/ref_harness.py) -> tests/golden/fusion_ref.npz.

State dicts use the reference's parameter names (e.g.
``grid2region_att.0.block_list.0.encoder_list.0.win_attn.self_attn.in_proj_weight``).
"""
import math

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------ helpers
def conv_module(x, sd, prefix, eps=1e-5):
    """mmcv ConvModule = Conv2d(3x3, pad 1, no bias) + BN2d(eval) + ReLU  (fusion_encoder.py:862-869 etc.)."""
    w = sd[prefix + ".conv.weight"]
    pad = w.shape[-1] // 2
    y = F.conv2d(x, w, None, 1, pad)
    y = F.batch_norm(y, sd[prefix + ".bn.running_mean"], sd[prefix + ".bn.running_var"], sd[prefix + ".bn.weight"],
                     sd[prefix + ".bn.bias"], False, 0.0, eps)
    return F.relu(y)


def layer_norm(x, sd, prefix, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def mha(q_in, k_in, v_in, sd, prefix, nhead, key_padding_mask=None):
    """Standard multi-head attention on [L, N, E] tensors (nn.MultiheadAttention semantics; the custom copy in
    fusion_encoder.py:221-470 is the same arithmetic): in_proj, scale q by head_dim^-0.5, softmax(QK^T) V,
    out_proj.  key_padding_mask [N, S] True = ignore."""
    L, N, E = q_in.shape
    S = k_in.shape[0]
    hd = E // nhead
    w, b = sd[prefix + ".in_proj_weight"], sd[prefix + ".in_proj_bias"]
    q = F.linear(q_in, w[:E], b[:E]) * (hd ** -0.5)
    k = F.linear(k_in, w[E:2 * E], b[E:2 * E])
    v = F.linear(v_in, w[2 * E:], b[2 * E:])
    q = q.reshape(L, N * nhead, hd).transpose(0, 1)
    k = k.reshape(S, N * nhead, hd).transpose(0, 1)
    v = v.reshape(S, N * nhead, hd).transpose(0, 1)
    att = torch.bmm(q, k.transpose(1, 2))  # [N*h, L, S]
    if key_padding_mask is not None:
        att = att.view(N, nhead, L, S).masked_fill(key_padding_mask[:, None, None, :], float("-inf")).view(N * nhead, L, S)
    att = torch.softmax(att, dim=-1)
    out = torch.bmm(att, v).transpose(0, 1).reshape(L, N, E)
    return F.linear(out, sd[prefix + ".out_proj.weight"], sd[prefix + ".out_proj.bias"])


# ------------------------------------------------------------------------------------------------ A10 / A11
def window_geometry(S, win, shift):
    """Dense S x S grid, 2-D windows (sst_ops.py:219-268 with sparse_z == win_z => no z shift):
    shift 0 adds `win` (aligned windows), shift 1 adds win//2.
    -> (window id [S,S] (row-major over (wx, wy) like batch_win_inds), in-window (y, x) [S,S,2], windows/side)."""
    off = win // 2 if shift else win
    c = torch.arange(S) + off
    w1, i1 = c // win, c % win
    nside = int(math.ceil(S / win) + 1)
    # coors[:, 3] is "x" (last dim), coors[:, 2] is "y": win id = wx * nside + wy
    wy, wx = torch.meshgrid(w1, w1, indexing="ij")
    iy, ix = torch.meshgrid(i1, i1, indexing="ij")
    return wx * nside + wy, torch.stack([iy, ix], -1), nside


def sst_pos_embed(in_win_yx, win, d, temperature=1000.0):
    """sst_input_layer_v2.py:224-290 (2-D, normalize_pos False): x then y, interleaved sin/cos."""
    y = in_win_yx[..., 0].float() - win / 2
    x = in_win_yx[..., 1].float() - win / 2
    pos_length = d // 2
    inv = torch.arange(pos_length, dtype=torch.float32)
    inv = temperature ** (2 * (inv // 2) / pos_length)
    ex = x[..., None] / inv
    ey = y[..., None] / inv
    ex = torch.stack([ex[..., ::2].sin(), ex[..., 1::2].cos()], dim=-1).flatten(-2)
    ey = torch.stack([ey[..., ::2].sin(), ey[..., 1::2].cos()], dim=-1).flatten(-2)
    return torch.cat([ex, ey], dim=-1)


def encoder_layer_window(x, sd, prefix, S, win, shift, nhead=8):
    """EncoderLayer (post-norm) over a dense [B, S, S, d] token grid: window MHA with q = k = x + pos, v = x
    (sst_basic_block_v2.py:41-75), +residual, LayerNorm, FFN (Linear, GELU, Linear), +residual, LayerNorm
    (:104-126).  Tokens of one window attend to each other only (partial edge windows of the shifted layer
    simply hold fewer tokens: the reference masks the empty slots, :293-303)."""
    B, _, _, d = x.shape
    wid, inwin, nside = window_geometry(S, win, shift)
    pos = sst_pos_embed(inwin, win, d).to(x.dtype)
    tok = x.reshape(B, S * S, d)
    out = torch.zeros_like(tok)
    widf = wid.reshape(-1)
    posf = pos.reshape(S * S, d)
    for w in torch.unique(widf):
        sel = torch.nonzero(widf == w).squeeze(1)
        xt = tok[:, sel].transpose(0, 1)  # [T, B, d]
        qk = xt + posf[sel][:, None, :]
        out[:, sel] = mha(qk, qk, xt, sd, prefix + ".win_attn.self_attn", nhead).transpose(0, 1)
    y = layer_norm(tok + out, sd, prefix + ".norm1")
    f = F.linear(F.gelu(F.linear(y, sd[prefix + ".linear1.weight"], sd[prefix + ".linear1.bias"])),
                 sd[prefix + ".linear2.weight"], sd[prefix + ".linear2.bias"])
    y = layer_norm(y + f, sd, prefix + ".norm2")
    return y.reshape(B, S, S, d)


def sstv2_forward(bev, sd, prefix, win=6):
    """get_regions[i] + grid2region_att[i] on a dense grid: bev [B, C, S, S] -> [B, d, S, S]
    (sst_v2.py:65-133; token n = y*S + x, fusion_encoder.py:1167-1173)."""
    B, C, S, _ = bev.shape
    x = bev.permute(0, 2, 3, 1)
    if prefix + ".linear0.weight" in sd:
        x = F.linear(x, sd[prefix + ".linear0.weight"], sd[prefix + ".linear0.bias"])
    for i in range(2):
        x = encoder_layer_window(x, sd, f"{prefix}.block_list.0.encoder_list.{i}", S, win, i)
    return x.permute(0, 3, 1, 2).contiguous()


# ------------------------------------------------------------------------------------------------ A8
def p2g_sample(pillars_xyz, pillar_coors, img_feat, lidar2img, img_aug, lidar_aug, input_shape, bs, bev, num_cam=6):
    """img_fv_to_bev + img_point_sampling (fusion_encoder.py:965-1070).
    pillars_xyz [M, T, 3] (zero-padded slots are sampled like real points, :1049,1067-1068), pillar_coors [M,4]
    (b, z, y, x), img_feat [bs*num_cam, C, H, W] -> [bs, C, bev, bev]."""
    C, H, W = img_feat.shape[1:]
    out = torch.zeros((bs, C, bev, bev), dtype=img_feat.dtype)
    feat = img_feat.view(bs, num_cam, C, H, W)
    for b in range(bs):
        sel = pillar_coors[:, 0] == b
        pts = pillars_xyz[sel].reshape(-1, 3).clone()
        T = pillars_xyz.shape[1]
        cur = pts - lidar_aug[b][:3, 3]
        cur = torch.inverse(lidar_aug[b][:3, :3]).matmul(cur.transpose(1, 0))                     # [3, N]
        cur = lidar2img[b][:, :3, :3].matmul(cur) + lidar2img[b][:, :3, 3].reshape(-1, 3, 1)       # [cam, 3, N]
        cur[:, 2, :] = torch.clamp(cur[:, 2, :], 1e-5, 1e5)
        cur[:, :2, :] = cur[:, :2, :] / cur[:, 2:3, :]
        cur = img_aug[b][:, :3, :3].matmul(cur) + img_aug[b][:, :3, 3].reshape(-1, 3, 1)
        uv = cur[:, :2, :].transpose(1, 2).clone()                                                   # [cam, N, 2]
        uv[..., 0] = uv[..., 0] / input_shape[1]
        uv[..., 1] = uv[..., 1] / input_shape[0]
        uv = (uv - 0.5) * 2
        acc = torch.zeros((C, pts.shape[0]), dtype=img_feat.dtype)
        for k in range(num_cam):
            s = F.grid_sample(feat[b, k][None], uv[k].view(1, -1, 1, 2), mode="bilinear", padding_mode="zeros",
                              align_corners=False)
            acc += s.view(C, -1)
        pc = pillar_coors[sel]
        out[b][:, pc[:, 2].long(), pc[:, 3].long()] = acc.view(C, -1, T).sum(dim=2)
    return out


# ------------------------------------------------------------------------------------------------ A12
def bev_pos_grid(S):
    """create_2D_grid (fusion_encoder.py:901-913): [1, S*S, 2] cell centres (i + .5, j + .5)."""
    g = torch.linspace(0, S - 1, S) + 0.5
    bx, by = torch.meshgrid(g, g, indexing="ij")
    return torch.stack([bx, by], 0).view(1, 2, -1).permute(0, 2, 1)


def instance_topk(ins_heatmap, instance_num=200, nms_kernel=3, pool1_classes=(8, 9)):
    """sigmoid -> 3x3 valid max-pool written into the interior (border stays 0) -> classes 8, 9 use a 1x1 pool ->
    keep local maxima -> top-k over all classes, index modulo H*W (fusion_encoder.py:1100-1131).
    -> (masked heatmap [B, K*H*W], top index mod HW [B, k], raw flat top index [B, k])."""
    B, K, H, W = ins_heatmap.shape
    heat = ins_heatmap.sigmoid()
    pad = nms_kernel // 2
    local_max = torch.zeros_like(heat)
    local_max[:, :, pad:-pad, pad:-pad] = F.max_pool2d(heat, nms_kernel, 1, 0)
    for c in pool1_classes:
        local_max[:, c] = heat[:, c]
    heat = heat * (heat == local_max)
    flat = heat.view(B, -1)
    top = flat.argsort(dim=-1, descending=True)[..., :instance_num]
    return flat, top % (H * W), top


# ------------------------------------------------------------------------------------------------ A13
def msda_core(value, spatial_shapes, sampling_locations, attention_weights):
    """Multi-scale deformable attention forward following the CUDA kernel of the reference
    (ops/src/cuda/ms_deform_im2col_cuda.cuh:33-84 bilinear with zero padding, :237-299 accumulation):
    value [B, S, Hh, D]; loc [B, Q, Hh, L, P, 2] in [0,1] (x, y); weights [B, Q, Hh, L, P] -> [B, Q, Hh*D]."""
    B, S, Hh, D = value.shape
    _, Q, _, L, P, _ = sampling_locations.shape
    out = torch.zeros((B, Q, Hh, D), dtype=value.dtype)
    start = 0
    for lvl in range(L):
        H, W = int(spatial_shapes[lvl][0]), int(spatial_shapes[lvl][1])
        v = value[:, start:start + H * W].view(B, H, W, Hh, D)
        start += H * W
        loc = sampling_locations[:, :, :, lvl]                     # [B,Q,Hh,P,2]
        w_im = loc[..., 0] * W - 0.5
        h_im = loc[..., 1] * H - 0.5
        h0 = torch.floor(h_im)
        w0 = torch.floor(w_im)
        lh, lw = h_im - h0, w_im - w0
        valid = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
        acc = torch.zeros((B, Q, Hh, P, D), dtype=value.dtype)
        bi = torch.arange(B).view(B, 1, 1, 1).expand(B, Q, Hh, P)
        hi = torch.arange(Hh).view(1, 1, Hh, 1).expand(B, Q, Hh, P)
        for dh, dw, wgt in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw), (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
            hh = (h0 + dh).long()
            ww = (w0 + dw).long()
            ok = valid & (hh >= 0) & (hh < H) & (ww >= 0) & (ww < W)
            g = v[bi, hh.clamp(0, H - 1), ww.clamp(0, W - 1), hi]   # [B,Q,Hh,P,D]
            acc += g * (wgt * ok)[..., None]
        out += (acc * attention_weights[:, :, :, lvl][..., None]).sum(dim=3)
    return out.reshape(B, Q, Hh * D)


def pos_embed_learned(xy, sd, prefix, eps=1e-5):
    """PositionEmbeddingLearned: Conv1d(2->E) + BN1d(eval) + ReLU + Conv1d(E->E) on [B, N, 2] -> [B, N, E]
    (fusion_encoder.py:173-189)."""
    x = xy.transpose(1, 2)
    h = F.conv1d(x, sd[prefix + ".position_embedding_head.0.weight"], sd[prefix + ".position_embedding_head.0.bias"])
    h = F.batch_norm(h, sd[prefix + ".position_embedding_head.1.running_mean"],
                     sd[prefix + ".position_embedding_head.1.running_var"],
                     sd[prefix + ".position_embedding_head.1.weight"], sd[prefix + ".position_embedding_head.1.bias"],
                     False, 0.0, eps)
    h = F.conv1d(F.relu(h), sd[prefix + ".position_embedding_head.3.weight"],
                 sd[prefix + ".position_embedding_head.3.bias"])
    return h.transpose(1, 2)


def ins_context_att(x_ins, query_pos, bev_pos, x_scene, sd, prefix, bev_size, num_layers=2, nhead=8, n_points=16):
    """InsContextAtt.forward (fusion_encoder.py:795-830) with DeformableTransformerDecoderLayer (:653-674) and
    MSDeformAttn (:560-600), eval mode (dropout off).  x_ins [B, E, Q], query_pos [B, Q, 2], bev_pos [B, HW, 2],
    x_scene [B, E, H, W] -> [B, E, Q]."""
    B, E, Q = x_ins.shape
    scene = x_scene.permute(0, 1, 3, 2)
    key_pos = pos_embed_learned(bev_pos / bev_size, sd, prefix + ".key_pos_embed")       # [B, HW, E]
    h, w = scene.shape[2:]
    src = scene.flatten(2).transpose(1, 2) + key_pos
    out = x_ins.transpose(1, 2)
    ref = query_pos / bev_size
    qpe = pos_embed_learned(ref, sd, prefix + ".query_pos_embed")                        # [B, Q, E]
    shapes = torch.tensor([[h, w]])
    hd = E // nhead
    for l in range(num_layers):
        p = f"{prefix}.layers.{l}"
        qk = (out + qpe).transpose(0, 1)
        t2 = mha(qk, qk, out.transpose(0, 1), sd, p + ".self_attn", nhead).transpose(0, 1)
        out = layer_norm(out + t2, sd, p + ".norm2")
        q = out + qpe
        value = F.linear(src, sd[p + ".cross_attn.value_proj.weight"], sd[p + ".cross_attn.value_proj.bias"])
        value = value.view(B, h * w, nhead, hd)
        off = F.linear(q, sd[p + ".cross_attn.sampling_offsets.weight"], sd[p + ".cross_attn.sampling_offsets.bias"])
        off = off.view(B, Q, nhead, 1, n_points, 2)
        aw = F.linear(q, sd[p + ".cross_attn.attention_weights.weight"], sd[p + ".cross_attn.attention_weights.bias"])
        aw = torch.softmax(aw.view(B, Q, nhead, n_points), -1).view(B, Q, nhead, 1, n_points)
        norm = torch.tensor([w, h], dtype=off.dtype)
        loc = ref[:, :, None, None, None, :] + off / norm
        t2 = msda_core(value, shapes, loc, aw)
        t2 = F.linear(t2, sd[p + ".cross_attn.output_proj.weight"], sd[p + ".cross_attn.output_proj.bias"])
        out = layer_norm(out + t2, sd, p + ".norm1")
        f = F.linear(F.relu(F.linear(out, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])),
                     sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
        out = layer_norm(out + f, sd, p + ".norm3")
    return out.transpose(1, 2)


# ------------------------------------------------------------------------------------------------ A14
def instance_to_scene(query, key, query_scene, sd, prefix, bs, bev_size, nhead=8):
    """Instane2SceneAtt.forward (fusion_encoder.py:480-502), eval mode: query [B, E, HW] = conv_ins(bev) flattened,
    key = x_ins [B, E, Q], query_scene = G2R output [B, E, H, W]."""
    q = query.permute(2, 0, 1)
    k = key.permute(2, 0, 1)
    q2 = mha(q, k, k, sd, prefix + ".multihead_attn", nhead)
    q = layer_norm(q + q2, sd, prefix + ".norm").permute(1, 2, 0)
    query_ins = q.reshape(bs, q.shape[1], bev_size, bev_size)
    att = torch.softmax(torch.matmul(query_scene, query_ins.transpose(2, 3)), dim=-1)
    return query_scene + torch.matmul(att, query_ins)


def instance_fusion(bev_feats, scene_feats, sd, bs, bev_size, instance_num=200):
    """ISFusionEncoder.instance_fusion (fusion_encoder.py:1090-1149) -> (features, ins_heatmap, top indices)."""
    bev_pos = bev_pos_grid(bev_size).repeat(bs, 1, 1)
    out = bev_feats.permute(0, 1, 3, 2).contiguous()
    hm = conv_module(out, sd, "conv_heatmap")
    hm = conv_module(hm, sd, "heatmap_head_1")
    hm = conv_module(hm, sd, "heatmap_head_2")
    hm = F.conv2d(hm, sd["heatmap_head_3.weight"], sd["heatmap_head_3.bias"], 1, 1)
    _, top_idx, _ = instance_topk(hm, instance_num)
    query_pos = bev_pos.gather(1, top_idx[:, :, None].expand(-1, -1, 2))
    query_pos_new = torch.stack([query_pos[..., 1], query_pos[..., 0]], -1)
    x_scene = conv_module(bev_feats.permute(0, 1, 3, 2), sd, "conv_scene")
    x_ins = x_scene.reshape(bs, x_scene.shape[1], -1).gather(2, top_idx[:, None, :].expand(-1, x_scene.shape[1], -1))
    x_ins = ins_context_att(x_ins, query_pos_new, bev_pos, x_scene, sd, "instance_att", bev_size)
    q = conv_module(bev_feats, sd, "conv_ins").flatten(2, 3)
    ret = instance_to_scene(q, x_ins, scene_feats, sd, "instance_to_scene_att", bs, bev_size)
    return ret, hm, top_idx


# ------------------------------------------------------------------------------------------------ A15 (stock convs)
def secondv2_stage(x, sd, prefix, stage, eps=1e-3):
    """SECONDV2.forward(x, stage) (backbones/second.py:200-230): stage1 = blocks[0] on x[0] then ds_layer;
    stage2 = blocks[1].  Plain Conv2d(bias=False)+BN(eps 1e-3)+ReLU stacks."""
    def seq(x, p, n):
        for i in range(n):
            w = sd[f"{p}.{3 * i}.weight"]
            stride = 1
            x = F.conv2d(x, w, None, stride, 1)
            x = F.batch_norm(x, sd[f"{p}.{3 * i + 1}.running_mean"], sd[f"{p}.{3 * i + 1}.running_var"],
                             sd[f"{p}.{3 * i + 1}.weight"], sd[f"{p}.{3 * i + 1}.bias"], False, 0.0, eps)
            x = F.relu(x)
        return x
    n0 = sum(1 for k in sd if k.startswith(prefix + ".blocks.0.") and k.endswith(".weight") and sd[k].dim() == 4)
    n1 = sum(1 for k in sd if k.startswith(prefix + ".blocks.1.") and k.endswith(".weight") and sd[k].dim() == 4)
    if stage == "stage1":
        feat = seq(x, prefix + ".blocks.0", n0)
        y = F.conv2d(feat, sd[prefix + ".ds_layer.0.weight"], None, 2, 1)
        y = F.batch_norm(y, sd[prefix + ".ds_layer.1.running_mean"], sd[prefix + ".ds_layer.1.running_var"],
                         sd[prefix + ".ds_layer.1.weight"], sd[prefix + ".ds_layer.1.bias"], False, 0.0, eps)
        return F.relu(y), feat
    return None, seq(x, prefix + ".blocks.1", n1)


# ------------------------------------------------------------------------------------------------ 8f #1 (head)
def transfusion_head_forward(inputs, sd, num_proposals=200, num_classes=10, nhead=8, nms_kernel=3,
                             pool1_classes=(8, 9), heads=("center", "height", "dim", "rot", "vel", "heatmap")):
    """TransFusionHeadV2.forward_single with one decoder layer, eval mode
    (dense_heads/transfusion_head_v2.py:771-892, decoder layer :80-120, FFN :561-590)."""
    B, _, X, Y = inputs.shape
    HW = X * Y
    feat = F.conv2d(inputs, sd["shared_conv.weight"], sd["shared_conv.bias"], 1, 1)
    E = feat.shape[1]
    flat = feat.view(B, E, HW)
    g = torch.linspace(0, X - 1, X) + 0.5
    g2 = torch.linspace(0, Y - 1, Y) + 0.5
    bx, by = torch.meshgrid(g, g2, indexing="ij")
    bev_pos = torch.stack([bx, by], 0).view(1, 2, -1).permute(0, 2, 1).repeat(B, 1, 1)
    hm = conv_module(feat, sd, "heatmap_head.0")
    dense_heatmap = F.conv2d(hm, sd["heatmap_head.1.weight"], sd["heatmap_head.1.bias"], 1, 1)
    masked, top_idx, top_raw = instance_topk(dense_heatmap, num_proposals, nms_kernel, pool1_classes)
    top_class = top_raw // HW
    query_pos = bev_pos.gather(1, top_idx[:, :, None].expand(-1, -1, 2))
    query = flat.gather(2, top_idx[:, None, :].expand(-1, E, -1))                         # [B, E, P]
    one_hot = F.one_hot(top_class, num_classes=num_classes).permute(0, 2, 1).float()
    query = query + F.conv1d(one_hot, sd["class_encoding.weight"], sd["class_encoding.bias"])
    p = "decoder.0"
    qpe = pos_embed_learned(query_pos, sd, p + ".self_posembed").permute(1, 0, 2)          # [P, B, E]
    kpe = pos_embed_learned(bev_pos, sd, p + ".cross_posembed").permute(1, 0, 2)           # [HW, B, E]
    q = query.permute(2, 0, 1)
    k = flat.permute(2, 0, 1)
    x = q + qpe
    q = layer_norm(q + mha(x, x, x, sd, p + ".self_attn", nhead), sd, p + ".norm1")
    kk = k + kpe
    q = layer_norm(q + mha(q + qpe, kk, kk, sd, p + ".multihead_attn", nhead), sd, p + ".norm2")
    f = F.linear(F.relu(F.linear(q, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])),
                 sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
    q = layer_norm(q + f, sd, p + ".norm3")
    qf = q.permute(1, 2, 0)                                                                # [B, E, P]
    out = {}
    for h in heads:
        hp = f"prediction_heads.0.{h}"
        y = F.conv1d(qf, sd[hp + ".0.conv.weight"])
        y = F.relu(F.batch_norm(y, sd[hp + ".0.bn.running_mean"], sd[hp + ".0.bn.running_var"], sd[hp + ".0.bn.weight"],
                                sd[hp + ".0.bn.bias"], False, 0.0, 1e-5))
        out[h] = F.conv1d(y, sd[hp + ".1.weight"], sd[hp + ".1.bias"])
    out["center"] = out["center"] + query_pos.permute(0, 2, 1)
    out["query_heatmap_score"] = masked.view(B, num_classes, HW).gather(2, top_idx[:, None, :].expand(-1, num_classes, -1))
    out["dense_heatmap"] = dense_heatmap
    out["top_idx"] = top_idx
    return out


def decode_boxes(preds, query_labels, num_proposals, out_size_factor, voxel_size, pc_range, post_center_range,
                 score_threshold=0.0, num_classes=10):
    """TransFusionHeadV2.get_bboxes with nms_type=None (dense_heads/transfusion_head_v2.py:1286-1312,1344-1418) around
    TransFusionBBoxCoder.decode(filter=True) (core/bbox/coders/transfusion_bbox_coder.py:39-124); the inputs are not
    modified (the reference decodes centre / dim in place).  -> per sample (boxes [n, 7|9], scores [n], labels [n])."""
    P = num_proposals
    score = preds["heatmap"][..., -P:].sigmoid()
    one_hot = F.one_hot(query_labels, num_classes=num_classes).permute(0, 2, 1)
    score = score * preds["query_heatmap_score"] * one_hot
    final_scores, final_preds = score.max(1)
    center = preds["center"][..., -P:].clone()
    center[:, 0] = center[:, 0] * out_size_factor * voxel_size[0] + pc_range[0]
    center[:, 1] = center[:, 1] * out_size_factor * voxel_size[1] + pc_range[1]
    dim = preds["dim"][..., -P:].exp()
    height = preds["height"][..., -P:] - dim[:, 2:3] * 0.5
    rot = preds["rot"][..., -P:]
    yaw = torch.atan2(rot[:, 0:1], rot[:, 1:2])
    parts = [center, height, dim, yaw] + ([preds["vel"][..., -P:]] if "vel" in preds else [])
    boxes = torch.cat(parts, 1).permute(0, 2, 1)
    rng = torch.tensor(post_center_range, dtype=boxes.dtype)
    mask = (boxes[..., :3] >= rng[:3]).all(2) & (boxes[..., :3] <= rng[3:]).all(2)
    if score_threshold:          # :108 -- a threshold of 0.0 is not applied
        mask &= final_scores > score_threshold
    return [(boxes[i, mask[i]], final_scores[i, mask[i]], final_preds[i, mask[i]]) for i in range(boxes.shape[0])]
