"""TransFusionHeadV2.forward_single (mmdet3d/models/dense_heads/transfusion_head_v2.py:771-892):
BEV feature map [B, 512, 180, 180] -> shared conv -> heat-map -> sigmoid + 3x3 NMS + top-200 proposals (the same fused
kernel as the encoder's instance mining) -> one transformer decoder layer (200-query self-attention, 200 x 32400
cross-attention over the BEV map with learned position embeddings, FFN) -> per-proposal regression / class heads.

Parameter / sub-module names equal the reference's (96 state-dict keys; checked with ``load_state_dict(strict=True)``
into the reference class in tests/golden/make_golden_head.py).  Inference: ``forward`` / ``forward_split`` and the box
decoding ``get_bboxes`` (isf_decode_boxes, below) are built and pinned by reference goldens; losses and target assignment
(``loss``, ``get_targets``) are training control plane around this path and are not.

HIP path: the two 3x3 convs on the f16x3 sparse-conv kernel over the dense grid, top-k through ``isf_instance_topk``,
all Linear layers through ``isf_linear_forward`` (the key / value projection of the 32400 x B BEV tokens folds the
learned key position embedding into a per-cell table), attention through ``isf_attention_forward`` (keys split into
512-key chunks + merge).  The 2 -> 128 position MLPs, the 128 -> 10 heat-map conv and the Conv1d prediction heads on 200
proposals stay stock torch ops (tiny, channel counts below the MFMA tile).
"""
import copy

import torch
from torch import nn

from . import fusion_ops as ops
from .dense_conv import PackedConvBN, SplitMap
from .fusion_modules import PositionEmbeddingLearned, _SelfAttn
from .spconv import from_split


class _ConvModule1d(nn.Sequential):
    """mmcv ConvModule(conv_cfg=Conv1d, norm_cfg=BN1d, bias='auto'): conv (no bias) / bn / activate"""

    def __init__(self, cin, cout):
        super().__init__()
        self.add_module("conv", nn.Conv1d(cin, cout, 1, bias=False))
        self.add_module("bn", nn.BatchNorm1d(cout))
        self.add_module("activate", nn.ReLU(inplace=True))


class _ConvModule2d(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__()
        self.add_module("conv", nn.Conv2d(cin, cout, 3, padding=1, bias=False))
        self.add_module("bn", nn.BatchNorm2d(cout))
        self.add_module("activate", nn.ReLU(inplace=True))


class FFN(nn.Module):
    """transfusion_head_v2.py:505-590: one Conv1d stack per output (`center`, `height`, ...)"""

    def __init__(self, in_channels, heads, head_conv=64):
        super().__init__()
        self.heads = heads
        for head, (classes, num_conv) in heads.items():
            layers, c = [], in_channels
            for _ in range(num_conv - 1):
                layers.append(_ConvModule1d(c, head_conv))
                c = head_conv
            layers.append(nn.Conv1d(c, classes, 1))
            setattr(self, head, nn.Sequential(*layers))

    def forward(self, x):
        return {head: getattr(self, head)(x) for head in self.heads}


class TransformerDecoderLayer(nn.Module):
    """transfusion_head_v2.py:42-120 parameter layout"""

    def __init__(self, d_model, nhead, dim_feedforward):
        super().__init__()
        self.nhead = nhead
        self.self_attn = _SelfAttn(d_model)
        self.multihead_attn = _SelfAttn(d_model)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.self_posembed = PositionEmbeddingLearned(2, d_model)
        self.cross_posembed = PositionEmbeddingLearned(2, d_model)


class TransFusionHeadV2(nn.Module):

    def __init__(self, num_proposals=200, auxiliary=True, in_channels=512, hidden_channel=128, num_classes=10,
                 num_decoder_layers=1, num_heads=8, nms_kernel_size=3, ffn_channel=256, common_heads=None,
                 num_heatmap_convs=2, test_cfg=None, bbox_coder=None, dense_conv="hip", **kwargs):
        super().__init__()
        self.num_classes, self.num_proposals, self.auxiliary = num_classes, num_proposals, auxiliary
        self.num_heads, self.num_decoder_layers, self.nms_kernel_size = num_heads, num_decoder_layers, nms_kernel_size
        self.test_cfg = test_cfg or dict(dataset="nuScenes", grid_size=[1440, 1440, 40], out_size_factor=8)
        self.dense_conv = dense_conv
        # TransFusionBBoxCoder arguments (configs/isfusion/isfusion_0075voxel.py:130-138); decoding runs in one kernel
        self.bbox_coder = dict(pc_range=[-54.0, -54.0], voxel_size=[0.075, 0.075], out_size_factor=8,
                               post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], score_threshold=0.0,
                               code_size=10)
        self.bbox_coder.update({k: v for k, v in (bbox_coder or {}).items() if k != "type"})
        common_heads = common_heads or dict(center=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2))
        self.shared_conv = nn.Conv2d(in_channels, hidden_channel, 3, padding=1, bias=True)
        self.heatmap_head = nn.Sequential(_ConvModule2d(hidden_channel, hidden_channel),
                                          nn.Conv2d(hidden_channel, num_classes, 3, padding=1, bias=True))
        self.class_encoding = nn.Conv1d(num_classes, hidden_channel, 1)
        self.decoder = nn.ModuleList([TransformerDecoderLayer(hidden_channel, num_heads, ffn_channel)
                                      for _ in range(num_decoder_layers)])
        self.prediction_heads = nn.ModuleList()
        for _ in range(num_decoder_layers):
            heads = copy.deepcopy(common_heads)
            heads.update(dict(heatmap=(num_classes, num_heatmap_convs)))
            self.prediction_heads.append(FFN(hidden_channel, heads))
        self.x_size = self.test_cfg["grid_size"][0] // self.test_cfg["out_size_factor"]
        self.y_size = self.test_cfg["grid_size"][1] // self.test_cfg["out_size_factor"]

    # ------------------------------------------------------------------------------------------------ pieces
    def _bev_pos(self, device):
        """create_2D_grid (:728-738): [1, X*Y, 2] cell centres, cell n = i*Y + j -> (i + .5, j + .5)"""
        gx = torch.linspace(0, self.x_size - 1, self.x_size, device=device) + 0.5
        gy = torch.linspace(0, self.y_size - 1, self.y_size, device=device) + 0.5
        bx, by = torch.meshgrid(gx, gy, indexing="ij")
        return torch.stack([bx, by], 0).view(1, 2, -1).permute(0, 2, 1)

    def _packed(self, device):
        c = ops._cache(self, device)                    # dropped when any parameter / buffer of the head changes
        if "shared" not in c:
            c["shared"] = PackedConvBN(self.shared_conv, None, relu=False)
            c["hm0"] = PackedConvBN(self.heatmap_head[0].conv, self.heatmap_head[0].bn, relu=True)
            c["hm1"] = PackedConvBN(self.heatmap_head[1], None, relu=False)   # 128 -> 10 (columns zero-padded to 32)
            bev_pos = self._bev_pos(device)
            c["bev_pos"] = bev_pos
            c["bev_pos_tok"] = bev_pos[0].contiguous()                            # [HW, 2] for isf_head_query_init
            c["layers"] = []
            E = self.shared_conv.out_channels
            for l in self.decoder:
                sa, ca = l.self_attn, l.multihead_attn
                ws, bs = sa.in_proj_weight.detach().float(), sa.in_proj_bias.detach().float()
                wc, bc = ca.in_proj_weight.detach().float(), ca.in_proj_bias.detach().float()
                # key = value = feat + key_pos_embed (:104-106): (feat + pos) W + b = feat W + (pos W + b), the second
                # term is input independent -> per-cell table added in the GEMM epilogue
                kpe = ops._pos_embed(l.cross_posembed, bev_pos)[0]                    # [HW, E]
                # the proposals sit on cell centres (query_pos = bev_pos[top_index]): their self-attention position
                # embedding is a row of this per-cell table (same MLP, same inputs) -- no GEMMs per forward
                qtab = ops._pos_embed(l.self_posembed, bev_pos)[0].contiguous()
                table = (kpe.double() @ wc[E:].double().t()).float().contiguous()     # [HW, 2E]
                c["layers"].append(dict(
                    s_qkv=ops.PackedLinear(ws, bs), s_out=ops.PackedLinear(sa.out_proj.weight, sa.out_proj.bias),
                    c_q=ops.PackedLinear(wc[:E], bc[:E]), c_kv=ops.PackedLinear(wc[E:], bc[E:]), kv_table=table,
                    c_out=ops.PackedLinear(ca.out_proj.weight, ca.out_proj.bias),
                    l1=ops.PackedLinear(l.linear1.weight, l.linear1.bias),
                    l2=ops.PackedLinear(l.linear2.weight, l.linear2.bias), qpe_table=qtab))
            c["pred"] = [self._pack_prediction_heads(ffn) for ffn in self.prediction_heads]
        return c

    @staticmethod
    def _pack_prediction_heads(ffn):
        """The per-output Conv1d stacks of an FFN (transfusion_head_v2.py:505-590; kernel size 1 = per-proposal linear
        layers) as GEMM pairs: the first layers of a GROUP of outputs side by side (eval BatchNorm folded, ReLU in the
        epilogue), their last layers as one block-diagonal matrix.  Groups of 4 / 2 / 1 outputs keep the hidden width
        (64 each) inside what the linear kernel tiles (<= 256): the shipped six outputs are 4 + 2 = four launches for 12
        convolutions + 6 BatchNorms + 6 ReLUs.  -> None when a stack is not the shipped (conv-bn-relu, conv) shape or
        width; the caller then runs the modules."""
        heads = []
        for name in ffn.heads:
            seq = getattr(ffn, name)
            if len(seq) != 2 or not isinstance(seq[0], _ConvModule1d) or not isinstance(seq[1], nn.Conv1d):
                return None
            conv, bn, last = seq[0].conv, seq[0].bn, seq[1]
            if conv.weight.shape[0] != 64 or conv.weight.shape[1] not in (32, 64, 128, 256):
                return None
            scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().float()
            heads.append((name, conv.weight.detach().float()[:, :, 0] * scale[:, None],
                          (bn.bias - bn.running_mean * scale).detach().float(), last.weight.detach().float()[:, :, 0],
                          last.bias.detach().float()))
        groups, i = [], 0
        while i < len(heads):
            n = 4 if len(heads) - i >= 4 else 2 if len(heads) - i >= 2 else 1
            grp = heads[i:i + n]
            i += n
            hid = 64 * n
            n_out = sum(h[3].shape[0] for h in grp)
            w2 = grp[0][1].new_zeros(((n_out + 15) // 16 * 16, hid))      # the linear kernel writes 16-column tiles
            b2 = grp[0][1].new_zeros((w2.shape[0],))
            cols, r = [], 0
            for k, (name, _, _, w, b) in enumerate(grp):
                w2[r:r + w.shape[0], 64 * k:64 * (k + 1)] = w
                b2[r:r + w.shape[0]] = b
                cols.append((name, r, r + w.shape[0]))
                r += w.shape[0]
            groups.append(dict(l1=ops.PackedLinear(torch.cat([h[1] for h in grp], 0), torch.cat([h[2] for h in grp], 0)),
                               l2=ops.PackedLinear(w2, b2), cols=cols))
        return groups

    @torch.no_grad()
    def forward_split(self, maps):
        """engine-level entry (ISFusionPtsPath.forward_pts): `maps` = SECONDFPN.forward_split(...) -- the neck output as
        split-format token matrices of the UN-permuted BEV map M [B, C, H, W] (the reference hands over
        P = M.permute(0, 1, 3, 2)).  Same result as forward_single(P): the 3x3 convolutions run with transposed taps on
        M's tokens and only the 10-channel heat-map is permuted."""
        assert self.dense_conv == "hip", "forward_split is the HIP path"
        return [self.forward_single(None, maps_yx=maps)[0]]

    @torch.no_grad()
    def forward_single(self, inputs, img_inputs=None, metas=None, maps_yx=None):
        """inputs [B, in_channels, X, Y] -> [dict(center, height, dim, rot, vel, heatmap, query_heatmap_score,
        dense_heatmap)] (one dict: num_decoder_layers results concatenated along the proposal axis when auxiliary)"""
        assert not self.training, "isfusion_amd.TransFusionHeadV2 is the inference path (eval mode)"
        E = self.shared_conv.out_channels
        tok_of_cell = None      # BEV cell x*Y + y (the head's indexing) -> row of feat_tok; None = identity
        if maps_yx is not None:
            m0 = maps_yx[0]
            B, X, Y = m0.B, m0.W, m0.H                                             # the head's X is M's W axis
            HW = X * Y
            dev = m0.data.device
            c = self._packed(dev)
            feat = c["shared"](maps_yx, transpose=True)                            # SplitMap of M: token (b, y, x)
            hm_mid = c["hm0"](feat, transpose=True)                                # SplitMap [B, E, H, W]
            feat_tok = from_split(feat.data, (B * HW, E))
            dense_heatmap = c["hm1"](hm_mid, transpose=True).to_nchw(self.heatmap_head[1].out_channels)
            dense_heatmap = dense_heatmap.permute(0, 1, 3, 2).contiguous()         # [B, classes, X, Y]
            key = ("tok_of_cell", X, Y)
            if key not in c:
                cell = torch.arange(HW, device=dev)
                c[key] = (cell % Y) * X + torch.div(cell, Y, rounding_mode="floor")           # (x, y) -> y*W + x
                c[("cell_of_tok", X, Y)] = ((cell % X) * Y + torch.div(cell, X, rounding_mode="floor")).int()
            tok_of_cell = c[key]
        else:
            B, _, X, Y = inputs.shape
            HW = X * Y
            dev = inputs.device
            c = self._packed(dev)
            if self.dense_conv == "hip":
                maps = [SplitMap.from_nchw(inputs, off, min(256, inputs.size(1) - off))
                        for off in range(0, inputs.size(1), 256)]
                feat = c["shared"](maps)                                          # SplitMap [B, E, X, Y]
                dense_heatmap = c["hm1"](c["hm0"](feat)).to_nchw(self.heatmap_head[1].out_channels).contiguous()
                feat_tok = from_split(feat.data, (B * HW, E))                     # token-major fp32
            else:
                lidar_feat = self.shared_conv(inputs)
                dense_heatmap = self.heatmap_head[1](self.heatmap_head[0](lidar_feat))
                feat_tok = ops.to_tokens(lidar_feat)
        pool1 = (8, 9) if self.test_cfg["dataset"] == "nuScenes" else (1, 2)
        top32, raw32, masked = ops.instance_topk(dense_heatmap, self.num_proposals, self.nms_kernel_size, pool1,
                                                 return_masked=True, as_int32=True)
        # labels, positions, query = feature column + class_encoding(one_hot) (= column `class` of the 1x1 conv + bias),
        # the first layer's position embedding (the proposals sit on cell centres: a row of the per-cell table), query +
        # position and the proposals' heat-map scores: one launch (isf_head_query_init)
        ce = self.class_encoding
        ctab = c.get("class_table")
        if ctab is None:
            ctab = c["class_table"] = (ce.weight[:, :, 0].t() + ce.bias).detach().float().contiguous()
        query, qpe0, x0, query_pos, top_index, top_class, q_score = ops.head_query_init(
            top32, raw32, feat_tok, tok_of_cell, ctab, c["layers"][0]["qpe_table"], c["bev_pos_tok"], masked,
            self.num_classes)
        self.query_labels = top_class
        self.last_top_index = top_index   # BEV cell of every proposal (for inspection)
        P = self.num_proposals
        ret_dicts = []
        for i, (l, p) in enumerate(zip(self.decoder, c["layers"])):
            if i == 0:   # = self_posembed(query_pos): the proposals sit on cell centres; later layers use refined centres
                qpe, x = qpe0, x0
            else:
                qpe = ops._pos_embed(l.self_posembed, query_pos).reshape(B * P, E)
                x = query + qpe
            # self attention: q = k = v = query + pos (:98-101)
            qkv = ops.linear(x, p["s_qkv"])
            att = ops.attention(qkv, qkv[:, E:], qkv[:, 2 * E:], B, P, P, E, l.nhead, ldkv=3 * E)
            query = ops.linear(att, p["s_out"], residual=query, ln=l.norm1)
            # cross attention over the BEV map (:104-108)
            qc = ops.linear(query + qpe, p["c_q"])
            if tok_of_cell is None:
                idx = c.setdefault(("kv_index", B, HW), torch.arange(HW, device=dev, dtype=torch.int32).repeat(B))
            else:   # key token (b, y, x) takes the position row of its cell x*Y + y
                idx = c.setdefault(("kv_index_yx", B, X, Y), c[("cell_of_tok", X, Y)].repeat(B))
            kv = ops.linear(feat_tok, p["c_kv"], table=p["kv_table"], index=idx)                     # [B*HW, 2E]
            att = ops.attention(qc, kv, kv[:, E:], B, P, HW, E, l.nhead, ldkv=2 * E)
            query = ops.linear(att, p["c_out"], residual=query, ln=l.norm2)
            h = ops.linear(query, p["l1"], act=ops.ACT_RELU)
            query = ops.linear(h, p["l2"], residual=query, ln=l.norm3)
            pp = c["pred"][i] if self.dense_conv == "hip" else None
            if pp is not None:
                blocks = []
                for grp in pp:
                    o = ops.linear(ops.linear(query, grp["l1"], act=ops.ACT_RELU), grp["l2"])       # [B*P, columns]
                    blocks += [(n, o, a, b - a) for n, a, b in grp["cols"]]
                # every output as [B, c, P], center += query_pos, the next layer's positions: one launch
                res, query_pos = ops.head_scatter_predictions(blocks, B, P, query_pos.contiguous())
                res = {n: res[n] for n in self.prediction_heads[i].heads}      # the reference's key order
            else:
                res = self.prediction_heads[i](query.view(B, P, E).transpose(1, 2).contiguous())       # [B, E, P]
                res["center"] = res["center"] + query_pos.permute(0, 2, 1)
                query_pos = res["center"].detach().clone().permute(0, 2, 1)
            ret_dicts.append(res)
        ret_dicts[0]["query_heatmap_score"] = q_score
        ret_dicts[0]["dense_heatmap"] = dense_heatmap
        if not self.auxiliary:
            return [ret_dicts[-1]]
        new_res = {}
        for key in ret_dicts[0].keys():
            if key not in ("dense_heatmap", "dense_heatmap_old", "query_heatmap_score"):
                # (one decoder layer, the shipped head: the tensor itself -- torch.cat of one tensor is a copy launch)
                new_res[key] = torch.cat([r[key] for r in ret_dicts], dim=-1) if len(ret_dicts) > 1 else ret_dicts[0][key]
            else:
                new_res[key] = ret_dicts[0][key]
        return [new_res]

    def forward(self, feats, img_feats=None, metas=None):
        """:894-908: one level"""
        if isinstance(feats, torch.Tensor):
            feats = [feats]
        # multi_apply transposes the per-level lists: a 1-tuple holding the list of per-level result dicts
        return ([self.forward_single(f, None, metas)[0] for f in feats],)

    @torch.no_grad()
    def get_bboxes(self, preds_dicts, metas=None, img=None, rescale=False, for_roi=False):
        """:1278-1418 with test_cfg nms_type=None (the shipped nuScenes setting): proposal scores, box decoding and the
        centre-range / score filter in one kernel (isf_decode_boxes).  -> one [boxes, scores, labels] per sample
        (the reference asserts a single sample, :1407-1408); boxes are wrapped in metas[i]['box_type_3d'] when the meta
        carries one."""
        if self.test_cfg.get("nms_type") is not None:
            raise NotImplementedError("get_bboxes: only nms_type=None (the shipped test_cfg) is built; circle / rotate "
                                      "NMS is the TTA path (SURVEY.md section 8, out of scope)")
        assert len(preds_dicts) == 1, "one feature level"
        pd = preds_dicts[0][0]
        P, bc = self.num_proposals, self.bbox_coder
        last = {k: pd[k][..., -P:] for k in ("heatmap", "center", "height", "dim", "rot")}
        vel = pd["vel"][..., -P:] if "vel" in pd else None
        cell = [bc["out_size_factor"] * bc["voxel_size"][0], bc["out_size_factor"] * bc["voxel_size"][1]]
        boxes, scores, labels, counts = ops.decode_boxes(
            last["heatmap"], pd["query_heatmap_score"], self.query_labels, last["center"], last["height"], last["dim"],
            last["rot"], vel, cell, bc["pc_range"], bc["post_center_range"], bc["score_threshold"])
        res = []
        for i, n in enumerate(counts.tolist()):
            b = boxes[i, :n]
            box_type = (metas[i] if metas and i < len(metas) else {}).get("box_type_3d")
            res.append([box_type(b, box_dim=b.shape[-1]) if box_type is not None else b, scores[i, :n], labels[i, :n]])
        return res
