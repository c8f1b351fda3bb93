"""ISFusionEncoder (drop-in for mmdet3d/models/middle_encoders/fusion_encoder.py:833-1189): HSF Point-to-Grid
(A8), Grid-to-Region window attention (A10/A11), IGF instance mining (A12), instance context attention with
multi-scale deformable attention (A13) and instance-to-scene attention (A14).

Same constructor kwargs, sub-module / parameter names and ``forward`` signature
    forward(img_mlvl_feats, lidar_feats, bs, **kwargs{pts_metas, img_metas, pts_backbone, lidar2img,
            img_aug_matrix, lidar_aug_matrix}) -> ([f1 [B,128,S,S], f2 [B,256,S/2,S/2]], ins_hm [B,10,S,S])
Custom arithmetic runs in libisf_hip.so (``fusion_ops``).  The 3x3 dense convolutions (conv_fusion, heatmap head,
conv_scene, conv_ins) run on the sparse encoder's f16x3 MFMA kernel over the dense grid (``dense_conv``,
SURVEY.md 8f #4; ``dense_conv="stock"`` keeps them on PyTorch-ROCm / MIOpen as the north_star's minimum prescribes).
``forward`` is the inference engine (eval-mode BN, dropout off); ``forward_train`` (training mode) applies the reference's
residual dropouts and the ``random_noise`` Point-to-Grid jitter (:992-995) -- see fusion_train.py for the one stochastic
op that is not reproduced (dropout on the attention probabilities).
"""
import torch
from torch import nn

from . import fusion_ops as ops
from .dense_conv import PackedConvBN, SplitMap
from .fusion_modules import (ConvModule, InsContextAtt, Instane2SceneAtt, SSTInputLayerV2, SSTv2)


class ISFusionEncoder(nn.Module):

    def __init__(self, num_points_in_pillar=10, embed_dims=256, num_classes=10, **kwargs):
        super().__init__()
        self.num_points_in_pillar = num_points_in_pillar
        self.bev_size = kwargs.get("bev_size", 180)
        self.num_views = kwargs.get("num_views", 6)
        self.dense_conv = kwargs.get("dense_conv", "hip")
        region_shape = kwargs.get("region_shape", None)
        grid_size = kwargs.get("grid_size", None)
        region_drop_info = kwargs.get("region_drop_info", None) or [None] * len(region_shape)
        self.embed_dims = embed_dims
        E = embed_dims // 2
        self.conv_fusion = ConvModule(embed_dims * 3, E)
        self.get_regions = nn.ModuleList()
        self.grid2region_att = nn.ModuleList()
        for l in range(len(region_shape)):
            d = E * (l + 1)
            assert tuple(region_shape[l][:2]) == (region_shape[l][0],) * 2, "square windows only"
            self.get_regions.append(SSTInputLayerV2(window_shape=region_shape[l], sparse_shape=grid_size[l],
                                                    drop_info=region_drop_info[l], pos_temperature=1000,
                                                    pos_embed=d))
            self.grid2region_att.append(SSTv2(d_model=[d] * 4, nhead=[8] * 4, num_blocks=1,
                                              dim_feedforward=[d] * 4, output_shape=grid_size[l][:2],
                                              in_channel=E if l == 0 else None))
        self.random_noise = 1.0          # fusion_encoder.py:859 (training only)
        self.instance_num = kwargs.get("instance_num", 200)
        self.nms_kernel_size = 3
        self.conv_ins = ConvModule(E, E)
        self.conv_scene = ConvModule(E, E)
        self.conv_heatmap = ConvModule(E, E)
        self.heatmap_head_1 = ConvModule(E, embed_dims // 4)
        self.heatmap_head_2 = ConvModule(embed_dims // 4, embed_dims // 4)
        self.heatmap_head_3 = nn.Conv2d(embed_dims // 4, num_classes, kernel_size=3, stride=1, padding=1)
        self.instance_att = InsContextAtt(num_layers=2, embed_dims=E, bev_size=self.bev_size)
        self.instance_to_scene_att = Instane2SceneAtt(d_model=E)

    # --------------------------------------------------------------------------------------------- pieces
    def _conv(self, name):
        """ConvModule `name` packed for the f16x3 kernel (cached per device)"""
        mod = getattr(self, name)
        if isinstance(mod, nn.Conv2d):                  # heatmap_head_3: plain conv with bias, no BN / ReLU
            c = ops._cache(mod, mod.weight.device)
            if "packed" not in c:
                c["packed"] = PackedConvBN(mod, None, relu=False)
            return c["packed"]
        c = ops._cache(mod, mod.conv.weight.device)     # dropped when the conv / BN tensors change
        if "packed" not in c:
            c["packed"] = PackedConvBN(mod.conv, mod.bn, relu=True)
        return c["packed"]

    def img_fv_to_bev(self, mlvl_feats, bs, **kwargs):
        """A8 Point-to-Grid: one kernel over (pillar, slot, camera) instead of B*6 grid_sample calls."""
        pm = kwargs["pts_metas"]
        return ops.p2g_sample(pm["pillars"], pm["pillar_coors"], mlvl_feats[0], kwargs.get("lidar2img"),
                              kwargs.get("img_aug_matrix"), kwargs.get("lidar_aug_matrix"),   # unused when p2g_cam is given
                              kwargs["img_metas"][0]["input_shape"], bs, self.bev_size, self.num_views,
                              cam=kwargs.get("p2g_cam"), out=kwargs.get("p2g_out"), split=bool(kwargs.get("p2g_split")))

    def fuse(self, img_bev, lidar_feats):
        """conv_fusion(cat([img_bev, lidar_feats])) (fusion_encoder.py:1163-1165) -> [B, E, S, S]"""
        if self.dense_conv != "hip":
            if isinstance(img_bev, SplitMap):
                img_bev = img_bev.to_nchw()
            if isinstance(lidar_feats, (list, tuple)):
                lidar_feats = torch.cat([m.to_nchw() for m in lidar_feats], 1)
            return self.conv_fusion(torch.cat([img_bev, lidar_feats], dim=1))
        maps = [img_bev if isinstance(img_bev, SplitMap) else SplitMap.from_nchw(img_bev)]   # (Point-to-Grid wrote split rows)
        if isinstance(lidar_feats, (list, tuple)):       # the LiDAR branch handed its map over in split form already
            maps += list(lidar_feats)                    # (LidarBranch.forward(bev_split=True))
        else:
            for off in range(0, lidar_feats.size(1), 256):
                maps.append(SplitMap.from_nchw(lidar_feats, off, min(256, lidar_feats.size(1) - off)))
        m = self._conv("conv_fusion")(maps)
        self.__dict__["_bev_split"] = m        # the instance branch convolves the same map: it takes the split form as it is
        return m.to_nchw()

    def grid2region(self, level, bev):
        """A10/A11: SSTInputLayerV2 + SSTv2 on the dense [B, C, S, S] grid."""
        win = self.get_regions[level].window_shape[0]
        return ops.sstv2_forward(self.grid2region_att[level], bev, win,
                                 float(self.get_regions[level].pos_temperature))

    def instance_fusion(self, bev_feats, scene_feats, bs, **kwargs):
        """A12-A14 (fusion_encoder.py:1090-1149).  The reference convolves the spatially transposed map
        (`bev_feats.permute(0, 1, 3, 2)`, :1093,1139); the hip path applies the transposed 3x3 kernels to the
        un-transposed tokens instead, so no transposed copy of the map is ever made."""
        S = self.bev_size
        if self.dense_conv == "hip":
            m = kwargs.get("bev_split")
            m = m if m is not None else SplitMap.from_nchw(bev_feats)
            t = self._conv("heatmap_head_2")(self._conv("heatmap_head_1")(self._conv("conv_heatmap")(m, True), True),
                                             True)                                   # un-transposed orientation
            # 64 -> 10 channels: the same kernel with the output columns zero-padded to its narrowest tile (32), again
            # with the transposed taps on the un-transposed tokens; the 10-channel result is then transposed into the
            # orientation the reference returns (round 2 ran this conv on MIOpen: three extra library kernels)
            h3 = self.heatmap_head_3
            hm = self._conv("heatmap_head_3")(t, True).to_nchw(h3.out_channels)
            hm = hm.permute(0, 1, 3, 2).contiguous()
            x_scene_t = self._conv("conv_scene")(m, True).to_nchw()                   # = conv_scene(out)^T
            q = self._conv("conv_ins")(m).to_nchw()
        else:
            out = bev_feats.permute(0, 1, 3, 2).contiguous()
            hm = self.heatmap_head_3(self.heatmap_head_2(self.heatmap_head_1(self.conv_heatmap(out))))
            x_scene_t = self.conv_scene(out).permute(0, 1, 3, 2).contiguous()
            q = self.conv_ins(bev_feats)
        top32, _, _ = ops.instance_topk(hm, self.instance_num, self.nms_kernel_size,
                                        (8, 9) if self.num_views == 6 else (1, 2), as_int32=True)
        # cell n' = y'*S + x' of the transposed map is cell x'*S + y' of the un-transposed one; query_pos = (x' + .5,
        # y' + .5) = create_2D_grid's position of that cell: features, positions, position embedding in one launch
        mined = ops.mined_instances(self.instance_att, top32, x_scene_t, S)
        self.last_top_idx = mined["top"]   # [B, instance_num] flat cell (of the transposed map) of every mined instance
        x_ins = ops.ins_context_att(self.instance_att, None, None, x_scene_t, S, mined=mined)
        ret = ops.instance_to_scene(self.instance_to_scene_att, q, x_ins, scene_feats, S)
        return ret, hm

    def forward_train(self, img_mlvl_feats, lidar_feats, bs, **kwargs):
        """training mode: the same data flow with gradients -- HIP forward kernels inside autograd
        Functions (fusion_train.py), the 3x3 convolutions + BatchNorm (batch statistics) on the sparse-conv kernels over the
        dense grid (dense_train.py; `dense_conv="stock"`: PyTorch-ROCm / MIOpen, what north_star allows there), instance
        mining without gradient (indices), like the reference."""
        from . import fusion_train as tr
        tr.pack_stock_convs(self)
        if kwargs.get("pts_backbone", None) is not None:
            tr.pack_stock_convs(kwargs["pts_backbone"])
        pm = kwargs["pts_metas"]
        S = self.bev_size
        noise = None
        if self.random_noise is not None and self.training:
            # one draw per sample, from the same two generators in the same order as the reference (:992-995)
            import random
            import numpy as np
            noise = [random.uniform(-self.random_noise, self.random_noise) if np.random.rand() > 0.5 else 0.0
                     for _ in range(bs)]
        img_bev = tr.p2g_sample(pm["pillars"], pm["pillar_coors"], img_mlvl_feats[1], kwargs["lidar2img"],
                                kwargs["img_aug_matrix"], kwargs["lidar_aug_matrix"],
                                kwargs["img_metas"][0]["input_shape"], bs, S, self.num_views, noise)
        from . import dense_train as dt
        # 3x3 conv + BatchNorm stacks: the sparse-conv kernels over the dense grid with their own backward (dense_train.py);
        # dense_conv = "stock": the nn modules (MIOpen)
        if self.dense_conv == "hip":
            stack = dt.conv_stack
        else:
            def stack(seq, t, transpose=False):
                t = torch.cat(list(t), 1) if isinstance(t, (list, tuple)) else t
                return seq(t.permute(0, 1, 3, 2).contiguous()).permute(0, 1, 3, 2) if transpose else seq(t)
        bev_feats = stack(self.conv_fusion, [img_bev, lidar_feats])
        pts_backbone = kwargs.get("pts_backbone", None)
        x, ins_hm, feats = bev_feats, None, []
        for i in range(len(self.get_regions)):
            win = self.get_regions[i].window_shape[0]
            x = tr.sstv2_forward(self.grid2region_att[i], x, win, float(self.get_regions[i].pos_temperature))
            if i == 0:
                scene_feats = x
                # the reference convolves the spatially transposed map (:1093, :1139): transposed taps on the un-transposed
                # tokens instead (transpose=True), as the inference path does
                t = stack(self.heatmap_head_2, stack(self.heatmap_head_1, stack(self.conv_heatmap, bev_feats, True), True),
                          True)
                ins_hm = self.heatmap_head_3(t.permute(0, 1, 3, 2))       # 64 -> 10 classes: stock conv
                x_scene_t = stack(self.conv_scene, bev_feats, True)
                q = stack(self.conv_ins, bev_feats)
                with torch.no_grad():
                    top_idx = ops.instance_topk(ins_hm.detach(), self.instance_num, self.nms_kernel_size,
                                                (8, 9) if self.num_views == 6 else (1, 2))
                self.last_top_idx = top_idx
                idx_t = (top_idx % S) * S + torch.div(top_idx, S, rounding_mode="floor")
                x_ins, _ = ops.gather_instances(x_scene_t, idx_t, S)
                _, query_pos = ops.gather_instances(x_scene_t, top_idx, S)
                x_ins = tr.ins_context_att(self.instance_att, x_ins, query_pos, x_scene_t, S)
                x = tr.instance_to_scene(self.instance_to_scene_att, q, x_ins, scene_feats, S)
            if kwargs.get("feats_split"):     # engine-level hand-over: the stage's results stay split-format token matrices
                nxt, _, this_feat = pts_backbone([x], "stage{}".format(i + 1), keep_split=True)
            else:
                nxt, _, this_feat = pts_backbone([x], "stage{}".format(i + 1))
            feats.append(this_feat)
            x = nxt
        return feats, ins_hm

    def forward(self, img_mlvl_feats, lidar_feats, bs, **kwargs):
        if self.training:
            return self.forward_train(img_mlvl_feats, lidar_feats, bs, **kwargs)
        with torch.no_grad():
            return self.forward_eval(img_mlvl_feats, lidar_feats, bs, **kwargs)

    def forward_eval(self, img_mlvl_feats, lidar_feats, bs, **kwargs):
        img_bev = kwargs.pop("img_bev", None)     # Point-to-Grid already run by the caller (detector: on its side stream)
        if img_bev is None:
            img_bev = self.img_fv_to_bev([img_mlvl_feats[1]], bs, **kwargs)
        return self.forward_tail(img_bev, lidar_feats, bs, **kwargs)

    def forward_tail(self, img_bev, lidar_feats, bs, **kwargs):
        """everything behind Point-to-Grid: shape-static for a given batch size (no pillar count in it), which is what
        ISFusionPtsPath captures in a HIP graph"""
        self.__dict__["_bev_split"] = None
        bev_feats = self.fuse(img_bev, lidar_feats)
        kwargs["bev_split"] = self.__dict__.pop("_bev_split", None)   # conv_fusion's output before its NCHW conversion
        pts_backbone = kwargs.get("pts_backbone", None)
        x = bev_feats
        ins_hm = None
        feats = []
        for i in range(len(self.get_regions)):
            x = self.grid2region(i, x)
            if i == 0:
                x, ins_hm = self.instance_fusion(bev_feats, x, bs, **kwargs)
            nxt, _, this_feat = pts_backbone([x], "stage{}".format(i + 1))
            feats.append(this_feat)
            x = nxt
        return feats, ins_hm
