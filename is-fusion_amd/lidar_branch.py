"""LiDAR branch of ISFusionDetector.extract_pts_feat (detectors/isfusion.py:103-111) as one engine:

    dynamic_voxelize(pts) -> DynamicVFE -> SparseEncoder -> spatial_features [B, 512, 180, 180]

``LidarBranch.forward`` is ONE C call (isf_lidar_branch_forward): no per-sample Python loop, no
``coors[-1,0].item()`` sync; the only host syncs left are the data-dependent voxel counts (one per
resolution level).  The sub-modules are the drop-in ``DynamicVFE`` / ``SparseEncoder`` classes, so a
reference state dict (keys ``pts_voxel_encoder.*`` / ``pts_middle_encoder.*``) loads directly.
"""
import ctypes

import torch
from torch import nn

from . import _lib
from .norm import fold_bn
from .sparse_encoder import SparseEncoder
from .voxel_encoder import DynamicVFE

# configs/isfusion/isfusion_0075voxel.py:5-7,60-84 (values restated, not imported)
ISFUSION_0075 = dict(
    voxel_size=[0.075, 0.075, 0.2],
    point_cloud_range=[-54.0, -54.0, -5.0, 54.0, 54.0, 3.0],
    pts_voxel_encoder=dict(in_channels=5, feat_channels=[64, 64], with_distance=False, with_cluster_center=True,
                           with_voxel_center=True, norm_cfg=dict(type="naiveSyncBN1d", eps=1e-3, momentum=0.01)),
    pts_middle_encoder=dict(in_channels=64, sparse_shape=[41, 1440, 1440], base_channels=32, output_channels=256,
                            order=("conv", "norm", "act"),
                            encoder_channels=((32, 32, 64), (64, 64, 128), (128, 128, 256), (256, 256)),
                            encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0)),
                            block_type="basicblock"),
)


class LidarBranch(nn.Module):

    def __init__(self, voxel_size=None, point_cloud_range=None, pts_voxel_encoder=None, pts_middle_encoder=None):
        super().__init__()
        cfg = ISFUSION_0075
        self.voxel_size = list(voxel_size or cfg["voxel_size"])
        self.point_cloud_range = list(point_cloud_range or cfg["point_cloud_range"])
        ve = dict(pts_voxel_encoder or cfg["pts_voxel_encoder"])
        ve.pop("type", None)
        ve.setdefault("voxel_size", self.voxel_size)
        ve.setdefault("point_cloud_range", self.point_cloud_range)
        me = dict(pts_middle_encoder or cfg["pts_middle_encoder"])
        me.pop("type", None)
        self.pts_voxel_encoder = DynamicVFE(**ve)
        self.pts_middle_encoder = SparseEncoder(**me)
        self.last_stats = None
        self._vfe_cache = None
        self._vfe_key = None

    def randomize_bn_(self, seed=0):
        """Random but well-conditioned BN running statistics / affine parameters (bench + parity tests:
        a freshly constructed BN is the identity, which would hide scale/shift mistakes)."""
        g = torch.Generator().manual_seed(seed)
        for m in self.modules():
            if isinstance(m, nn.modules.batchnorm._BatchNorm):
                with torch.no_grad():
                    m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                    m.running_var.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)
                    m.weight.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)
                    m.bias.copy_(torch.randn(m.num_features, generator=g) * 0.1)
        return self

    def randomize_weights_(self, seed=0, fan_taps=9.0):
        """Random weights sized for the SPARSE fan-in (roughly ``fan_taps`` of the 27 taps are active for
        LiDAR voxels) and for metre-scale inputs, so activations stay O(1) through the VFE and the 21 conv
        layers and a 1e-3 absolute parity bound is a meaningful relative bound too.  (The default init
        follows the reference's kaiming_uniform on the [k,k,k,Cin,Cout] layout, conv.py:105-112, which
        shrinks activations by ~100x over the encoder.)"""
        g = torch.Generator().manual_seed(seed)
        from .spconv import SparseConvolution
        for m in self.modules():
            if isinstance(m, SparseConvolution):
                std = (1.0 / (fan_taps * m.in_channels)) ** 0.5
                with torch.no_grad():
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * std)
            elif isinstance(m, nn.Linear):
                # layer 1 sees raw coordinates (tens of metres); layer 2 sees post-ReLU features
                std = 0.02 if m.in_features < 32 else (1.0 / m.in_features) ** 0.5
                with torch.no_grad():
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * std)
        return self

    def freeze(self, flag=True):
        """Skip the per-call parameter-change scans (weights are static at inference).  Ends by itself on a
        load_state_dict below this module or a forward in training mode (fusion_ops.freeze / frozen)."""
        from . import fusion_ops as ops
        self._frozen = bool(flag)
        ops.freeze(self, flag)
        return self

    def _vfe_params(self):
        vfe = self.pts_voxel_encoder
        from .fusion_ops import frozen
        if self._vfe_cache is not None and getattr(self, "_frozen", False) and frozen(self):
            return self._vfe_cache
        key = tuple((p._version, p.data_ptr()) for p in list(vfe.parameters()) + list(vfe.buffers()))
        if self._vfe_cache is not None and self._vfe_key == key:
            return self._vfe_cache
        l1, l2 = vfe.vfe_layers
        s1, b1 = fold_bn(l1.norm)
        s2, b2 = fold_bn(l2.norm)
        w1 = l1.linear.weight.detach().float().contiguous()
        w2 = l2.linear.weight.detach().float().contiguous()
        p = _lib.VfeParams()
        p.in_channels, p.c1, p.c2 = vfe.raw_in_channels, w1.size(0), w2.size(0)
        p.w1, p.scale1, p.shift1 = w1.data_ptr(), s1.data_ptr(), b1.data_ptr()
        p.w2, p.scale2, p.shift2 = w2.data_ptr(), s2.data_ptr(), b2.data_ptr()
        for j in range(3):
            p.voxel_size[j] = float(self.voxel_size[j])
        for j in range(6):
            p.coors_range[j] = float(self.point_cloud_range[j])
        self._vfe_cache = (p, (w1, w2, s1, b1, s2, b2))
        self._vfe_key = key
        return self._vfe_cache

    def forward_train(self, points):
        """training mode: the reference's composition -- dynamic voxelize, DynamicVFE module path
        (DynamicScatter with backward, BatchNorm batch statistics), SparseEncoder module by module on the sparse-conv
        autograd Function -- instead of the one-call inference engine.  -> spatial_features [B, C*D, H, W]"""
        from .voxelize import dynamic_voxelize_batched
        pts, coors = dynamic_voxelize_batched(points, self.voxel_size, self.point_cloud_range)
        keep = (coors[:, 1:] >= 0).all(1)          # the reference's DynamicScatter drops out-of-range points the same way
        vf, vc = self.pts_voxel_encoder(pts[keep].float(), coors[keep])
        return self.pts_middle_encoder.forward_modules(vf, vc, len(points))[0]

    def forward(self, points, time_layers=False, want_stats=False, precision=0, conv_diag=0, stage_rows=0,
                stage_mask=0, out=None, bev_split=False):
        """points: list of [P_i, C] tensors (one per sample) -> spatial_features [B, C*D, H, W]
        (bev_split=True, inference: the same map as a list of dense_conv.SplitMap, one per 256-channel group -- the form the
        fusion encoder's convolutions read; isf_encoder_options.bev_format = 1).
        precision: 0 = f16x3 split MFMA (default, fp32-class), 1 = fp32 MFMA kernels, 2 = single-pass f16 (opt-in,
        fp16-autocast accuracy); conv_diag: timing diagnostics of the conv kernels (results garbage except 16);
        stage_rows / stage_mask: LDS staging of the conv input rows (isf_encoder_options; 0 = library default)."""
        if self.training:
            return self.forward_train(points)
        with torch.no_grad():
            return self.forward_eval(points, time_layers, want_stats, precision, conv_diag, stage_rows, stage_mask, out,
                                     bev_split)

    def forward_eval(self, points, time_layers=False, want_stats=False, precision=0, conv_diag=0, stage_rows=0,
                     stage_mask=0, out=None, bev_split=False):
        pts = torch.cat(points, dim=0).contiguous().float()
        _lib.require_cuda(pts)
        vfe = self.pts_voxel_encoder
        if pts.size(1) != vfe.raw_in_channels:
            raise _lib.IsfError(f"LidarBranch: points have {pts.size(1)} columns, the voxel encoder expects "
                                f"{vfe.raw_in_channels}")
        if not vfe._fusable():
            raise _lib.IsfError("LidarBranch: the one-call engine implements the IS-Fusion DynamicVFE configuration "
                                "(2 layers of 64, cluster + voxel centre, max pooling, no distance feature); build the "
                                "path from the sub-modules for anything else")
        offs = [0]
        for p in points:
            offs.append(offs[-1] + p.size(0))
        B = len(points)
        me = self.pts_middle_encoder
        arr, n, _keep, _plan = me._c_plan()
        vp, _keep2 = self._vfe_params()
        cd, H, W = me.out_channels_and_shape()
        if out is None:    # out: a caller-owned [B, C*D, H, W] buffer (the detector's HIP-graph input)
            out = torch.empty((B, cd, H, W), dtype=torch.float32, device=pts.device)
        assert tuple(out.shape) == (B, cd, H, W) and out.is_contiguous() and out.dtype == torch.float32
        oshape = (ctypes.c_int * 4)()
        stats = _lib.EncoderStats() if (want_stats or time_layers) else None
        lib = _lib.load()
        _lib.check(lib.isf_lidar_branch_forward(
            _lib.ptr(pts), (ctypes.c_int64 * len(offs))(*offs), B, ctypes.byref(vp), _lib.i3(me.sparse_shape),
            arr, n, _lib.ptr(out), oshape, ctypes.byref(stats) if stats is not None else None,
            int(bool(time_layers)), _lib.encoder_options(precision, conv_diag, stage_rows, stage_mask, int(bool(bev_split))),
            _lib.stream()), "isf_lidar_branch_forward")
        self.last_stats = stats
        if bev_split:     # the buffer holds cd / 256 split-format token matrices [B*H*W, 256] (same bytes as the fp32 map)
            from .dense_conv import SplitMap
            if cd % 256:
                raise _lib.IsfError(f"LidarBranch: bev_split needs whole 256-channel groups, the BEV map has {cd} channels")
            raw = out.view(torch.uint8).view(-1)
            step = B * H * W * 256 * 4
            return [SplitMap(raw[g * step:(g + 1) * step], B, 256, H, W) for g in range(cd // 256)]
        return out

    def conv_layer_table(self):
        """[(kind, c_in, c_out, K)] of the encoder plan, for roofline accounting."""
        plan = self.pts_middle_encoder.export_plan()
        return [(L["kind"], L["c_in"], L["c_out"], L["ksize"][0] * L["ksize"][1] * L["ksize"][2])
                for L in plan["layers"]]
