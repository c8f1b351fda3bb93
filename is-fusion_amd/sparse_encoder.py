"""SparseEncoder (drop-in for mmdet3d/models/middle_encoders/sparse_encoder.py:17-216).

Same constructor kwargs, same sub-module names (``conv_input``, ``encoder_layers.encoder_layer{i}``,
``conv_out``) => same state-dict keys, same 3-tuple return ``(spatial_features, encode_features, kwargs)``.

Eval / no-grad forward is ONE C call (``isf_sparse_encoder_forward``): the module tree is flattened once
into a plan of conv layers with folded BatchNorm, ReLU and residual sources; the library builds one
rulebook per resolution, runs 21 fused conv kernels and writes the dense BEV tensor.  In that mode
``encode_features`` (unused by ISFusionDetector, isfusion.py:111) is a lazy list: the per-stage tensors are computed by
the module path the first time somebody reads it (the fused engine does not keep them).
"""
import ctypes

import numpy as np
import torch
from torch import nn

from . import _lib
from .norm import fold_bn
from .sparse_block import SparseBasicBlock, make_sparse_convmodule
from .spconv import SparseConvolution, SparseConvTensor, SparseSequential


class _LazyEncodeFeatures(list):
    """The `encode_features` entry of SparseEncoder.forward's 3-tuple on the fused path: a list whose content (the output
    SparseConvTensor of every encoder stage, mmdet3d/models/middle_encoders/sparse_encoder.py:131-138) is produced by
    `make()` -- the module-by-module forward -- on the first read.  ISFusionDetector.extract_pts_feat never reads it
    (isfusion.py:111), so the hot path pays nothing; a caller that does gets the reference's list."""

    def __init__(self, make):
        super().__init__()
        self._make = make

    def _fill(self):
        if self._make is not None:
            make, self._make = self._make, None
            with torch.no_grad():
                super().extend(make())

    def __len__(self):
        self._fill()
        return super().__len__()

    def __getitem__(self, i):
        self._fill()
        return super().__getitem__(i)

    def __iter__(self):
        self._fill()
        return super().__iter__()

    def __repr__(self):
        return "<encode_features: not computed>" if self._make is not None else super().__repr__()


class SparseEncoder(nn.Module):

    def __init__(self, in_channels, sparse_shape, order=("conv", "norm", "act"),
                 norm_cfg=dict(type="BN1d", eps=1e-3, momentum=0.01), base_channels=16, output_channels=128,
                 encoder_channels=((16,), (32, 32, 32), (64, 64, 64), (64, 64, 64)),
                 encoder_paddings=((1,), (1, 1, 1), (1, 1, 1), ((0, 1, 1), 1, 1)), block_type="conv_module",
                 **kwargs):
        super().__init__()
        assert block_type in ["conv_module", "basicblock"]
        assert isinstance(order, tuple) and len(order) == 3 and set(order) == {"conv", "norm", "act"}
        self.sparse_shape = list(sparse_shape)
        self.in_channels = in_channels
        self.order = order
        self.base_channels = base_channels
        self.output_channels = output_channels
        self.encoder_channels = encoder_channels
        self.encoder_paddings = encoder_paddings
        self.stage_num = len(encoder_channels)
        self.fp16_enabled = False
        pre_act = order[0] != "conv"
        self.conv_input = make_sparse_convmodule(in_channels, base_channels, 3, norm_cfg=norm_cfg, padding=1,
                                                 indice_key="subm1", conv_type="SubMConv3d",
                                                 order=("conv",) if pre_act else ("conv", "norm", "act"))
        out_ch = self._make_encoder_layers(norm_cfg, base_channels, block_type)
        self.conv_out = make_sparse_convmodule(out_ch, output_channels, kernel_size=(3, 1, 1), stride=(2, 1, 1),
                                               norm_cfg=norm_cfg, padding=0, indice_key="spconv_down2",
                                               conv_type="SparseConv3d")
        self._plan = None
        self._plan_key = None
        self._frozen = False

    def _make_encoder_layers(self, norm_cfg, in_channels, block_type):
        """Stage layout of sparse_encoder.py:142-216."""
        self.encoder_layers = SparseSequential()
        out_channels = in_channels
        last_stage = len(self.encoder_channels) - 1
        for i, blocks in enumerate(self.encoder_channels):
            mods = []
            blocks = tuple(blocks)
            for j, out_channels in enumerate(blocks):
                padding = tuple(self.encoder_paddings[i])[j]
                strided = dict(stride=2, padding=padding, indice_key=f"spconv{i + 1}", conv_type="SparseConv3d")
                if block_type == "conv_module" and i != 0 and j == 0:
                    mods.append(make_sparse_convmodule(in_channels, out_channels, 3, norm_cfg=norm_cfg, **strided))
                elif block_type == "basicblock":
                    if j == len(blocks) - 1 and i != last_stage:
                        mods.append(make_sparse_convmodule(in_channels, out_channels, 3, norm_cfg=norm_cfg,
                                                           **strided))
                    else:
                        mods.append(SparseBasicBlock(out_channels, out_channels, norm_cfg=norm_cfg,
                                                     conv_cfg=dict(type="SubMConv3d")))
                else:
                    mods.append(make_sparse_convmodule(in_channels, out_channels, 3, norm_cfg=norm_cfg,
                                                       padding=padding, indice_key=f"subm{i + 1}",
                                                       conv_type="SubMConv3d"))
                in_channels = out_channels
            self.encoder_layers.add_module(f"encoder_layer{i + 1}", SparseSequential(*mods))
        return out_channels

    # ------------------------------------------------------------------------------------ plan
    def export_plan(self):
        """Flatten the module tree into conv layers with folded BN (python dicts; tensors stay on device).
        Also consumed (through ``plan_to_numpy``) by the CPU oracle in tests."""
        layers = []

        def conv_entry(conv, bn, relu, residual_from):
            assert conv.bias is None, "sparse convs on the IS-Fusion path have no bias (sparse_block.py:184)"
            scale, shift = fold_bn(bn) if bn is not None else (None, None)
            layers.append(dict(kind="subm" if conv.subm else "spconv", ksize=list(conv.kernel_size),
                               stride=list(conv.stride), padding=list(conv.padding), c_in=conv.in_channels,
                               c_out=conv.out_channels, conv=conv, scale=scale, shift=shift, relu=relu,
                               residual_from=residual_from))

        def walk_seq(seq):
            mods = list(seq._modules.values())
            i = 0
            while i < len(mods):
                m = mods[i]
                if isinstance(m, SparseBasicBlock):
                    src = len(layers) - 1  # output of the previous layer (-1 = encoder input)
                    conv_entry(m.conv1, m.bn1, True, None)
                    conv_entry(m.conv2, m.bn2, True, src)
                    i += 1
                elif isinstance(m, SparseConvolution):
                    bn, relu = None, False
                    j = i + 1
                    if j < len(mods) and isinstance(mods[j], nn.modules.batchnorm._BatchNorm):
                        bn = mods[j]
                        j += 1
                    if j < len(mods) and isinstance(mods[j], nn.ReLU):
                        relu = True
                        j += 1
                    conv_entry(m, bn, relu, None)
                    i = j
                elif isinstance(m, SparseSequential):
                    walk_seq(m)
                    i += 1
                else:
                    raise NotImplementedError(f"SparseEncoder fused plan: unsupported module {type(m).__name__} "
                                              "(pre-activation orders run through the module path)")
            return

        walk_seq(self.conv_input)
        walk_seq(self.encoder_layers)
        walk_seq(self.conv_out)
        return dict(sparse_shape=list(self.sparse_shape), layers=layers)

    def plan_to_numpy(self, plan=None):
        """CPU copy of the plan in the format oracle.sparse_encoder_forward consumes."""
        plan = plan or self.export_plan()
        out = []
        for L in plan["layers"]:
            out.append(dict(kind=L["kind"], ksize=L["ksize"], stride=L["stride"], padding=L["padding"],
                            weight=L["conv"].weight.detach().float().cpu().numpy(),
                            scale=L["scale"].cpu().numpy() if L["scale"] is not None else
                            np.ones(L["c_out"], np.float32),
                            shift=L["shift"].cpu().numpy() if L["shift"] is not None else
                            np.zeros(L["c_out"], np.float32),
                            relu=L["relu"], residual_from=L["residual_from"]))
        return dict(sparse_shape=plan["sparse_shape"], layers=out)

    def freeze(self, flag=True):
        """Inference deployments: skip the per-call "did a parameter change?" scan (about 130 tensors).  Ends by itself
        on a load_state_dict below this module or a forward in training mode (fusion_ops.freeze / frozen)."""
        from . import fusion_ops as ops
        self._frozen = bool(flag)
        ops.freeze(self, flag)
        return self

    def _c_plan(self):
        """ctypes array of isf_conv_layer, rebuilt when a parameter / buffer changed."""
        from .fusion_ops import frozen
        if self._plan is not None and self._frozen and frozen(self):
            return self._plan
        key = tuple((p._version, p.data_ptr()) for p in list(self.parameters()) + list(self.buffers()))
        if self._plan is not None and self._plan_key == key:
            return self._plan
        plan = self.export_plan()
        n = len(plan["layers"])
        arr = (_lib.ConvLayer * n)()
        keep = []
        for i, L in enumerate(plan["layers"]):
            c = arr[i]
            c.conv_type = _lib.CONV_SUBM if L["kind"] == "subm" else _lib.CONV_SPARSE
            for j in range(3):
                c.ksize[j], c.stride[j], c.padding[j] = L["ksize"][j], L["stride"][j], L["padding"][j]
            c.c_in, c.c_out = L["c_in"], L["c_out"]
            packed = L["conv"].packed_weight()
            packed16 = L["conv"].packed16_weight()
            c.packed16 = packed16.data_ptr() if packed16 is not None else None
            keep.append(packed16)
            scale = L["scale"] if L["scale"] is not None else torch.ones(L["c_out"], device=packed.device)
            shift = L["shift"] if L["shift"] is not None else torch.zeros(L["c_out"], device=packed.device)
            keep += [packed, scale, shift]
            c.packed, c.scale, c.shift = packed.data_ptr(), scale.data_ptr(), shift.data_ptr()
            c.relu = int(L["relu"])
            c.residual_from = -2 if L["residual_from"] is None else int(L["residual_from"])
        self._plan = (arr, n, keep, plan)
        self._plan_key = key
        self._out_shape = self._compute_out_shape(plan)
        return self._plan

    def out_channels_and_shape(self):
        """(C*D, H, W) of spatial_features for this configuration."""
        self._c_plan()
        return self._out_shape

    def _compute_out_shape(self, plan):
        shape = list(self.sparse_shape)
        lib = _lib.load()
        c = None
        for L in plan["layers"]:
            if L["kind"] == "spconv":
                o = _lib.i3([0, 0, 0])
                _lib.check(lib.isf_conv_out_shape(_lib.i3(shape), _lib.i3(L["ksize"]), _lib.i3(L["stride"]),
                                                  _lib.i3(L["padding"]), o))
                shape = list(o)
            c = L["c_out"]
        return c * shape[0], shape[1], shape[2]

    # ------------------------------------------------------------------------------------ forward
    def forward_fused(self, voxel_features, coors, batch_size, stats=None, time_layers=False, precision=0,
                      conv_diag=0, stage_rows=0, stage_mask=0):
        """precision: 0 = f16x3 split MFMA (default), 1 = fp32 MFMA, 2 = single-pass f16 (opt-in, fp16-autocast
        accuracy); conv_diag: timing diagnostics of the conv kernels (include/isf_hip.h) -- per call, no global state"""
        _lib.require_cuda(voxel_features, coors)
        arr, n, _keep, _plan = self._c_plan()
        cd, H, W = self.out_channels_and_shape()
        out = torch.empty((batch_size, cd, H, W), dtype=torch.float32, device=voxel_features.device)
        oshape = (ctypes.c_int * 4)()
        lib = _lib.load()
        vf, vc = voxel_features.contiguous().float(), coors.contiguous().int()   # locals: must outlive the C call
        _lib.check(lib.isf_sparse_encoder_forward(
            _lib.ptr(vf), _lib.ptr(vc),
            voxel_features.size(0), int(batch_size), _lib.i3(self.sparse_shape), arr, n, _lib.ptr(out), oshape,
            ctypes.byref(stats) if stats is not None else None, int(bool(time_layers)),
            _lib.encoder_options(precision, conv_diag, stage_rows, stage_mask), _lib.stream()),
            "isf_sparse_encoder_forward")
        assert (oshape[0], oshape[1], oshape[2]) == (cd, H, W)
        return out

    def forward_modules(self, voxel_features, coors, batch_size):
        x = self.conv_input(SparseConvTensor(voxel_features, coors.int(), self.sparse_shape, batch_size))
        encode_features = [x]
        for encoder_layer in self.encoder_layers._modules.values():
            x = encoder_layer(x)
            encode_features.append(x)
        out = self.conv_out(encode_features[-1])
        sp = out.dense()
        N, C, D, H, W = sp.shape
        return sp.view(N, C * D, H, W), encode_features

    def forward(self, voxel_features, coors, batch_size, swin_format=False, img_feats=None, **kwargs):
        fused_ok = (not self.training) and not (torch.is_grad_enabled() and voxel_features.requires_grad) \
            and self.order[0] == "conv"
        if fused_ok:
            with torch.no_grad():
                # encode_features (the reference's per-stage SparseConvTensor list, sparse_encoder.py:131-138; unused by
                # isfusion.py:111): materialised by the module path on first access, never on the hot path
                lazy = _LazyEncodeFeatures(lambda: self.forward_modules(voxel_features, coors, batch_size)[1])
                return self.forward_fused(voxel_features, coors, batch_size), lazy, kwargs
        spatial, enc = self.forward_modules(voxel_features, coors, batch_size)
        return spatial, enc, kwargs
