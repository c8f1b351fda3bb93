"""Import the REFERENCE's Python modules for the HSF / IGF rows in the authoring container (no mmcv / mmdet /
spconv / cv2 / ipdb installed, no GPU) so that their own code can generate golden vectors.

Only third-party packages that are ABSENT are stubbed (registries, decorators, ConvModule = Conv2d+BN+ReLU,
...).  Every mmdet3d module is loaded from its real file under /root/reference; nothing is copied.  The two
compiled ops the path needs are replaced by the reference's own pure-torch implementations:
  * mmcv _ext.ms_deform_attn_forward  -> ops/functions/ms_deform_attn_func.py:ms_deform_attn_core_pytorch
  * TorchEx ingroup_indices.forward   -> first-come rank inside each group (the CUDA kernel's atomic order is
                                         arbitrary; any rank order is valid because windows never overflow)
Used by tests/golden/make_golden_fusion.py.  Not importable on the GPU box (there is no /root/reference there).
"""
import importlib.util
import os
import sys
import types

import torch
from torch import nn

REF = os.environ.get("ISF_REFERENCE_ROOT", "/root/reference")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, path=None, **attrs):
    m = _mod(name, **attrs)
    m.__path__ = [path] if path else []
    return m


class _Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls
        return deco(module) if module is not None else deco

    def build(self, cfg):
        cfg = dict(cfg)
        return self.module_dict[cfg.pop("type")](**cfg)


def _identity_decorator(*a, **k):
    def deco(fn):
        return fn
    return deco


class ConvModule(nn.Sequential):
    """mmcv.cnn.ConvModule for the configurations the path uses: Conv{1,2}d (bias = 'auto' -> no norm) + BN{1,2}d +
    ReLU, with the sub-module names mmcv gives them (conv / bn / activate)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type="ReLU"), bias="auto", **kw):
        super().__init__()
        ctype = (conv_cfg or {}).get("type", "Conv2d")
        conv = {"Conv2d": nn.Conv2d, "Conv1d": nn.Conv1d, None: nn.Conv2d}[ctype]
        if bias == "auto":
            bias = norm_cfg is None
        self.add_module("conv", conv(in_channels, out_channels, kernel_size, stride, padding, bias=bool(bias)))
        if norm_cfg is not None:
            ntype = norm_cfg.get("type", "BN")
            norm = {"BN": nn.BatchNorm2d, "BN2d": nn.BatchNorm2d, "BN1d": nn.BatchNorm1d}[ntype]
            self.add_module("bn", norm(out_channels, eps=norm_cfg.get("eps", 1e-5),
                                       momentum=norm_cfg.get("momentum", 0.1)))
        if act_cfg is not None:
            self.add_module("activate", nn.ReLU(inplace=True))


def build_conv_layer(cfg, *args, **kwargs):
    cfg = dict(cfg or dict(type="Conv2d"))
    t = cfg.pop("type")
    conv = {"Conv2d": nn.Conv2d, "Conv1d": nn.Conv1d, None: nn.Conv2d}[t]
    if "bias" in kwargs:
        kwargs["bias"] = bool(kwargs["bias"])   # mmcv passes bias='auto' straight through: truthy
    return conv(*args, **kwargs, **cfg)


def build_norm_layer(cfg, num_features, postfix=""):
    cfg = dict(cfg)
    t = cfg.pop("type")
    cfg.pop("requires_grad", None)
    cls = {"BN": nn.BatchNorm2d, "BN2d": nn.BatchNorm2d, "BN1d": nn.BatchNorm1d, "LN": nn.LayerNorm,
           "naiveSyncBN1d": nn.BatchNorm1d, "naiveSyncBN2d": nn.BatchNorm2d}[t]
    return "bn" + str(postfix), cls(num_features, **cfg)


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


_installed = {}


def install():
    """Install the stubs and load the reference modules; returns a dict of the loaded modules."""
    if _installed:
        return _installed
    if not os.path.isdir(os.path.join(REF, "mmdet3d")):
        raise RuntimeError("reference tree not available")
    _mod("ipdb", set_trace=lambda *a, **k: None)
    _mod("cv2")
    regs = {n: _Registry(n) for n in ("ATTENTION", "TRANSFORMER_LAYER", "TRANSFORMER_LAYER_SEQUENCE", "BACKBONES",
                                      "MIDDLE_ENCODERS", "FUSION_LAYERS")}
    mmcv = _pkg("mmcv")
    _pkg("mmcv.cnn", ConvModule=ConvModule, build_conv_layer=build_conv_layer, build_norm_layer=build_norm_layer)
    _pkg("mmcv.cnn.bricks")
    _mod("mmcv.cnn.bricks.registry", ATTENTION=regs["ATTENTION"], TRANSFORMER_LAYER=regs["TRANSFORMER_LAYER"],
         TRANSFORMER_LAYER_SEQUENCE=regs["TRANSFORMER_LAYER_SEQUENCE"])

    class BaseModule(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()

    _pkg("mmcv.runner", force_fp32=_identity_decorator, auto_fp16=_identity_decorator, BaseModule=BaseModule)
    _mod("mmcv.runner.base_module", BaseModule=BaseModule, ModuleList=nn.ModuleList, Sequential=nn.Sequential)
    # the pure-torch MS-deformable attention of the reference's own ops/ tree stands in for the CUDA extension
    _mod("MultiScaleDeformableAttention")
    core = _load("isf_ref_ms_deform_attn_func", "ops/functions/ms_deform_attn_func.py")

    class _Ext:
        @staticmethod
        def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_locations, attention_weights,
                                   im2col_step=64):
            return core.ms_deform_attn_core_pytorch(value, spatial_shapes, sampling_locations, attention_weights)

        @staticmethod
        def ms_deform_attn_backward(*a, **k):
            raise NotImplementedError

    _pkg("mmcv.utils", ext_loader=types.SimpleNamespace(load_ext=lambda *a, **k: _Ext), ConfigDict=dict,
         build_from_cfg=None, deprecated_api_warning=_identity_decorator, to_2tuple=lambda x: (x, x))
    _pkg("mmcv.ops", SparseConvTensor=object, SparseSequential=nn.Sequential)
    _mod("mmcv.ops.multi_scale_deform_attn", multi_scale_deformable_attn_pytorch=core.ms_deform_attn_core_pytorch)
    mmcv.__version__ = "stub"
    _pkg("mmdet")
    _pkg("mmdet.models", BACKBONES=regs["BACKBONES"])

    base = os.path.join(REF, "mmdet3d")
    _pkg("mmdet3d", base)
    _mod("mmdet3d.core", draw_heatmap_gaussian=None, gaussian_radius=None)
    builder = _mod("mmdet3d.models.builder", FUSION_LAYERS=regs["FUSION_LAYERS"],
                   MIDDLE_ENCODERS=regs["MIDDLE_ENCODERS"],
                   build_middle_encoder=lambda cfg: regs["MIDDLE_ENCODERS"].build(cfg),
                   build_backbone=lambda cfg: regs["BACKBONES"].build(cfg))
    _pkg("mmdet3d.models", os.path.join(base, "models"), builder=builder)
    _pkg("mmdet3d.models.middle_encoders", os.path.join(base, "models", "middle_encoders"))
    _pkg("mmdet3d.models.sst", os.path.join(base, "models", "sst"))
    _pkg("mmdet3d.models.backbones", os.path.join(base, "models", "backbones"))
    ops = _pkg("mmdet3d.ops", os.path.join(base, "ops"), SparseBasicBlock=None, make_sparse_convmodule=None)
    _pkg("mmdet3d.ops.spconv", IS_SPCONV2_AVAILABLE=False)
    ops.spconv = sys.modules["mmdet3d.ops.spconv"]

    # TorchEx ingroup_indices: rank of every element inside its group, first come first served
    def _ingroup_forward(group_inds, out_inds):
        g = group_inds.cpu()
        order = torch.argsort(g, stable=True)
        gs = g[order]
        start = torch.ones_like(gs, dtype=torch.bool)
        start[1:] = gs[1:] != gs[:-1]
        first = torch.cummax(torch.where(start, torch.arange(len(gs)), torch.zeros_like(gs)), 0).values
        rank = torch.arange(len(gs)) - first
        out_inds[order.to(out_inds.device)] = rank.to(out_inds.device)

    _mod("ingroup_indices", forward=_ingroup_forward)
    sst_ops = _load("mmdet3d.ops.sst.sst_ops", "mmdet3d/ops/sst/sst_ops.py")
    if not torch.cuda.is_available():
        _orig = sst_ops.IngroupIndicesFunction.forward

        def _cpu_forward(ctx, group_inds):
            out = torch.zeros_like(group_inds) - 1
            _ingroup_forward(group_inds, out)
            ctx.mark_non_differentiable(out)
            return out
        sst_ops.IngroupIndicesFunction.forward = staticmethod(_cpu_forward)
        sst_ops.get_inner_win_inds = sst_ops.IngroupIndicesFunction.apply
    for n in ("flat2window_v2", "window2flat_v2", "make_continuous_inds", "get_flat2win_inds_v2", "get_window_coors",
              "get_inner_win_inds", "flat2window", "window2flat", "get_flat2win_inds"):
        setattr(ops, n, getattr(sst_ops, n))
    m = {}
    m["sst_ops"] = sst_ops
    m["sst_input_layer_v2"] = _load("mmdet3d.models.sst.sst_input_layer_v2", "mmdet3d/models/sst/sst_input_layer_v2.py")
    m["sst_basic_block_v2"] = _load("mmdet3d.models.sst.sst_basic_block_v2", "mmdet3d/models/sst/sst_basic_block_v2.py")
    m["sst_v2"] = _load("mmdet3d.models.backbones.sst_v2", "mmdet3d/models/backbones/sst_v2.py")
    m["second"] = _load("mmdet3d.models.backbones.second", "mmdet3d/models/backbones/second.py")
    m["msda_fn"] = _load("mmdet3d.models.middle_encoders.multi_scale_deformable_attn_function",
                         "mmdet3d/models/middle_encoders/multi_scale_deformable_attn_function.py")
    m["fusion_encoder"] = _load("mmdet3d.models.middle_encoders.fusion_encoder",
                                "mmdet3d/models/middle_encoders/fusion_encoder.py")
    m["ms_deform_core"] = core
    # detection head: everything it imports beyond torch is loss / target / decoding machinery that
    # forward_single never touches -> inert stubs
    sys.modules["mmcv.cnn"].kaiming_init = lambda *a, **k: None
    core_mod = sys.modules["mmdet3d.core"]
    for n in ("circle_nms", "xywhr2xyxyr", "limit_period", "PseudoSampler", "Box3DMode", "LiDARInstance3DBoxes"):
        setattr(core_mod, n, None)
    _pkg("mmdet3d.core.bbox")
    _mod("mmdet3d.core.bbox.structures", rotation_3d_in_axis=None)
    regs["HEADS"] = _Registry("HEADS")
    builder.HEADS = regs["HEADS"]
    builder.build_loss = lambda cfg: None
    sys.modules["mmdet3d.models"].builder = builder
    _mod("mmdet3d.models.utils", clip_sigmoid=None)
    _mod("mmdet3d.models.fusion_layers", apply_3d_transformation=None)
    _pkg("mmdet3d.ops.iou3d")
    _mod("mmdet3d.ops.iou3d.iou3d_utils", nms_gpu=None)
    _mod("mmdet.core", build_bbox_coder=lambda cfg: None, multi_apply=None, build_assigner=lambda cfg: None,
         build_sampler=lambda *a, **k: None, AssignResult=None)
    # box decoding (get_bboxes): the coder is plain torch; its base class / registry are inert stubs
    regs["BBOX_CODERS"] = _Registry("BBOX_CODERS")
    _pkg("mmdet.core.bbox", BaseBBoxCoder=object)
    _mod("mmdet.core.bbox.builder", BBOX_CODERS=regs["BBOX_CODERS"])
    m["bbox_coder"] = _load("isf_ref_transfusion_bbox_coder", "mmdet3d/core/bbox/coders/transfusion_bbox_coder.py")
    _pkg("mmdet3d.models.dense_heads", os.path.join(base, "models", "dense_heads"))
    m["transfusion_head"] = _load("mmdet3d.models.dense_heads.transfusion_head_v2",
                                  "mmdet3d/models/dense_heads/transfusion_head_v2.py")
    _installed.update(m)
    return m


_pipelines = {}


def install_pipelines():
    """Load the reference's point-cloud input pipeline: datasets/pipelines/loading.py and
    transforms_3d.py with core/points/*.  Everything they import beyond numpy / torch is registry, image or box
    machinery the point transforms never touch -> inert stubs."""
    if _pipelines:
        return _pipelines
    install()
    base = os.path.join(REF, "mmdet3d")

    class FileClient:
        def __init__(self, backend="disk", **kw):
            pass

        def get(self, path):
            with open(path, "rb") as f:
                return f.read()

    mmcv = sys.modules["mmcv"]
    mmcv.FileClient = FileClient
    mmcv.is_tuple_of = lambda seq, t: isinstance(seq, tuple) and all(isinstance(x, t) for x in seq)
    reg = _Registry("PIPELINES")
    _pkg("mmdet.datasets")
    _mod("mmdet.datasets.builder", PIPELINES=reg)
    _mod("mmdet.datasets.pipelines", LoadAnnotations=object, LoadImageFromFile=object, RandomFlip=object)
    _pkg("mmdet3d.core.points", os.path.join(base, "core", "points"))
    pts = _load("mmdet3d.core.points", "mmdet3d/core/points/__init__.py")
    core = sys.modules["mmdet3d.core"]
    core.points, core.VoxelGenerator = pts, None
    bbox = sys.modules["mmdet3d.core.bbox"]
    for n in ("CameraInstance3DBoxes", "DepthInstance3DBoxes", "LiDARInstance3DBoxes", "box_np_ops"):
        setattr(bbox, n, None)
    _pkg("mmdet3d.datasets", os.path.join(base, "datasets"))
    _mod("mmdet3d.datasets.builder", OBJECTSAMPLERS=_Registry("OBJECTSAMPLERS"))
    _pkg("mmdet3d.datasets.pipelines", os.path.join(base, "datasets", "pipelines"))
    _mod("mmdet3d.datasets.pipelines.data_augment_utils", noise_per_object_v3_=None)
    _mod("torchvision")
    m = dict(points=pts)
    m["loading"] = _load("mmdet3d.datasets.pipelines.loading", "mmdet3d/datasets/pipelines/loading.py")
    m["transforms_3d"] = _load("mmdet3d.datasets.pipelines.transforms_3d",
                               "mmdet3d/datasets/pipelines/transforms_3d.py")
    _pipelines.update(m)
    return m


_voxel_encoders = {}


def install_voxel_encoders():
    """Load models/voxel_encoders/voxel_encoder.py (HardSimpleVFE: BASELINE configs[0]).  DynamicScatter has no CPU
    implementation in the reference and stays an inert name: only HardSimpleVFE is exercised."""
    if _voxel_encoders:
        return _voxel_encoders
    install()
    base = os.path.join(REF, "mmdet3d")
    sys.modules["mmdet3d.ops"].DynamicScatter = None
    builder = sys.modules["mmdet3d.models.builder"]
    builder.VOXEL_ENCODERS = _Registry("VOXEL_ENCODERS")
    _pkg("mmdet3d.models.voxel_encoders", os.path.join(base, "models", "voxel_encoders"))
    _load("mmdet3d.models.voxel_encoders.utils", "mmdet3d/models/voxel_encoders/utils.py")
    _voxel_encoders["voxel_encoder"] = _load("mmdet3d.models.voxel_encoders.voxel_encoder",
                                             "mmdet3d/models/voxel_encoders/voxel_encoder.py")
    return _voxel_encoders


_sparse_encoder = {}


def install_sparse_encoder():
    """Load the reference's OWN Python layers of the LiDAR backbone -- the vendored spconv-1 Python package
    (ops/bevfusion-ops/spconv: conv.py, modules.py, structure.py, functional.py, ops.py), ops/sparse_block.py and
    models/middle_encoders/sparse_encoder.py -- on top of a stand-in for exactly ONE thing: the compiled module
    `sparse_conv_ext` (the native-op boundary this build replaces), whose two entry points the path uses are served
    by the CPU oracle (oracle.get_indice_pairs / oracle.indice_conv).  What this pins is everything ABOVE that
    boundary: SparseConvolution.forward (indice_key sharing, output-shape rule, SubM padding), SparseSequential,
    SparseBasicBlock, make_sparse_convmodule and SparseEncoder's stage wiring, plus the state-dict key names.
    mmdet's ResNet BasicBlock (absent third-party) is restated from its documented constructor: conv1 / norm1 /
    conv2 / norm2 / relu built through build_conv_layer / build_norm_layer."""
    if _sparse_encoder:
        return _sparse_encoder
    install()
    import numpy as np
    import oracle
    oracle.build()
    base = os.path.join(REF, "mmdet3d")
    conv_layers = _Registry("CONV_LAYERS")
    sys.modules["mmcv.cnn"].CONV_LAYERS = conv_layers

    def build_conv(cfg, *args, **kwargs):
        cfg = dict(cfg or dict(type="Conv2d"))
        t = cfg.pop("type")
        if t in conv_layers.module_dict:
            return conv_layers.module_dict[t](*args, **kwargs, **cfg)
        return build_conv_layer(dict(type=t, **cfg), *args, **kwargs)

    sys.modules["mmcv.cnn"].build_conv_layer = build_conv

    # ---- the compiled module, served by the oracle
    def get_indice_pairs_3d(indices, batch_size, out_shape, spatial_shape, ksize, stride, padding, dilation,
                            out_padding, subm, transpose):
        assert not transpose and list(dilation) == [1, 1, 1] and list(out_padding) == [0, 0, 0]
        out_idx, pairs, num = oracle.get_indice_pairs(indices.numpy().astype(np.int32), int(batch_size),
                                                      list(spatial_shape), list(ksize), list(stride), list(padding),
                                                      subm=bool(subm))
        return torch.from_numpy(out_idx), torch.from_numpy(pairs), torch.from_numpy(num)

    def indice_conv_fp32(features, filters, indice_pairs, indice_num, num_act_out, inverse, subm):
        assert not inverse
        y = oracle.indice_conv(features.detach().numpy(), filters.detach().numpy(), indice_pairs.numpy(),
                               indice_num.numpy(), int(num_act_out))
        return torch.from_numpy(y)

    pkg = "isf_ref_spconv"
    _pkg(pkg, os.path.join(base, "ops", "bevfusion-ops", "spconv"))
    _mod(pkg + ".sparse_conv_ext", get_indice_pairs_3d=get_indice_pairs_3d, indice_conv_fp32=indice_conv_fp32)
    sp = {n: _load(f"{pkg}.{n}", f"mmdet3d/ops/bevfusion-ops/spconv/{n}.py")
          for n in ("structure", "modules", "ops", "functional", "conv")}
    mops = sys.modules["mmcv.ops"]
    mops.SparseConvTensor = sp["structure"].SparseConvTensor
    mops.SparseSequential = sp["modules"].SparseSequential
    mops.SparseModule = sp["modules"].SparseModule

    class BasicBlock(nn.Module):
        expansion = 1

        def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style="pytorch", with_cp=False,
                     conv_cfg=None, norm_cfg=dict(type="BN"), dcn=None, plugins=None, init_cfg=None):
            super().__init__()
            self.norm1_name, norm1 = build_norm_layer(norm_cfg, planes, postfix=1)
            self.norm2_name, norm2 = build_norm_layer(norm_cfg, planes, postfix=2)
            self.conv1 = build_conv(conv_cfg, inplanes, planes, 3, stride=stride, padding=dilation, dilation=dilation,
                                    bias=False)
            self.add_module(self.norm1_name, norm1)
            self.conv2 = build_conv(conv_cfg, planes, planes, 3, padding=1, bias=False)
            self.add_module(self.norm2_name, norm2)
            self.relu = nn.ReLU(inplace=True)
            self.downsample, self.stride, self.dilation, self.with_cp = downsample, stride, dilation, with_cp

        @property
        def norm1(self):
            return getattr(self, self.norm1_name)

        @property
        def norm2(self):
            return getattr(self, self.norm2_name)

    _pkg("mmdet.models.backbones")
    _mod("mmdet.models.backbones.resnet", BasicBlock=BasicBlock, Bottleneck=BasicBlock)
    block = _load("mmdet3d.ops.sparse_block", "mmdet3d/ops/sparse_block.py")
    ops_pkg = sys.modules["mmdet3d.ops"]
    ops_pkg.SparseBasicBlock, ops_pkg.make_sparse_convmodule = block.SparseBasicBlock, block.make_sparse_convmodule
    enc = _load("mmdet3d.models.middle_encoders.sparse_encoder", "mmdet3d/models/middle_encoders/sparse_encoder.py")
    _sparse_encoder.update(spconv=sp, sparse_block=block, sparse_encoder=enc)
    return _sparse_encoder


_dynamic_vfe = {}


def install_dynamic_vfe():
    """Load the reference's DynamicVFE (models/voxel_encoders/voxel_encoder.py:287-547, utils.py) and its Python
    DynamicScatter wrapper (ops/voxel/scatter_points.py) on top of a stand-in for ONE compiled entry point:
    voxel_layer.dynamic_point_to_voxel_forward (GPU-only in the reference, voxelization.h:118), served by the CPU
    oracle's restatement (itself pinned by the brute-force reference of the reference's own test).  Pins the feature
    decoration (cluster centre, voxel centre, map_voxel_center_to_point), the layer stack and the key names."""
    if _dynamic_vfe:
        return _dynamic_vfe
    install()
    import numpy as np
    import oracle
    oracle.build()
    base = os.path.join(REF, "mmdet3d")

    def dynamic_point_to_voxel_forward(feats, coors, reduce_type):
        f, c, cmap, cnt = oracle.dynamic_scatter(feats.detach().numpy(), coors.numpy().astype(np.int32), reduce_type)
        return [torch.from_numpy(f), torch.from_numpy(c).to(coors.dtype), torch.from_numpy(cmap), torch.from_numpy(cnt)]

    def dynamic_point_to_voxel_backward(*a, **k):
        raise NotImplementedError("inference golden")

    pkg = "isf_ref_voxel"
    _pkg(pkg, os.path.join(base, "ops", "voxel"))
    _mod(pkg + ".voxel_layer", dynamic_point_to_voxel_forward=dynamic_point_to_voxel_forward,
         dynamic_point_to_voxel_backward=dynamic_point_to_voxel_backward)
    scatter = _load(pkg + ".scatter_points", "mmdet3d/ops/voxel/scatter_points.py")
    sys.modules["mmdet3d.ops"].DynamicScatter = scatter.DynamicScatter
    builder = sys.modules["mmdet3d.models.builder"]
    builder.VOXEL_ENCODERS = _Registry("VOXEL_ENCODERS")
    builder.build_fusion_layer = lambda cfg: None
    for stale in ("mmdet3d.models.voxel_encoders.utils", "mmdet3d.models.voxel_encoders.voxel_encoder"):
        sys.modules.pop(stale, None)          # (re)load against the DynamicScatter above
    _pkg("mmdet3d.models.voxel_encoders", os.path.join(base, "models", "voxel_encoders"))
    _load("mmdet3d.models.voxel_encoders.utils", "mmdet3d/models/voxel_encoders/utils.py")
    _dynamic_vfe["voxel_encoder"] = _load("mmdet3d.models.voxel_encoders.voxel_encoder",
                                          "mmdet3d/models/voxel_encoders/voxel_encoder.py")
    _dynamic_vfe["scatter_points"] = scatter
    return _dynamic_vfe


_detector = {}


class _AttrDict(dict):
    """mmcv ConfigDict stand-in: cfg.pts style attribute access"""
    __getattr__ = dict.get


def install_detector():
    """Load the reference's detector code -- models/detectors/isfusion.py (ISFusionDetector) over mvx_two_stage.py and
    base.py -- with every point-cloud sub-module being the reference's OWN class (DynamicVFE, SparseEncoder,
    ISFusionEncoder, SECONDV2, SECONDFPN; the Python Voxelization wrapper of ops/voxel/voxelize.py), on top of the
    stand-ins for the compiled ops only (voxel_layer.{dynamic,hard}_voxelize, dynamic_point_to_voxel_forward,
    sparse_conv_ext.*: all served by the CPU oracle, which is pinned op by op elsewhere).  mmdet's BaseDetector is an
    nn.Module with init_cfg.  Pins the GLUE of extract_pts_feat: batching / padding of coordinates, pillar layer
    parameters, kwargs hand-over, concatenation order, neck."""
    if _detector:
        return _detector
    m = dict(install())
    m.update(install_sparse_encoder())
    m.update(install_dynamic_vfe())
    import numpy as np
    import oracle
    base = os.path.join(REF, "mmdet3d")
    builder = sys.modules["mmdet3d.models.builder"]

    # ---- ops/voxel/voxelize.py over the oracle's voxelization (itself bit-exact against the reference C++)
    def dynamic_voxelize(points, coors, voxel_size, coors_range, ndim=3):
        coors.copy_(torch.from_numpy(oracle.dynamic_voxelize(points.numpy(), list(voxel_size), list(coors_range))))

    def hard_voxelize(points, voxels, coors, num_points_per_voxel, voxel_size, coors_range, max_points, max_voxels,
                      ndim=3, deterministic=True):
        v, c, n = oracle.hard_voxelize(points.numpy(), list(voxel_size), list(coors_range), int(max_points),
                                       int(max_voxels))
        k = v.shape[0]
        voxels[:k] = torch.from_numpy(v)
        coors[:k] = torch.from_numpy(c)
        num_points_per_voxel[:k] = torch.from_numpy(n)
        return k

    vl = sys.modules["isf_ref_voxel.voxel_layer"]
    vl.dynamic_voxelize, vl.hard_voxelize = dynamic_voxelize, hard_voxelize
    voxelize = _load("isf_ref_voxel.voxelize", "mmdet3d/ops/voxel/voxelize.py")
    sys.modules["mmdet3d.ops"].Voxelization = voxelize.Voxelization

    # ---- registries / builders the detector constructor calls
    necks, dets = _Registry("NECKS"), _Registry("DETECTORS")
    mm = sys.modules["mmdet.models"]
    mm.NECKS, mm.DETECTORS = necks, dets
    sys.modules["mmcv.cnn"].build_upsample_layer = lambda cfg, *a, **k: (
        nn.ConvTranspose2d(*a, **{kk: vv for kk, vv in k.items()}, bias=cfg.get("bias", False)))
    sys.modules["mmcv.runner"].auto_fp16 = _identity_decorator
    _pkg("mmdet3d.models.necks", os.path.join(base, "models", "necks"))
    m["second_fpn"] = _load("mmdet3d.models.necks.second_fpn", "mmdet3d/models/necks/second_fpn.py")
    venc = builder.VOXEL_ENCODERS

    def build_any(cfg):      # mmdet3d's MIDDLE_ENCODERS / FUSION_LAYERS / ... are one MODELS registry
        for reg in (builder.MIDDLE_ENCODERS, builder.FUSION_LAYERS):
            if cfg["type"] in reg.module_dict:
                return reg.build(cfg)
        raise KeyError(cfg["type"])

    builder.build_middle_encoder = build_any
    builder.build_voxel_encoder = lambda cfg: venc.build(cfg)
    builder.build_neck = lambda cfg: necks.build(cfg)
    builder.build_head = lambda cfg: None
    builder.build_fusion_layer = lambda cfg: None

    class BaseDetector(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()

    _pkg("mmcv.parallel", DataContainer=object)
    _pkg("mmdet.models.detectors", BaseDetector=BaseDetector)
    core = sys.modules["mmdet3d.core"]
    for n in ("Box3DMode", "Coord3DMode", "bbox3d2result", "merge_aug_bboxes_3d", "show_result"):
        setattr(core, n, None)
    sys.modules["mmdet.core"].multi_apply = None
    _pkg("mmdet3d.models.detectors", os.path.join(base, "models", "detectors"))
    _load("mmdet3d.models.detectors.base", "mmdet3d/models/detectors/base.py")
    _load("mmdet3d.models.detectors.mvx_two_stage", "mmdet3d/models/detectors/mvx_two_stage.py")
    m["isfusion"] = _load("mmdet3d.models.detectors.isfusion", "mmdet3d/models/detectors/isfusion.py")
    m["AttrDict"] = _AttrDict
    _detector.update(m)
    return m


if __name__ == "__main__":
    mods = install()
    print({k: v.__name__ for k, v in mods.items()})
