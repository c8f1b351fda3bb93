"""Registry / config surface (SURVEY.md section 8b level 1): the type names configs/isfusion/isfusion_0075voxel.py uses
for the point-cloud path, mapped to this build's modules, plus a builder that takes the reference's config file
UNMODIFIED (an mmcv-style Python file) and returns the drop-in point-cloud path.

mmcv / mmdet3d are not importable in this image, so their registries cannot be filled here; `register_into` is what a
maintainer calls inside the reference tree: it registers the classes under the reference's
names with `force=True`, after which `build_model(cfg.model)` constructs these modules from the unmodified config."""
from torch import nn

from .detector import ISFusionPtsPath
from .fusion_encoder import ISFusionEncoder
from .fusion_modules import SECONDFPN, SECONDV2, SSTInputLayerV2, SSTv2
from .norm import NaiveSyncBatchNorm1d, NaiveSyncBatchNorm2d
from .sparse_encoder import SparseEncoder
from .spconv import SparseConv3d, SubMConv3d
from .transfusion_head import TransFusionHeadV2
from .voxel_encoder import DynamicVFE, HardSimpleVFE
from .voxelize import Voxelization

# reference registry -> {type name: class}; file:line = where the reference registers the name
MODULES = {
    "MODELS": {                       # mmdet3d.models.builder (VOXEL_ENCODERS / MIDDLE_ENCODERS / FUSION_LAYERS / HEADS ...)
        "DynamicVFE": DynamicVFE,                # voxel_encoders/voxel_encoder.py:287
        "HardSimpleVFE": HardSimpleVFE,          # voxel_encoders/voxel_encoder.py:13
        "SparseEncoder": SparseEncoder,          # middle_encoders/sparse_encoder.py:18
        "ISFusionEncoder": ISFusionEncoder,      # middle_encoders/fusion_encoder.py:833
        "SSTInputLayerV2": SSTInputLayerV2,      # sst/sst_input_layer_v2.py:18
        "TransFusionHeadV2": TransFusionHeadV2,  # dense_heads/transfusion_head_v2.py
    },
    "BACKBONES": {"SSTv2": SSTv2, "SECONDV2": SECONDV2},          # backbones/sst_v2.py:11, second.py:98
    "NECKS": {"SECONDFPN": SECONDFPN},                            # necks/second_fpn.py (stock conv stack)
    "NORM_LAYERS": {"naiveSyncBN1d": NaiveSyncBatchNorm1d, "naiveSyncBN2d": NaiveSyncBatchNorm2d},  # ops/norm.py:136,205
    "CONV_LAYERS": {"SubMConv3d": SubMConv3d, "SparseConv3d": SparseConv3d},   # ops/spconv (write_spconv2.py:20-36)
}
PLAIN_CLASSES = {"Voxelization": Voxelization}   # constructed directly (mvx_two_stage.py:43, isfusion.py:47)


def lookup(type_name):
    for table in MODULES.values():
        if type_name in table:
            return table[type_name]
    if type_name in PLAIN_CLASSES:
        return PLAIN_CLASSES[type_name]
    raise KeyError(f"'{type_name}' is not a module of the IS-Fusion point-cloud path built here "
                   f"(known: {sorted(n for t in MODULES.values() for n in t)})")


def build(cfg, **default_args):
    """mmcv.utils.build_from_cfg semantics for the names above: `type` selects the class, the rest are kwargs."""
    if isinstance(cfg, nn.Module):
        return cfg
    args = dict(cfg)
    args.update({k: v for k, v in default_args.items() if k not in args})
    return lookup(args.pop("type"))(**args)


def register_into(registries):
    """registries: {"MODELS": mmdet3d registry object, "BACKBONES": ..., ...} (objects with mmcv's
    `register_module(name=..., force=..., module=...)`).  Names not present in `registries` are skipped."""
    done = []
    for reg_name, table in MODULES.items():
        reg = registries.get(reg_name)
        if reg is None:
            continue
        for name, cls in table.items():
            reg.register_module(name=name, force=True, module=cls)
            done.append(f"{reg_name}.{name}")
    return done


def load_config(path):
    """Execute an mmcv-style Python config file (plain assignments, no `_base_` chain) -> dict of its variables."""
    ns = {}
    with open(path) as f:
        exec(compile(f.read(), path, "exec"), ns)   # a config file is code by definition (mmcv does the same)
    if "_base_" in ns:
        raise NotImplementedError("config inheritance (_base_) is mmcv's job; pass the resolved config dict instead")
    return {k: v for k, v in ns.items() if not k.startswith("__")}


def build_pts_path(config):
    """config: path of configs/isfusion/isfusion_0075voxel.py (unmodified), its variable dict, or its `model` dict
    -> ISFusionPtsPath with every sub-module built from the config's own kwargs (isfusion.py:21-51 and
    mvx_two_stage.py:36-75: the head receives test_cfg = model.test_cfg.pts)."""
    if isinstance(config, str):
        config = load_config(config)
    model = config["model"] if "model" in config else config
    assert model.get("type", "ISFusionDetector") == "ISFusionDetector", model.get("type")
    layer = model.get("pts_voxel_layer", {})
    assert layer.get("max_num_points", -1) == -1, "isfusion.py:123-146 voxelizes the fine grid dynamically"
    head = dict(model["pts_bbox_head"])
    test_cfg = (model.get("test_cfg") or {}).get("pts")
    if test_cfg is not None:
        head["test_cfg"] = dict(test_cfg)
    return ISFusionPtsPath(voxel_size=model["voxel_size"], pc_range=model["pc_range"],
                           out_size_factor=model.get("out_size_factor"),
                           pts_voxel_encoder=model["pts_voxel_encoder"], pts_middle_encoder=model["pts_middle_encoder"],
                           fusion_encoder=model["fusion_encoder"], pts_backbone=model["pts_backbone"],
                           pts_neck=model["pts_neck"], pts_bbox_head=head)
