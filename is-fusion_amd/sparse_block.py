"""SparseBasicBlock / make_sparse_convmodule (drop-in for mmdet3d/ops/sparse_block.py:79-199).

Parameter names match the reference (mmdet ``BasicBlock`` naming: conv1, bn1, conv2, bn2), so state dicts
are interchangeable.  ``conv_layer()`` / ``norm_layer()`` replace the mmcv registry lookups for the few
types the IS-Fusion config uses.
"""
from torch import nn

from .norm import bn1d_relu, build_norm_layer
from .spconv import SparseConv3d, SparseModule, SparseSequential, SubMConv3d

CONV_TYPES = {"SubMConv3d": SubMConv3d, "SparseConv3d": SparseConv3d}


def build_conv_layer(cfg, *args, **kwargs):
    cfg = dict(cfg)
    t = cfg.pop("type")
    if t not in CONV_TYPES:
        raise KeyError(f"conv type {t} is outside the IS-Fusion sparse path")
    return CONV_TYPES[t](*args, **kwargs, **cfg)


class SparseBasicBlock(SparseModule):
    """SubM -> BN -> ReLU -> SubM -> BN -> (+identity) -> ReLU   (sparse_block.py:117-134)."""

    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, conv_cfg=None, norm_cfg=None):
        super().__init__()
        assert stride == 1 and downsample is None, "only the stride-1 identity block is on the IS-Fusion path"
        conv_cfg = conv_cfg or dict(type="SubMConv3d")
        norm_cfg = norm_cfg or dict(type="BN1d", eps=1e-3, momentum=0.01)
        self.conv1 = build_conv_layer(conv_cfg, inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = build_norm_layer(norm_cfg, planes)[1]
        self.conv2 = build_conv_layer(conv_cfg, planes, planes, 3, padding=1, bias=False)
        self.bn2 = build_norm_layer(norm_cfg, planes)[1]
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    @property
    def norm1(self):
        return self.bn1

    @property
    def norm2(self):
        return self.bn2

    def forward(self, x):
        identity = x.features
        assert x.features.dim() == 2, f"x.features.dim()={x.features.dim()}"
        out = self.conv1(x)
        # bn1d_relu: relu(bn(x) + identity) -- one fused HIP pass per direction in training mode (norm.py), the stock
        # modules otherwise
        out = out.replace_feature(bn1d_relu(self.bn1, out.features))
        out = self.conv2(out)
        out = out.replace_feature(bn1d_relu(self.bn2, out.features, residual=identity))
        return out


def make_sparse_convmodule(in_channels, out_channels, kernel_size, indice_key, stride=1, padding=0,
                           conv_type="SubMConv3d", norm_cfg=None, order=("conv", "norm", "act")):
    """SparseSequential(conv[, norm][, act]) in the requested order (sparse_block.py:137-199)."""
    assert isinstance(order, tuple) and len(order) <= 3
    assert set(order) | {"conv", "norm", "act"} == {"conv", "norm", "act"}
    layers = []
    for layer in order:
        if layer == "conv":
            layers.append(build_conv_layer(dict(type=conv_type, indice_key=indice_key), in_channels,
                                           out_channels, kernel_size, stride=stride, padding=padding,
                                           bias=False))
        elif layer == "norm":
            layers.append(build_norm_layer(norm_cfg, out_channels)[1])
        elif layer == "act":
            layers.append(nn.ReLU(inplace=True))
    return SparseSequential(*layers)
