"""isfusion_amd -- MI355X (gfx950) implementation of IS-Fusion's instance-scene fusion hot path.

The package directory is ``is-fusion_amd/`` (not an importable name); ``import isfusion_amd`` resolves to
it through the shim package ``isfusion_amd/`` at the repository root.  Everything numerical runs in
``libisf_hip.so`` (hand-written HIP kernels behind the C ABI of ``include/isf_hip.h``); these modules are the
host-side mirror of the reference's mmdet3d operator / module interface.  There is no CPU fallback.
"""
from . import _lib  # noqa: F401
from .fusion_encoder import ISFusionEncoder  # noqa: F401
from .fusion_modules import SECONDV2, SSTInputLayerV2, SSTv2  # noqa: F401
from .lidar_branch import ISFUSION_0075, LidarBranch  # noqa: F401
from .norm import NaiveSyncBatchNorm1d, NaiveSyncBatchNorm2d  # noqa: F401
from .scatter_points import DynamicScatter, dynamic_scatter  # noqa: F401
from .sparse_block import SparseBasicBlock, make_sparse_convmodule  # noqa: F401
from .sparse_encoder import SparseEncoder  # noqa: F401
from .spconv import (SparseConv3d, SparseConvTensor, SparseModule, SparseSequential,  # noqa: F401
                     SubMConv3d)
from .voxel_encoder import DynamicVFE, HardSimpleVFE  # noqa: F401
from .voxelize import Voxelization, voxelization  # noqa: F401

__version__ = "0.1.0"
