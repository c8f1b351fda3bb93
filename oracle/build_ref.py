"""Build the REFERENCE's own voxelization C++ (CPU path) in place -> oracle/_ref/  (TEST INFRA ONLY).

Sources are compiled where they lie under /root/reference (never copied into this repo):
    mmdet3d/ops/voxel/src/voxelization.cpp        (pybind module: hard_voxelize, dynamic_voxelize, ...)
    mmdet3d/ops/voxel/src/voxelization_cpu.cpp
    mmdet3d/ops/voxel/src/scatter_points_cpu.cpp
They need only libtorch headers, which this image ships; WITH_CUDA is left undefined, so the GPU
entry points are compiled out (voxelization.h:21).  No stand-in headers are written.

The vendored spconv-1.x tree (mmdet3d/ops/bevfusion-ops/spconv) is NOT built: it includes
<cuda_runtime_api.h> and <ATen/cuda/CUDAContext.h>, which this ROCm image lacks, and providing
stand-ins for them is not allowed -> "unbuildable here"; the sparse-conv oracle is pinned by the
dense-conv3d identity instead (tests/test_oracle.py: the oracle against tests/golden/spconv_dense_ref.npz).

Usage:  python oracle/build_ref.py        (about 1 min; only works where /root/reference exists)
"""
import os
import sys

REF = os.environ.get("ISF_REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
NAME = "isf_ref_voxel_layer"


def available():
    return os.path.isdir(os.path.join(REF, "mmdet3d", "ops", "voxel", "src"))


def build(verbose=False):
    if not available():
        raise RuntimeError(f"reference tree not found under {REF}")
    from torch.utils.cpp_extension import load
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(REF, "mmdet3d", "ops", "voxel", "src")
    return load(
        name=NAME,
        sources=[os.path.join(src, f) for f in
                 ("voxelization.cpp", "voxelization_cpu.cpp", "scatter_points_cpu.cpp")],
        extra_cflags=["-O2"],
        build_directory=OUT,
        verbose=verbose,
    )


def load_prebuilt():
    """Import oracle/_ref/isf_ref_voxel_layer.so if it was built earlier (authoring container only: oracle/_ref/ is
    listed in .gpurunignore, the GPU box never sees it -- the goldens it generated are what travels)."""
    import importlib.util
    import torch  # noqa: F401  (the extension links against libtorch)
    path = os.path.join(OUT, NAME + ".so")
    if not os.path.exists(path):
        return None
    spec = importlib.util.spec_from_file_location(NAME, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    m = build(verbose="-v" in sys.argv)
    print("built", m.__file__)
