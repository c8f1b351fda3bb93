"""ctypes/numpy front-end of the C oracle (oracle/isf_oracle.c).  TEST INFRASTRUCTURE ONLY.

Each wrapper mirrors one reference op; the C function it calls cites the reference file:line.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libisf_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    """Compile oracle/isf_oracle.c with gcc (seconds)."""
    src = os.path.join(_HERE, "isf_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_hard_voxelize.restype = ctypes.c_int
        _lib.orc_dynamic_scatter.restype = ctypes.c_int
        _lib.orc_dynamic_vfe.restype = ctypes.c_int
        _lib.orc_get_indice_pairs.restype = ctypes.c_int
        _lib.orc_num_threads.restype = ctypes.c_int
    return _lib


def num_threads():
    """threads the conv loop of the oracle uses (OpenMP over the pairs of one tap; every other stage is scalar)"""
    return int(lib().orc_num_threads())


def _f(a):
    return a.ctypes.data_as(_f32p)


def _i(a):
    return a.ctypes.data_as(_i32p)


def _c3(v, t):
    return (t * len(v))(*v)


def _cf(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ci(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# ------------------------------------------------------------------------------------------ A1/A2
def dynamic_voxelize(points, voxel_size, coors_range):
    """-> coors [P,3] int32 (z,y,x), invalid rows (-1,-1,-1).  voxelization_cpu.cpp:8-43."""
    points = _cf(points)
    P, C = points.shape
    coors = np.zeros((P, 3), np.int32)
    lib().orc_dynamic_voxelize(_f(points), P, C, _c3(voxel_size, ctypes.c_float),
                               _c3(coors_range, ctypes.c_float), _i(coors))
    return coors


def hard_voxelize(points, voxel_size, coors_range, max_points, max_voxels):
    """-> (voxels [M,T,C], coors [M,3], num_points [M]).  voxelization_cpu.cpp:45-144."""
    points = _cf(points)
    P, C = points.shape
    voxels = np.zeros((max_voxels, max_points, C), np.float32)
    coors = np.zeros((max_voxels, 3), np.int32)
    npts = np.zeros((max_voxels,), np.int32)
    m = lib().orc_hard_voxelize(_f(points), P, C, _c3(voxel_size, ctypes.c_float),
                                _c3(coors_range, ctypes.c_float), int(max_points), int(max_voxels),
                                _f(voxels), _i(coors), _i(npts))
    assert m >= 0
    return voxels[:m].copy(), coors[:m].copy(), npts[:m].copy()


# --------------------------------------------------------------------------------------------- A3
_REDUCE = {"sum": 0, "mean": 1, "max": 2}


def dynamic_scatter(feats, coors, reduce_type="max"):
    """-> (voxel_feats [M,C], voxel_coors [M,3], point2voxel_map [P], count [M]).
    scatter_points_cuda.cu:183-239."""
    feats = _cf(feats)
    coors = _ci(coors)
    P, C = feats.shape
    if P == 0:
        return feats.copy(), coors.copy(), np.zeros((0,), np.int32), np.zeros((0,), np.int32)
    out_f = np.zeros((P, C), np.float32)
    out_c = np.zeros((P, 3), np.int32)
    cmap = np.zeros((P,), np.int32)
    cnt = np.zeros((P,), np.int32)
    m = lib().orc_dynamic_scatter(_f(feats), _i(coors), P, C, _REDUCE[reduce_type], _f(out_f),
                                  _i(out_c), _i(cmap), _i(cnt))
    return out_f[:m].copy(), out_c[:m].copy(), cmap, cnt[:m].copy()


def dynamic_scatter_backward(grad_reduced, feats, reduced, coors_map, count, reduce_type):
    """scatter_points_cuda.cu:241-308."""
    feats = _cf(feats)
    P, C = feats.shape
    M = reduced.shape[0]
    g = np.zeros((P, C), np.float32)
    lib().orc_dynamic_scatter_backward(_f(g), _f(_cf(grad_reduced)), _f(feats), _f(_cf(reduced)),
                                       _i(_ci(coors_map)), _i(_ci(count)), P, M, C,
                                       _REDUCE[reduce_type])
    return g


def dynamic_scatter_batched(feats, coors4, reduce_type):
    """DynamicScatter.forward for [P,4] (b,z,y,x) coors: per-sample loop + batch-index pad.
    scatter_points.py:75-96."""
    coors4 = _ci(coors4)
    B = int(coors4[-1, 0]) + 1
    fs, cs = [], []
    for b in range(B):
        m = coors4[:, 0] == b
        f, c, _, _ = dynamic_scatter(feats[m], coors4[m][:, 1:], reduce_type)
        fs.append(f)
        cs.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
    return np.concatenate(fs, 0), np.concatenate(cs, 0)


def hard_simple_vfe(voxels, num_points, num_features):
    """voxel_encoder.py:28-45."""
    voxels = _cf(voxels)
    M, T, C = voxels.shape
    out = np.zeros((M, num_features), np.float32)
    lib().orc_hard_simple_vfe(_f(voxels), _i(_ci(num_points)), M, T, C, int(num_features), _f(out))
    return out


# --------------------------------------------------------------------------------------------- A4
def fold_bn(weight, bias, mean, var, eps):
    """eval BatchNorm -> y = x*scale + shift (fp32)."""
    scale = (_cf(weight) / np.sqrt(_cf(var) + np.float32(eps))).astype(np.float32)
    shift = (_cf(bias) - _cf(mean) * scale).astype(np.float32)
    return scale, shift


def dynamic_vfe(points, coors4, voxel_size, coors_range, w1, bn1, w2, bn2):
    """DynamicVFE.forward (cluster+voxel centre, max), voxel_encoder.py:453-547.
    bn = (scale, shift) folded.  -> (voxel_feats [N,c2], voxel_coors [N,4], pt2vox [P])."""
    points = _cf(points)
    coors4 = _ci(coors4)
    P, Cin = points.shape
    w1 = _cf(w1)
    w2 = _cf(w2)
    c1, c2 = w1.shape[0], w2.shape[0]
    assert w1.shape[1] == Cin + 6 and w2.shape[1] == 2 * c1
    vf = np.zeros((max(P, 1), c2), np.float32)
    vc = np.zeros((max(P, 1), 4), np.int32)
    p2v = np.zeros((max(P, 1),), np.int32)
    n = lib().orc_dynamic_vfe(_f(points), _i(coors4), P, Cin, _c3(voxel_size, ctypes.c_float),
                              _c3(coors_range, ctypes.c_float), _f(w1), _f(_cf(bn1[0])),
                              _f(_cf(bn1[1])), c1, _f(w2), _f(_cf(bn2[0])), _f(_cf(bn2[1])), c2,
                              _f(vf), _i(vc), _i(p2v))
    return vf[:n].copy(), vc[:n].copy(), p2v[:P].copy()


# ------------------------------------------------------------------------------------------ A5/A6
def conv_out_shape(in_shape, ksize, stride, padding, dilation=(1, 1, 1)):
    """ops.py:20-31."""
    out = (ctypes.c_int * 3)()
    lib().orc_conv_out_shape(_c3(in_shape, ctypes.c_int), _c3(ksize, ctypes.c_int),
                             _c3(stride, ctypes.c_int), _c3(padding, ctypes.c_int),
                             _c3(dilation, ctypes.c_int), out)
    return [int(v) for v in out]


def get_indice_pairs(indices, batch_size, spatial_shape, ksize, stride, padding,
                     dilation=(1, 1, 1), subm=False):
    """-> (out_indices [N_out,4], indice_pairs [K,2,N], indice_num [K]).  geometry.h:24-297."""
    indices = _ci(indices)
    assert tuple(dilation) == (1, 1, 1), "oracle restates dilation 1 only (all the config uses)"
    N = indices.shape[0]
    K = int(np.prod(ksize))
    out_shape = list(spatial_shape) if subm else conv_out_shape(spatial_shape, ksize, stride, padding)
    cap = N if subm else min(N * K, int(np.prod(out_shape)) * batch_size)
    out_idx = np.zeros((max(cap, 1), 4), np.int32)
    pairs = np.zeros((K, 2, max(N, 1)), np.int32)
    num = np.zeros((K,), np.int32)
    if N == 0:
        return out_idx[:0], pairs[:, :, :0] - 1, num
    n_out = lib().orc_get_indice_pairs(_i(indices), N, int(batch_size),
                                       _c3(spatial_shape, ctypes.c_int), _c3(ksize, ctypes.c_int),
                                       _c3(stride, ctypes.c_int), _c3(padding, ctypes.c_int),
                                       _c3(dilation, ctypes.c_int), int(bool(subm)), _i(out_idx),
                                       _i(pairs), _i(num))
    return out_idx[:n_out].copy(), pairs, num


def indice_conv(features, filters, indice_pairs, indice_num, num_act_out):
    """filters [kD,kH,kW,Cin,Cout] (spconv1 layout).  spconv_ops.h:260-361."""
    features = _cf(features)
    filters = _cf(filters)
    Cin, Cout = filters.shape[-2], filters.shape[-1]
    K = int(np.prod(filters.shape[:-2]))
    N = features.shape[0]
    assert indice_pairs.shape == (K, 2, max(N, 1)) or indice_pairs.shape == (K, 2, N)
    out = np.zeros((num_act_out, Cout), np.float32)
    if N == 0 or num_act_out == 0:
        return out
    lib().orc_indice_conv(_f(features), N, Cin, _f(filters), K, Cout, _i(_ci(indice_pairs)),
                          _i(_ci(indice_num)), int(num_act_out), _f(out))
    return out


def indice_conv_backward(features, filters, out_bp, indice_pairs, indice_num):
    """spconv_ops.h:363-456 -> (input_bp, filters_bp)."""
    features = _cf(features)
    filters = _cf(filters)
    Cin, Cout = filters.shape[-2], filters.shape[-1]
    K = int(np.prod(filters.shape[:-2]))
    N = features.shape[0]
    in_bp = np.zeros((N, Cin), np.float32)
    f_bp = np.zeros(filters.shape, np.float32)
    lib().orc_indice_conv_backward(_f(features), N, Cin, _f(filters), K, Cout, _f(_cf(out_bp)),
                                   _i(_ci(indice_pairs)), _i(_ci(indice_num)), _f(in_bp), _f(f_bp))
    return in_bp, f_bp


def bn_act(x, scale, shift, residual=None, relu=True):
    x = _cf(x).copy()
    N, C = x.shape
    r = None if residual is None else _f(_cf(residual))
    lib().orc_bn_act(_f(x), N, C, _f(_cf(scale)), _f(_cf(shift)), r, int(bool(relu)))
    return x


def dense_bev(features, indices, batch_size, spatial_shape):
    """SparseConvTensor.dense() + view(N, C*D, H, W): structure.py:49-59, sparse_encoder.py:133-136."""
    features = _cf(features)
    N, C = features.shape
    D, H, W = spatial_shape
    out = np.zeros((batch_size, C * D, H, W), np.float32)
    lib().orc_dense_bev(_f(features), _i(_ci(indices)), N, C, int(batch_size), D, H, W, _f(out))
    return out


# --------------------------------------------------------------------------------------------- A7
def sparse_encoder_forward(params, voxel_features, coors, batch_size):
    """SparseEncoder.forward (sparse_encoder.py:107-138) for block_type='basicblock' or
    'conv_module', eval-mode BN.

    ``params`` is the plain description produced by ``isfusion_amd.sparse_encoder.export_plan``:
      dict(sparse_shape=[D,H,W], layers=[layer...]) with layer =
        dict(kind='subm'|'spconv', ksize, stride, padding, weight[kD,kH,kW,Cin,Cout], scale, shift,
             relu, residual_from=None|int (index of the layer OUTPUT to add before ReLU, -1 = input))
    -> (spatial_features [B, C*D, H, W], list of per-layer (features, indices, shape))
    """
    feats = _cf(voxel_features)
    idx = _ci(coors)
    shape = list(params["sparse_shape"])
    outs = []
    cache = {}
    for L in params["layers"]:
        subm = L["kind"] == "subm"
        # SubM rulebooks at one level are identical (same active set), so they are built once.
        ckey = ("subm", tuple(shape), tuple(L["ksize"]))
        if subm and ckey in cache:
            out_idx, pairs, num = cache[ckey]
        else:
            out_idx, pairs, num = get_indice_pairs(idx, batch_size, shape, L["ksize"], L["stride"],
                                                   L["padding"], subm=subm)
            if subm:
                cache[ckey] = (out_idx, pairs, num)
        y = indice_conv(feats, L["weight"], pairs, num, out_idx.shape[0])
        res = None
        if L.get("residual_from") is not None:
            r = L["residual_from"]
            res = outs[r][0] if r >= 0 else _cf(voxel_features)
        y = bn_act(y, L["scale"], L["shift"], res, L["relu"])
        if not subm:
            shape = conv_out_shape(shape, L["ksize"], L["stride"], L["padding"])
            cache = {}
        feats, idx = y, out_idx
        outs.append((feats, idx, list(shape)))
    bev = dense_bev(feats, idx, batch_size, shape)
    return bev, outs
