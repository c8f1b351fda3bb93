"""Sparse-convolution module surface (drop-in for what the IS-Fusion path imports from ``spconv.pytorch`` /
``mmcv.ops`` / the vendored mmdet3d/ops/bevfusion-ops/spconv: SparseConvTensor, SparseModule,
SparseSequential, SubMConv3d, SparseConv3d).

Differences that matter to a maintainer:
  * the rulebook is an output-stationary neighbour table (``isf_build_rulebook``), cached on the tensor per
    (type, kernel, stride, padding, spatial shape) -- so SubM convs WITHOUT an indice_key (SparseBasicBlock,
    ops/sparse_block.py:100-115) share the level's rulebook too (same active set => identical rulebook);
  * weights live in the spconv-1 / mmcv layout [kD,kH,kW,Cin,Cout] (bevfusion-ops/spconv/conv.py:100);
    checkpoints in the spconv-2 layout [Cout,kD,kH,kW,Cin] are permuted on load
    (ops/spconv/overwrite_spconv/write_spconv2.py:62-124 does the opposite conversion);
  * autograd: SparseConvFunction (forward + dX / dW kernels) when a tensor requires grad.
"""
import ctypes
import math

import weakref

import numpy as np
import torch
from torch import nn

from . import _lib

# the HIP kernels compute in fp32: under torch.autocast (the reference trains with mixed precision) inputs are cast
# to fp32 on the way in and autocast is off inside forward / backward
_amp_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_amp_bwd = torch.amp.custom_bwd(device_type="cuda")


class SparseConvTensor:
    """structure.py:21-63 (same attributes / methods)."""

    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        self.features = features
        self.indices = indices if indices.dtype == torch.int32 else indices.int()
        self.spatial_shape = list(spatial_shape)
        self.batch_size = batch_size
        self.indice_dict = {}
        self.grid = grid
        self._rulebooks = {}

    @property
    def spatial_size(self):
        return int(np.prod(self.spatial_shape))

    def find_indice_pair(self, key):
        return self.indice_dict.get(key) if key is not None else None

    def replace_feature(self, new_features):
        out = SparseConvTensor(new_features, self.indices, self.spatial_shape, self.batch_size, self.grid)
        out.indice_dict = self.indice_dict
        out._rulebooks = self._rulebooks
        return out

    def dense(self, channels_first=True):
        """[B, C, D, H, W] (structure.py:49-59) through isf_sparse_to_dense_bev."""
        _lib.require_cuda(self.features)
        out = _SparseToDense.apply(self.features, self.indices, tuple(self.spatial_shape), self.batch_size)
        return out if channels_first else out.permute(0, 2, 3, 4, 1).contiguous()

    @property
    def sparity(self):
        return self.indices.shape[0] / np.prod(self.spatial_shape) / self.batch_size


class _SparseToDense(torch.autograd.Function):
    """dense() with a gradient: forward = isf_sparse_to_dense_bev, backward = the rows of the dense gradient at the
    active sites (what autograd derives for the reference's scatter_nd, structure.py:8-25)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, features, indices, spatial_shape, batch_size):
        D, H, W = spatial_shape
        C = features.size(1)
        out = torch.empty((batch_size, C * D, H, W), dtype=torch.float32, device=features.device)
        f, idx = features.detach().contiguous().float(), indices.contiguous()    # locals: must outlive the C call
        _lib.check(_lib.load().isf_sparse_to_dense_bev(_lib.ptr(f), _lib.ptr(idx), features.size(0), C,
                                                       batch_size, D, H, W, _lib.ptr(out), _lib.stream()),
                   "isf_sparse_to_dense_bev")
        ctx.save_for_backward(indices)
        return out.view(batch_size, C, D, H, W)

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad):
        idx = ctx.saved_tensors[0].long()
        return grad[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]].contiguous(), None, None, None


class Rulebook:
    """Output-stationary neighbour table of one conv geometry on one active set."""

    def __init__(self, nbr, stride, num_in, num_out, out_indices, out_shape):
        self.nbr, self.stride = nbr, stride
        self.num_in, self.num_out = num_in, num_out
        self.out_indices, self.out_shape = out_indices, out_shape


def build_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, subm):
    """isf_build_rulebook wrapper -> Rulebook."""
    _lib.require_cuda(indices)
    lib = _lib.load()
    n_in = indices.size(0)
    K = int(np.prod(ksize))
    dev = indices.device
    if subm:
        out_shape = list(spatial_shape)
        cap = n_in
    else:
        o = _lib.i3([0, 0, 0])
        _lib.check(lib.isf_conv_out_shape(_lib.i3(spatial_shape), _lib.i3(ksize), _lib.i3(stride),
                                          _lib.i3(padding), o))
        out_shape = list(o)
        # every input feeds at most prod(ceil(k/s)) outputs; the grid bounds it too
        per_in = int(np.prod([math.ceil(k / s) for k, s in zip(ksize, stride)]))
        cap = max(1, min(n_in * per_in, int(np.prod(out_shape)) * batch_size))
    nstride = lib.isf_nbr_stride(cap)
    nbr = torch.empty((K, nstride), dtype=torch.int32, device=dev)
    out_idx = indices if subm else torch.empty((cap, 4), dtype=torch.int32, device=dev)
    n_out = ctypes.c_int(0)
    _lib.check(lib.isf_build_rulebook(_lib.ptr(indices), n_in, batch_size, _lib.i3(spatial_shape),
                                      _lib.i3(ksize), _lib.i3(stride), _lib.i3(padding),
                                      _lib.CONV_SUBM if subm else _lib.CONV_SPARSE,
                                      None if subm else _lib.ptr(out_idx), cap, _lib.ptr(nbr), nstride,
                                      ctypes.byref(n_out), _lib.stream()), "isf_build_rulebook")
    m = n_out.value
    return Rulebook(nbr, nstride, n_in, m, out_idx if subm else out_idx[:m], out_shape)


def pack_filters(weight):
    """[kD,kH,kW,Cin,Cout] -> MFMA fragment order (isf_pack_filters)."""
    w = weight.detach().float().contiguous()
    K = int(np.prod(w.shape[:-2]))
    packed = torch.empty(w.numel(), dtype=torch.float32, device=w.device)
    lib = _lib.load()
    _lib.check(lib.isf_pack_filters(_lib.ptr(w), K, w.shape[-2], w.shape[-1], _lib.ptr(packed), _lib.stream()),
               "isf_pack_filters")
    return packed


F16X3_CHANNELS = (32, 64, 128, 256)


def f16x3_supported(c_in, c_out):
    return c_in in F16X3_CHANNELS and c_out in F16X3_CHANNELS


def pack_filters_f16x3(weight, transposed=False):
    """[kD,kH,kW,Cin,Cout] fp32 -> pre-split (hi|lo f16, power-of-two scaled) MFMA fragments + header.
    transposed: pack the per-tap TRANSPOSED filters ([..., Cout, Cin] as a Cout -> Cin convolution: the data gradient's)
    straight from `weight`, without a transposed copy."""
    w = weight.detach().float().contiguous()
    K = int(np.prod(w.shape[:-2]))
    lib = _lib.load()
    c_in, c_out = (w.shape[-1], w.shape[-2]) if transposed else (w.shape[-2], w.shape[-1])
    nbytes = lib.isf_packed_filter16_bytes(K, c_in, c_out)
    packed = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    if transposed:
        _lib.check(lib.isf_pack_filters_f16x3_transposed(_lib.ptr(w), K, c_in, c_out, _lib.ptr(packed), _lib.stream()),
                   "isf_pack_filters_f16x3_transposed")
    else:
        _lib.check(lib.isf_pack_filters_f16x3(_lib.ptr(w), K, c_in, c_out, _lib.ptr(packed), _lib.stream()),
                   "isf_pack_filters_f16x3")
    return packed


def to_split(x):
    """fp32 [N,C] -> split activation buffer (same bytes; opaque uint8 tensor)."""
    x = x.contiguous().float()
    out = torch.empty(x.numel() * 4, dtype=torch.uint8, device=x.device)
    _lib.check(_lib.load().isf_f32_to_split(_lib.ptr(x), x.numel(), _lib.ptr(out), _lib.stream()))
    return out


def from_split(xs, shape):
    out = torch.empty(shape, dtype=torch.float32, device=xs.device)
    _lib.check(_lib.load().isf_split_to_f32(_lib.ptr(xs), out.numel(), _lib.ptr(out), _lib.stream()))
    return out


def to_half(x):
    """fp32 [N, C] -> f16 rows (the f16 storage mode's format; opaque uint8 tensor of N*C*2 bytes)"""
    x = x.contiguous().float()
    out = torch.empty(x.numel() * 2, dtype=torch.uint8, device=x.device)
    _lib.check(_lib.load().isf_f32_to_half(_lib.ptr(x), x.numel(), _lib.ptr(out), _lib.stream()))
    return out


def from_half(xh, shape):
    out = torch.empty(shape, dtype=torch.float32, device=xh.device)
    _lib.check(_lib.load().isf_half_to_f32(_lib.ptr(xh), out.numel(), _lib.ptr(out), _lib.stream()))
    return out


def tile_order(rb, c_in, c_out, mode=0, dma=False):
    """int32 tile order of the launch sparse_conv_forward_f16x3 (dma=True: sparse_conv_forward_dma -- its launch plan
    differs) makes for this rulebook and channel shape (isf_sparse_conv_tile_order), or None when the launch is not a
    single resident round.  Cached on the rulebook per (shape, mode, kernel)."""
    cache = rb.__dict__.setdefault("_tile_order", {})
    mode = int(mode) | (2048 if dma else 0)
    key = (int(c_in), int(c_out), mode)
    if key not in cache:
        lib = _lib.load()
        K = rb.nbr.numel() // rb.stride
        dev = rb.nbr.device
        work = torch.empty(8 * 255, dtype=torch.int32, device=dev)
        order = torch.empty(8 * 255, dtype=torch.int32, device=dev)
        n = ctypes.c_int(0)
        _lib.check(lib.isf_sparse_conv_tile_order(_lib.ptr(rb.nbr), rb.stride, K, rb.num_out, c_in, c_out, int(mode),
                                                  _lib.ptr(work), _lib.ptr(order), ctypes.byref(n), _lib.stream()),
                   "isf_sparse_conv_tile_order")
        cache[key] = (order[:n.value], work[:n.value]) if n.value else None
    return cache[key][0] if cache[key] is not None else None


def tile_table(rb, c_in, c_out, mode=0):
    """Equal-work tile table of the launch isf_sparse_conv_forward_f16x3 would make on this Rulebook for (c_in, c_out,
    mode) (isf_sparse_conv_tile_table), cached; None when the launch is not one resident round.  Opt-in: measured slower
    than uniform tiles + tile_order."""
    cache = rb.__dict__.setdefault("_tile_tables", {})
    key = (c_in, c_out, mode)
    if key not in cache:
        lib = _lib.load()
        K = rb.nbr.numel() // rb.stride
        dev = rb.nbr.device
        ng = (rb.num_out + 15) // 16
        scratch = torch.empty((2 * ng + 2,), dtype=torch.int32, device=dev)
        table = torch.zeros((8 * 2 * 3 * 64,), dtype=torch.int32, device=dev)     # parts x slots x 2 at most
        n = ctypes.c_int(0)
        _lib.check(lib.isf_sparse_conv_tile_table(_lib.ptr(rb.nbr), rb.stride, K, rb.num_out, c_in, c_out, int(mode),
                                                  _lib.ptr(scratch), _lib.ptr(table), ctypes.byref(n), _lib.stream()),
                   "isf_sparse_conv_tile_table")
        cache[key] = table[:n.value] if n.value else None
    return cache[key]


def tile_table_host(work, part_groups, parts, cus=32, wgs_per_cu=3, groups_per_tile=8):
    """isf_sparse_conv_tile_table_host: conv16_table_part on the CPU -> (tiles [parts, wgs * cus, 2], fits)."""
    work = np.ascontiguousarray(work, dtype=np.int32)
    tiles = np.zeros((parts, wgs_per_cu * cus, 2), dtype=np.int32)
    fits = ctypes.c_int(0)
    _lib.check(_lib.load().isf_sparse_conv_tile_table_host(work.ctypes.data, len(work), part_groups, parts, cus, wgs_per_cu,
                                                           groups_per_tile, tiles.ctypes.data, ctypes.byref(fits)),
               "isf_sparse_conv_tile_table_host")
    return tiles, bool(fits.value)


def sparse_conv_forward_f16x3(features, packed16, K, c_in, c_out, rb, scale=None, shift=None, residual=None,
                              relu=False, mode=0, order=None, table=None):
    """fp32 in / fp32 out convenience wrapper around the split-precision kernel (converts at both ends); mode 257 (f16
    storage) converts through f16 rows instead of split rows.  order: tile_order(rb, c_in, c_out, mode) or None."""
    _lib.require_cuda(features)
    f16io = (mode & ~32) == 257
    xs = to_half(features) if f16io else to_split(features)
    rs = None if residual is None else (to_half(residual) if f16io else to_split(residual))
    ys = torch.empty(rb.num_out * c_out * (2 if f16io else 4), dtype=torch.uint8, device=features.device)
    lib = _lib.load()
    args = (_lib.ptr(xs), rb.num_in, c_in, _lib.ptr(packed16), K, c_out, _lib.ptr(rb.nbr), rb.stride, rb.num_out,
            _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(rs), int(bool(relu)), _lib.ptr(ys), int(mode))
    if table is not None:
        _lib.check(lib.isf_sparse_conv_forward_f16x3_tiled(*args, _lib.ptr(table), _lib.stream()),
                   "isf_sparse_conv_forward_f16x3_tiled")
    elif order is None:
        _lib.check(lib.isf_sparse_conv_forward_f16x3(*args, _lib.stream()), "isf_sparse_conv_forward_f16x3")
    else:
        _lib.check(lib.isf_sparse_conv_forward_f16x3_ordered(*args, _lib.ptr(order), _lib.stream()),
                   "isf_sparse_conv_forward_f16x3_ordered")
    return from_half(ys, (rb.num_out, c_out)) if f16io else from_split(ys, (rb.num_out, c_out))


def sparse_conv_forward_dma(features, packed16, K, c_in, c_out, rb, scale=None, shift=None, residual=None, relu=False,
                            mode=0, order=None):
    """sparse_conv_forward_f16x3 on the LDS-DMA gather kernel of the narrow layers (isf_sparse_conv_forward_dma:
    c_in, c_out in {32, 64}); bit-identical results."""
    _lib.require_cuda(features)
    f16io = (mode & ~32) == 257
    xs = to_half(features) if f16io else to_split(features)
    rs = None if residual is None else (to_half(residual) if f16io else to_split(residual))
    ys = torch.empty(rb.num_out * c_out * (2 if f16io else 4), dtype=torch.uint8, device=features.device)
    _lib.check(_lib.load().isf_sparse_conv_forward_dma(
        _lib.ptr(xs), rb.num_in, c_in, _lib.ptr(packed16), K, c_out, _lib.ptr(rb.nbr), rb.stride, rb.num_out,
        _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(rs), int(bool(relu)), _lib.ptr(ys), int(mode), _lib.ptr(order),
        _lib.stream()), "isf_sparse_conv_forward_dma")
    return from_half(ys, (rb.num_out, c_out)) if f16io else from_split(ys, (rb.num_out, c_out))


def rulebook_lines(rb, taps_per_line=3):
    """(lines int32 [K / tpl, stride], mask int32 [stride], flag int32 [1]) -- the LINE-COMPRESSED form of a Rulebook's
    neighbour table (isf_rulebook_to_lines), cached on it; flag != 0: the table is not in rank order and has no such form."""
    if getattr(rb, "_lines", None) is None:
        K = rb.nbr.numel() // rb.stride
        dev = rb.nbr.device
        lines = torch.empty((K // taps_per_line, rb.stride), dtype=torch.int32, device=dev)
        mask = torch.empty((rb.stride,), dtype=torch.int32, device=dev)
        flag = torch.zeros((1,), dtype=torch.int32, device=dev)
        _lib.check(_lib.load().isf_rulebook_to_lines(_lib.ptr(rb.nbr), rb.stride, K, taps_per_line, rb.num_out,
                                                     _lib.ptr(lines), _lib.ptr(mask), _lib.ptr(flag), _lib.stream()),
                   "isf_rulebook_to_lines")
        rb._lines = (lines, mask, flag)
    return rb._lines


def lines_to_nbr(lines, mask, K, taps_per_line=3):
    """isf_lines_to_rulebook: the dense table [K, stride] a line-compressed one stands for."""
    stride = mask.numel()
    nbr = torch.empty((K, stride), dtype=torch.int32, device=mask.device)
    _lib.check(_lib.load().isf_lines_to_rulebook(_lib.ptr(lines), _lib.ptr(mask), stride, K, taps_per_line, _lib.ptr(nbr),
                                                 _lib.stream()), "isf_lines_to_rulebook")
    return nbr


def sparse_conv_forward_dma_lines(features, packed16, K, c_in, c_out, rb, scale=None, shift=None, residual=None,
                                  relu=False, mode=0, taps_per_line=3):
    """sparse_conv_forward_dma reading the line-compressed table (isf_sparse_conv_forward_dma_lines); bit-identical."""
    _lib.require_cuda(features)
    f16io = (mode & ~32) == 257
    xs = to_half(features) if f16io else to_split(features)
    rs = None if residual is None else (to_half(residual) if f16io else to_split(residual))
    ys = torch.empty(rb.num_out * c_out * (2 if f16io else 4), dtype=torch.uint8, device=features.device)
    lines, mask, _flag = rulebook_lines(rb, taps_per_line)
    _lib.check(_lib.load().isf_sparse_conv_forward_dma_lines(
        _lib.ptr(xs), rb.num_in, c_in, _lib.ptr(packed16), K, taps_per_line, c_out, _lib.ptr(lines), _lib.ptr(mask),
        rb.stride, rb.num_out, _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(rs), int(bool(relu)), _lib.ptr(ys), int(mode),
        _lib.stream()), "isf_sparse_conv_forward_dma_lines")
    return from_half(ys, (rb.num_out, c_out)) if f16io else from_split(ys, (rb.num_out, c_out))


CU_CAP8_VARIANTS = (9, 15)      # isf_conv_cu_plan.variant values that work on units of <= 8 groups


def cu_plan(rb, variant=0):
    """Unit plan of the one-workgroup-per-CU kernel for a Rulebook (isf_sparse_conv_cu_plan), cached on it per unit shape
    (16 groups per unit; 8 for the two-workgroups-per-CU variants): (isf_conv_cu_plan struct, the int32 buffer it points
    into).  One plan serves every 256-column layer on the rulebook."""
    cap8 = int(variant) in CU_CAP8_VARIANTS
    key = "_cu_plan8" if cap8 else "_cu_plan"
    if getattr(rb, key, None) is None:
        lib = _lib.load()
        K = rb.nbr.numel() // rb.stride
        n = ctypes.c_size_t(0)
        _lib.check(lib.isf_sparse_conv_cu_plan_ints(rb.num_out, ctypes.byref(n)), "isf_sparse_conv_cu_plan_ints")
        buf = torch.zeros((n.value,), dtype=torch.int32, device=rb.nbr.device)
        plan = _lib.ConvCuPlan()
        plan.variant = 9 if cap8 else 0       # read by the planner: decides the unit shape
        _lib.check(lib.isf_sparse_conv_cu_plan(_lib.ptr(rb.nbr), rb.stride, K, rb.num_out, _lib.ptr(buf),
                                               ctypes.byref(plan), _lib.stream()), "isf_sparse_conv_cu_plan")
        setattr(rb, key, (plan, buf))
    return getattr(rb, key)


def cu_plan_units(rb, variant=0):
    """The plan's unit table as a CPU tensor [num_units, 2] = (first 16-row group, groups) and the group masks [groups]."""
    plan, buf = cu_plan(rb, variant)
    base = buf.data_ptr()
    n = int(buf[(plan.num_units - base) // 4].item())
    uo = (plan.units - base) // 4
    ng = (rb.num_out + 15) // 16
    return buf[uo:uo + 2 * n].view(n, 2).cpu(), buf[:ng].cpu()


def cu_plan_host(work, cus=256):
    """isf_sparse_conv_cu_plan_host: the plan arithmetic on the CPU (no device work): work [groups] int32 -> units
    [num_units, 2]."""
    import numpy as np
    lib = _lib.load()
    work = np.ascontiguousarray(work, dtype=np.int32)
    cap = lib.isf_sparse_conv_cu_max_units(len(work), cus)
    units = np.zeros((cap, 2), dtype=np.int32)
    n = ctypes.c_int(0)
    _lib.check(lib.isf_sparse_conv_cu_plan_host(work.ctypes.data, len(work), cus, units.ctypes.data, cap, ctypes.byref(n)),
               "isf_sparse_conv_cu_plan_host")
    return units[:n.value]


def sparse_conv_cu_supported(c_in, c_out):
    return c_out == 256 and c_in in (128, 256)


def sparse_conv_forward_cu(features, packed16, K, c_in, c_out, rb, scale=None, shift=None, residual=None, relu=False,
                           variant=0):
    """sparse_conv_forward_f16x3 (mode 0) on the one-workgroup-per-CU kernel of the 256-column layers
    (isf_sparse_conv_forward_cu; c_out = 256, c_in in {128, 256}); bit-identical results.  variant: isf_conv_cu_plan
    .variant (0 round 4's shape; 4 / 5 = 4 waves at prefetch depth 1 / 2, 6 / 7 = 8 waves at depth 1 / 2; 8 = 8 waves with the
    assembly multiply phase; 9 / 10 = two 4-wave workgroups per CU over units of <= 8 groups: valid results; 1 / 2 / 3,
    11-15 timing knock-outs)."""
    _lib.require_cuda(features)
    xs = to_split(features)
    rs = None if residual is None else to_split(residual)
    ys = torch.empty(rb.num_out * c_out * 4, dtype=torch.uint8, device=features.device)
    plan, _buf = cu_plan(rb, variant)
    plan = _lib.ConvCuPlan(plan.group_masks, plan.units, plan.num_units, plan.max_units, plan.num_out, int(variant), plan.cap)
    _lib.check(_lib.load().isf_sparse_conv_forward_cu(
        _lib.ptr(xs), rb.num_in, c_in, _lib.ptr(packed16), K, c_out, _lib.ptr(rb.nbr), rb.stride, rb.num_out,
        _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(rs), int(bool(relu)), _lib.ptr(ys), ctypes.byref(plan),
        _lib.stream()), "isf_sparse_conv_forward_cu")
    return from_split(ys, (rb.num_out, c_out))


def sparse_conv_forward_best(features, packed16, K, c_in, c_out, rb, scale=None, shift=None, residual=None, relu=False,
                             mode=0):
    """The kernel choice of isf_sparse_encoder_forward for one layer driven from Python (all choices give the same
    bits): the LDS-DMA gather kernel for the narrow shapes, the tile-order table for launches of one resident round
    (the one-workgroup-per-CU kernel of the 256-column shapes is an opt-in: measured slower)."""
    if c_in <= 64 and c_out <= 64 and (mode & ~32) in (0, 1, 257):
        return sparse_conv_forward_dma(features, packed16, K, c_in, c_out, rb, scale, shift, residual, relu, mode)
    order = tile_order(rb, c_in, c_out, mode) if (mode & ~32) in (0, 1, 257) else None
    return sparse_conv_forward_f16x3(features, packed16, K, c_in, c_out, rb, scale, shift, residual, relu, mode, order)


def sparse_conv_trace(xs, packed16, K, c_in, c_out, rb, scale=None, shift=None, residual_split=None, relu=False,
                      order=None):
    """DIAGNOSTIC (isf_sparse_conv_trace): one production launch of a 128 -> 128 / 256 -> 256 layer on split rows `xs`
    -> (out_split, trace int64 [workgroups, 8]): time stamps (100 MHz) at entry / after the prologue / after the
    multiply loop / at exit, steps, HW_ID, XCC_ID, first row | half tile << 32."""
    _lib.require_cuda(xs)
    lib = _lib.load()
    ys = torch.empty(rb.num_out * c_out * 4, dtype=torch.uint8, device=xs.device)
    cap = 8 * 255
    trace = torch.zeros((cap * 8,), dtype=torch.int64, device=xs.device)
    n = ctypes.c_int(0)
    _lib.check(lib.isf_sparse_conv_trace(_lib.ptr(xs), rb.num_in, c_in, _lib.ptr(packed16), K, c_out, _lib.ptr(rb.nbr),
                                         rb.stride, rb.num_out, _lib.ptr(scale), _lib.ptr(shift),
                                         _lib.ptr(residual_split), int(bool(relu)), _lib.ptr(ys), _lib.ptr(order),
                                         _lib.ptr(trace), cap, ctypes.byref(n), _lib.stream()), "isf_sparse_conv_trace")
    return ys, trace[:n.value * 8].view(n.value, 8)


def sparse_conv_dma_trace(xs, packed16, K, c_in, c_out, rb, scale=None, shift=None, residual_split=None, relu=False,
                          lines=True):
    """DIAGNOSTIC (isf_sparse_conv_dma_trace): one production launch of a narrow layer (c_in, c_out in {32, 64}) on split
    rows `xs` -> (out_split, trace int64 [workgroups, 16]): the eight columns of sparse_conv_trace + wave 0's shader-clock
    cycles at the per-step vmcnt(0) / at the barrier / in the read-and-issue section / in the multiply section, and of the
    read-and-issue section: the fragment reads' LDS round trip, index arithmetic + weight run (the rest: the row gathers)."""
    _lib.require_cuda(xs)
    lib = _lib.load()
    ys = torch.empty(rb.num_out * c_out * 4, dtype=torch.uint8, device=xs.device)
    table, mask, stride, nx = rb.nbr, None, rb.stride, 0
    if lines:
        lt = rulebook_lines(rb, 3)
        if lt is not None:
            table, mask, nx = lt[0], lt[1], 3
    cap = 8 * 1024
    trace = torch.zeros((cap * 16,), dtype=torch.int64, device=xs.device)
    n = ctypes.c_int(0)
    _lib.check(lib.isf_sparse_conv_dma_trace(_lib.ptr(xs), rb.num_in, c_in, _lib.ptr(packed16), K, nx, c_out, _lib.ptr(table),
                                             _lib.ptr(mask) if mask is not None else None, stride, rb.num_out,
                                             _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(residual_split), int(bool(relu)),
                                             _lib.ptr(ys), _lib.ptr(trace), cap, ctypes.byref(n), _lib.stream()),
               "isf_sparse_conv_dma_trace")
    return ys, trace[:n.value * 16].view(n.value, 16)


def sparse_conv_phase_trace(xs, packed16, K, c_in, c_out, rb, scale=None, shift=None, residual_split=None, relu=False,
                            order=None):
    """DIAGNOSTIC (isf_sparse_conv_phase_trace): sparse_conv_trace plus, per wave, shader-clock stamps of every step of
    the multiply loop -> (out_split, wg int64 [workgroups, 8], waves uint32 [workgroups, waves, 8 + 8 * 216]): header
    {clock lo, hi at loop entry, HW_ID, steps, tap masks of the wave's two row groups, of the workgroup, clock at loop
    exit} then (top, after vmcnt(0), after barrier, after load issue, after the index reads, after the gathers, 0, 0) per step.  tools/conv_phase_trace.py."""
    _lib.require_cuda(xs)
    lib = _lib.load()
    ys = torch.empty(rb.num_out * c_out * 4, dtype=torch.uint8, device=xs.device)
    nbytes = 8 * 255 * (64 + 8 * (8 + 8 * 216) * 4)
    trace = torch.zeros((nbytes // 8,), dtype=torch.int64, device=xs.device)
    n, nw, dpw = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(lib.isf_sparse_conv_phase_trace(
        _lib.ptr(xs), rb.num_in, c_in, _lib.ptr(packed16), K, c_out, _lib.ptr(rb.nbr), rb.stride, rb.num_out,
        _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(residual_split), int(bool(relu)), _lib.ptr(ys), _lib.ptr(order),
        _lib.ptr(trace), nbytes, ctypes.byref(n), ctypes.byref(nw), ctypes.byref(dpw), _lib.stream()),
        "isf_sparse_conv_phase_trace")
    wg = trace[:n.value * 8].view(n.value, 8)
    waves = trace[n.value * 8:].view(torch.int32)[:n.value * nw.value * dpw.value].view(n.value, nw.value, dpw.value)
    return ys, wg, waves


def stage_tables(rb):
    """(slots uint16 [K, stride], ulist int32 [stride / 64, cap], ucount int32 [stride / 64]) of a Rulebook
    (isf_rulebook_stage_tables), cached on it: the distinct input rows of every 64-row unit and where each table entry
    sits in that list -- what the LDS-staged conv kernel copies once per tile."""
    if getattr(rb, "_stage", None) is None:
        lib = _lib.load()
        K = rb.nbr.numel() // rb.stride
        units = rb.stride // lib.isf_stage_unit_rows()
        dev = rb.nbr.device
        slots = torch.empty((K, rb.stride), dtype=torch.int16, device=dev)
        ulist = torch.empty((units, lib.isf_stage_unit_cap()), dtype=torch.int32, device=dev)
        ucount = torch.empty((units,), dtype=torch.int32, device=dev)
        _lib.check(lib.isf_rulebook_stage_tables(_lib.ptr(rb.nbr), rb.stride, K, _lib.ptr(slots), _lib.ptr(ulist),
                                                 _lib.ptr(ucount), _lib.stream()), "isf_rulebook_stage_tables")
        rb._stage = (slots, ulist, ucount)
    return rb._stage


def sparse_conv_forward_staged(features, packed16, K, c_in, c_out, rb, scale=None, shift=None, residual=None,
                               relu=False, stage_rows=512, mode=0):
    """sparse_conv_forward_f16x3 with the tile's input rows staged in LDS (isf_sparse_conv_forward_staged)."""
    _lib.require_cuda(features)
    xs = to_split(features)
    rs = to_split(residual) if residual is not None else None
    ys = torch.empty(rb.num_out * c_out * 4, dtype=torch.uint8, device=features.device)
    slots, ulist, ucount = stage_tables(rb)
    lib = _lib.load()
    _lib.check(lib.isf_sparse_conv_forward_staged(
        _lib.ptr(xs), rb.num_in, c_in, _lib.ptr(packed16), K, c_out, _lib.ptr(slots), rb.stride, _lib.ptr(ulist),
        _lib.ptr(ucount), rb.num_out, _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(rs), int(bool(relu)), _lib.ptr(ys),
        int(stage_rows), int(mode), _lib.stream()), "isf_sparse_conv_forward_staged")
    return from_split(ys, (rb.num_out, c_out))


def sparse_conv_forward(features, packed, K, c_in, c_out, rb, scale=None, shift=None, residual=None, relu=False):
    """isf_sparse_conv_forward_packed wrapper (conv + optional folded BN / residual / ReLU)."""
    _lib.require_cuda(features)
    out = torch.empty((rb.num_out, c_out), dtype=torch.float32, device=features.device)
    lib = _lib.load()
    _lib.check(lib.isf_sparse_conv_forward_packed(
        _lib.ptr(features), rb.num_in, c_in, _lib.ptr(packed), K, c_out, _lib.ptr(rb.nbr), rb.stride,
        rb.num_out, _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(residual), int(bool(relu)), _lib.ptr(out),
        _lib.stream()), "isf_sparse_conv_forward_packed")
    return out


def sparse_conv_split(xs, packed16, K, c_in, c_out, rb, ordered=True, mode=0):
    """the f16x3 convolution on SPLIT rows in and out (no epilogue, no format passes): the kernel choice of
    sparse_conv_forward_best -- LDS-DMA gathers for the narrow shapes, tile-order table for one-round launches
    (ordered=False: tiles in launch order; the training path rebuilds its rulebooks every step and the three kernels that
    build a table -- 0.6 ms per step over the encoder -- cost more than the 2 % an ordered launch gains)."""
    lib = _lib.load()
    if not ordered:
        ys = torch.empty(rb.num_out * c_out * 4, dtype=torch.uint8, device=xs.device)
        args = (_lib.ptr(xs), rb.num_in, c_in, _lib.ptr(packed16), K, c_out, _lib.ptr(rb.nbr), rb.stride, rb.num_out,
                None, None, None, 0, _lib.ptr(ys), int(mode))   # mode 1: single-pass f16 (hi halves only)
        if c_in <= 64 and c_out <= 64:
            _lib.check(lib.isf_sparse_conv_forward_dma(*args, None, _lib.stream()), "isf_sparse_conv_forward_dma")
        else:
            _lib.check(lib.isf_sparse_conv_forward_f16x3(*args, _lib.stream()), "isf_sparse_conv_forward_f16x3")
        return ys
    assert mode == 0, "ordered launches: fp32-class mode only (the training path passes ordered=False)"
    ys = torch.empty(rb.num_out * c_out * 4, dtype=torch.uint8, device=xs.device)
    args = (_lib.ptr(xs), rb.num_in, c_in, _lib.ptr(packed16), K, c_out, _lib.ptr(rb.nbr), rb.stride, rb.num_out,
            None, None, None, 0, _lib.ptr(ys), 0)
    if c_in <= 64 and c_out <= 64:
        _lib.check(lib.isf_sparse_conv_forward_dma(*args, _lib.ptr(tile_order(rb, c_in, c_out, 0, dma=True)),
                                                   _lib.stream()), "isf_sparse_conv_forward_dma")
    else:
        order = tile_order(rb, c_in, c_out, 0)
        if order is None:
            _lib.check(lib.isf_sparse_conv_forward_f16x3(*args, _lib.stream()), "isf_sparse_conv_forward_f16x3")
        else:
            _lib.check(lib.isf_sparse_conv_forward_f16x3_ordered(*args, _lib.ptr(order), _lib.stream()),
                       "isf_sparse_conv_forward_f16x3_ordered")
    return ys


def pair_lists(rb):
    """(indice_pairs int32 [K, 2, cap], indice_num int32 [K], cap) of a Rulebook (isf_rulebook_pair_lists): the spconv-1
    interchange format the reference's backward walks, cached on the rulebook (every layer of a level shares it)."""
    if getattr(rb, "_pairs", None) is None:
        lib = _lib.load()
        K = rb.nbr.numel() // rb.stride
        cap = lib.isf_pair_list_capacity(rb.num_in, rb.num_out)
        pairs = torch.empty((K, 2, cap), dtype=torch.int32, device=rb.nbr.device)
        num = torch.empty((K,), dtype=torch.int32, device=rb.nbr.device)
        _lib.check(lib.isf_rulebook_pair_lists(_lib.ptr(rb.nbr), rb.stride, rb.num_out, K, cap, _lib.ptr(pairs),
                                               _lib.ptr(num), _lib.stream()), "isf_rulebook_pair_lists")
        rb._pairs = (pairs, num, cap)
    return rb._pairs


def grad_to_split(g):
    """fp32 gradient rows -> (split rows of g * s, scale float32 [2] = {s, 1 / s}): s the power of two that brings
    max|g| into [2^9, 2^10) (isf_grad_to_split; device-side, no host sync)."""
    g = g.contiguous().float()
    gs = torch.empty(g.numel() * 4, dtype=torch.uint8, device=g.device)
    sc = torch.empty(2, dtype=torch.float32, device=g.device)
    _lib.check(_lib.load().isf_grad_to_split(_lib.ptr(g), g.numel(), _lib.ptr(gs), _lib.ptr(sc), _lib.stream()),
               "isf_grad_to_split")
    return gs, sc


def from_split_scaled(xs, shape, mul):
    """split rows -> fp32 * mul[0] (mul: device float tensor)"""
    out = torch.empty(shape, dtype=torch.float32, device=xs.device)
    _lib.check(_lib.load().isf_split_to_f32_scaled(_lib.ptr(xs), out.numel(), _lib.ptr(mul), _lib.ptr(out),
                                                   _lib.stream()), "isf_split_to_f32_scaled")
    return out


def sparse_conv_backward_filter_f16x3(xs, c_in, gs, c_out, rb, inv_scale, wshape, mode=0):
    """dW [*wshape] on the f16 matrix cores (isf_sparse_conv_backward_filter_f16x3): xs / gs split rows of the layer's
    input / of its scaled output gradient, inv_scale a device float (1 / the gradient's scale); mode 1: single-pass f16."""
    pairs, num, cap = pair_lists(rb)
    K = pairs.shape[0]
    grad_w = torch.empty(wshape, dtype=torch.float32, device=xs.device)
    _lib.check(_lib.load().isf_sparse_conv_backward_filter_f16x3(
        _lib.ptr(xs), rb.num_in, c_in, _lib.ptr(gs), rb.num_out, c_out, _lib.ptr(pairs), _lib.ptr(num), cap, K,
        _lib.ptr(inv_scale), _lib.ptr(grad_w), int(mode), _lib.stream()), "isf_sparse_conv_backward_filter_f16x3")
    return grad_w


# under torch.autocast the reference's sparse convolutions compute in half (functional.py:24 custom_fwd(cast_inputs=torch.half)
# -> indice_conv_half / indice_conv_backward_half); True: ours run the single-pass f16 kernels there (fp16 operands, fp32
# accumulation, fp32 rows in and out), False: fp32-class f16x3 arithmetic under autocast too
AUTOCAST_HALF = True

# round 5: dW on the f16 matrix cores + the gradient split once per layer (False: the round-2 path -- fp32-MFMA dW,
# torch-op gradient scaling -- kept as the cross-check the tests compare against)
WGRAD_F16X3 = True

# Backward kernels (isf_spconv_bwd.hip): validated on an MI355X in round 2 (tests/test_gpu_widened.py: dX / dW against the
# C restatement of indice_conv_backward on the golden geometries, every channel shape, a two-layer training step against
# dense torch autograd) and on by default since.  False restores the loud NotImplementedError for autograd calls.
TRAINING_KERNELS = True


def transposed_nbr(rb):
    """nbr_t [K, stride_t] of a Rulebook (isf_transpose_rulebook), cached on it: input row -> output row per tap."""
    if getattr(rb, "nbr_t", None) is None:
        lib = _lib.load()
        K = rb.nbr.numel() // rb.stride
        st = lib.isf_nbr_stride(max(rb.num_in, 1))
        nbr_t = torch.empty((K, st), dtype=torch.int32, device=rb.nbr.device)
        _lib.check(lib.isf_transpose_rulebook(_lib.ptr(rb.nbr), rb.stride, rb.num_out, K, rb.num_in, _lib.ptr(nbr_t),
                                              st, _lib.stream()), "isf_transpose_rulebook")
        rb.nbr_t, rb.stride_t = nbr_t, st
    return rb.nbr_t, rb.stride_t


def _f16x3_shape(c_in, c_out):
    return c_in in (32, 64, 128, 256) and c_out in (32, 64, 128, 256)


class _TransposedRulebook:
    """the view of a Rulebook the dX pass needs: rows of the conv's OUTPUT feed rows of its INPUT"""

    def __init__(self, nbr_t, stride_t, num_out, num_in):
        self.nbr, self.stride, self.num_in, self.num_out = nbr_t, stride_t, num_out, num_in


_PACKED_PAIRS = {}   # id(weight parameter) -> (version, data_ptr, packed forward filters, packed transposed filters, weakref)


# The cache below is keyed on (parameter identity, Tensor._version, data_ptr).  torch.optim steps, load_state_dict and
# every in-place op on the parameter bump _version; writes through `.data` (p.data.copy_(), EMA swaps, fp16 copy-back of
# older mmcv optim wrappers) do NOT -- after such an update call drop_packed_pairs() (or set PACKED_PAIR_CACHE = False for
# the run: 5 more launches per layer and step), otherwise forward and dX keep multiplying with the old filters.
PACKED_PAIR_CACHE = True


def drop_packed_pairs():
    """Forget every cached (forward, transposed) packed-filter pair: call after updating sparse-conv weights through
    `.data` (which does not bump Tensor._version, the cache's change detector)."""
    _PACKED_PAIRS.clear()


def _packed_pair(weight, w, K, c_in, c_out):
    """(packed filters of the forward conv, packed per-tap TRANSPOSED filters of the dX conv) of a weight parameter,
    packed once per parameter version: the forward pass packs both, the backward pass finds its half here instead of
    transposing + packing again (5 launches per layer and step)."""
    if not PACKED_PAIR_CACHE:
        return pack_filters_f16x3(w), pack_filters_f16x3(w, transposed=True)
    key = id(weight)
    hit = _PACKED_PAIRS.get(key)
    # the weak reference tells a live parameter from a new tensor that reuses a dead one's id / address / version 0
    if hit is not None and hit[4]() is weight and hit[0] == weight._version and hit[1] == weight.data_ptr():
        return hit[2], hit[3]
    pair = (pack_filters_f16x3(w), pack_filters_f16x3(w, transposed=True))    # (no transposed copy: the pack kernel reads it)
    if len(_PACKED_PAIRS) > 256:
        _PACKED_PAIRS.clear()
    _PACKED_PAIRS[key] = (weight._version, weight.data_ptr(), pair[0], pair[1], weakref.ref(weight))
    return pair


class SparseConvFunction(torch.autograd.Function):
    """SparseConvFunction / SubMConvFunction of the reference (ops/spconv/functional.py:22-97): forward =
    indice_conv, backward = indice_conv_backward -> (input_bp, filters_bp), on the HIP kernels.  `weight` is the
    module parameter [kD, kH, kW, Cin, Cout]."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, features, weight, rb, half=False):
        """half: single-pass f16 arithmetic (fp16 operands, fp32 accumulation) in forward, dX and dW -- the reference's
        SparseConvFunction is decorated custom_fwd(cast_inputs=torch.half) (bevfusion-ops/spconv/functional.py:24,46): under
        autocast its sparse convolutions run indice_conv_half / indice_conv_backward_half.  SparseConvolution.forward sets
        it from torch.is_autocast_enabled() when AUTOCAST_HALF is on; inputs and outputs stay fp32 rows either way."""
        _lib.require_cuda(features, weight)
        K = int(np.prod(weight.shape[:-2]))
        c_in, c_out = weight.shape[-2], weight.shape[-1]
        w = weight.detach().float().contiguous()
        features = features.detach().float().contiguous()
        xs = None
        if _f16x3_shape(c_in, c_out) and WGRAD_F16X3 and rb.num_out > 0 and rb.num_in > 0:
            # split rows once: the conv reads them, and so will dW in the backward pass (saved INSTEAD of the fp32 rows)
            xs = to_split(features)
            out = from_split(sparse_conv_split(xs, _packed_pair(weight, w, K, c_in, c_out)[0], K, c_in, c_out, rb, ordered=False,
                                               mode=1 if half else 0), (rb.num_out, c_out))
        elif _f16x3_shape(c_in, c_out):    # the inference kernel (f16x3 split MFMA): 3-4x the fp32-MFMA kernel's rate
            out = sparse_conv_forward_best(features, pack_filters_f16x3(w), K, c_in, c_out, rb)
        else:
            out = torch.empty((rb.num_out, c_out), dtype=torch.float32, device=features.device)
            _lib.check(_lib.load().isf_sparse_conv_forward(
                _lib.ptr(features), rb.num_in, c_in, _lib.ptr(w), K, c_out, _lib.ptr(rb.nbr), rb.stride, rb.num_out,
                None, None, None, 0, _lib.ptr(out), _lib.stream()), "isf_sparse_conv_forward")
        ctx.split_saved = xs is not None
        ctx.half = bool(half) and xs is not None
        ctx.packed_t = _packed_pair(weight, w, K, c_in, c_out)[1] if xs is not None else None   # dX's filters
        ctx.save_for_backward(xs if xs is not None else features, w)
        ctx.rb, ctx.wshape = rb, tuple(weight.shape)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_out):
        features, w = ctx.saved_tensors
        rb = ctx.rb
        K, c_in, c_out = int(np.prod(ctx.wshape[:-2])), ctx.wshape[-2], ctx.wshape[-1]
        g = grad_out.contiguous().float()
        lib = _lib.load()
        grad_in = grad_w = None
        if ctx.split_saved:
            # one scaled split of the gradient serves dX (the forward kernel over the transposed rulebook with the
            # transposed filters) and dW (isf_spconv_wgrad16.hip); the scale is a device scalar, undone in the last pass
            gs, sc = grad_to_split(g)
            if ctx.needs_input_grad[0]:
                nbr_t, st = transposed_nbr(rb)
                rbt = rb.__dict__.get("_rbt")
                if rbt is None:      # cached: its tile-order tables are built once per rulebook, not once per layer
                    rbt = rb._rbt = _TransposedRulebook(nbr_t, st, rb.num_out, rb.num_in)
                grad_in = from_split_scaled(sparse_conv_split(gs, ctx.packed_t, K, c_out, c_in, rbt, ordered=False,
                                                              mode=1 if ctx.half else 0),
                                            (rb.num_in, c_in), sc[1:])
            if ctx.needs_input_grad[1]:
                grad_w = sparse_conv_backward_filter_f16x3(features, c_in, gs, c_out, rb, sc[1:], ctx.wshape,
                                                           mode=1 if ctx.half else 0)
            return grad_in, grad_w, None, None
        if ctx.needs_input_grad[0]:
            nbr_t, st = transposed_nbr(rb)
            if _f16x3_shape(c_out, c_in):
                # dX[i] = sum_k g[nbr_t[k][i]] W_k^T: the forward kernel over the transposed rulebook with the
                # per-tap transposed filters.  Its operands travel as f16 hi + lo halves, so the gradient is brought
                # into f16's normal range by a power of two first (exact; see _lib.pow2_rescale) and scaled back
                gs, sc = _lib.pow2_rescale(g)
                wt = w.view(K, c_in, c_out).transpose(1, 2).contiguous().view(*ctx.wshape[:-2], c_out, c_in)
                rbt = _TransposedRulebook(nbr_t, st, rb.num_out, rb.num_in)
                grad_in = sparse_conv_forward_best(gs, pack_filters_f16x3(wt), K, c_out, c_in, rbt) / sc
            else:   # fp32-MFMA kernel: no f16 halves, gradients of any magnitude are safe
                grad_in = torch.empty((rb.num_in, c_in), dtype=torch.float32, device=g.device)
                _lib.check(lib.isf_sparse_conv_backward_input(_lib.ptr(g), rb.num_out, c_out, _lib.ptr(w), K, c_in,
                                                              _lib.ptr(nbr_t), st, rb.num_in, _lib.ptr(grad_in),
                                                              _lib.stream()), "isf_sparse_conv_backward_input")
        if ctx.needs_input_grad[1]:
            grad_w = torch.empty(ctx.wshape, dtype=torch.float32, device=g.device)
            _lib.check(lib.isf_sparse_conv_backward_filter(_lib.ptr(features), rb.num_in, c_in, _lib.ptr(g),
                                                           rb.num_out, c_out, _lib.ptr(rb.nbr), rb.stride, K,
                                                           _lib.ptr(grad_w), _lib.stream()),
                       "isf_sparse_conv_backward_filter")
        return grad_in, grad_w, None, None


class SparseModule(nn.Module):
    """marker base class (modules.py:38-41)."""


def _triple(v):
    return [int(v)] * 3 if isinstance(v, int) else [int(x) for x in v]


class SparseConvolution(SparseModule):
    """conv.py:41-223 surface: weight [k,k,k,Cin,Cout], optional bias, ``indice_key`` rulebook sharing."""

    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, subm=False, output_padding=0, transposed=False, inverse=False,
                 indice_key=None, fused_bn=False):
        super().__init__()
        assert ndim == 3 and groups == 1 and not transposed and not inverse, \
            "only the 3-D forward convs of the IS-Fusion path are built"
        assert _triple(dilation) == [1, 1, 1], "dilation 1 only"
        self.ndim = ndim
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _triple(kernel_size), _triple(stride)
        self.padding, self.dilation = _triple(padding), _triple(dilation)
        self.conv1x1 = int(np.prod(self.kernel_size)) == 1
        self.transposed, self.inverse = transposed, inverse
        self.output_padding = _triple(output_padding)
        self.groups, self.subm, self.indice_key, self.fused_bn = groups, subm, indice_key, fused_bn
        self.weight = nn.Parameter(torch.empty(*self.kernel_size, in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self._packed = None
        self._packed_key = None
        self._packed16 = None
        self._packed16_key = None
        self.reset_parameters()

    def reset_parameters(self):
        # conv.py:105-112: kaiming uniform with fan_in computed the torch way on this layout
        n = self.in_channels
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = n * int(np.prod(self.kernel_size))
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        key = prefix + "weight"
        if key in state_dict:
            w = state_dict[key]
            native = tuple(self.weight.shape)
            spconv2 = (self.out_channels, *self.kernel_size, self.in_channels)
            if tuple(w.shape) != native and tuple(w.shape) == spconv2:
                state_dict[key] = w.permute(1, 2, 3, 4, 0).contiguous()
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    def packed_weight(self):
        key = (self.weight._version, self.weight.data_ptr(), self.weight.device)
        if self._packed is None or self._packed_key != key:
            self._packed = pack_filters(self.weight)
            self._packed_key = key
        return self._packed

    def packed16_weight(self):
        """Split-precision packing, or None when (Cin, Cout) is outside the f16x3 kernel's shapes."""
        if not f16x3_supported(self.in_channels, self.out_channels):
            return None
        key = (self.weight._version, self.weight.data_ptr(), self.weight.device)
        if self._packed16 is None or self._packed16_key != key:
            self._packed16 = pack_filters_f16x3(self.weight)
            self._packed16_key = key
        return self._packed16

    def rulebook_for(self, x):
        key = ("subm" if self.subm else "conv", tuple(self.kernel_size), tuple(self.stride),
               tuple(self.padding), tuple(x.spatial_shape))
        if self.subm:
            key = ("subm", tuple(self.kernel_size), tuple(x.spatial_shape))
        rb = x._rulebooks.get(key)
        if rb is None:
            rb = build_rulebook(x.indices.contiguous(), x.batch_size, x.spatial_shape, self.kernel_size,
                                self.stride, self.padding, self.subm)
            x._rulebooks[key] = rb
            if self.indice_key is not None:
                x.indice_dict[self.indice_key] = rb
        return rb

    def forward(self, input):
        assert isinstance(input, SparseConvTensor)
        feats = input.features
        training = torch.is_grad_enabled() and (feats.requires_grad or self.weight.requires_grad)
        if training and not TRAINING_KERNELS:
            raise NotImplementedError(
                "isfusion_amd sparse conv: autograd is switched off (isfusion_amd.spconv.TRAINING_KERNELS = False); "
                "call under torch.no_grad() / module.eval() with requires_grad_(False), or switch it back on")
        rb = self.rulebook_for(input)
        if training:
            # training: the reference's SparseConvFunction / SubMConvFunction (functional.py:22-97) -- conv without
            # epilogue through autograd, the bias added by a stock broadcast (conv.py:209-210)
            half = AUTOCAST_HALF and torch.is_autocast_enabled()
            out_f = SparseConvFunction.apply(feats.contiguous().float(), self.weight, rb, half)
            if self.bias is not None:
                out_f = out_f + self.bias
            return self._wrap(input, out_f, rb)
        K = int(np.prod(self.kernel_size))
        shift = self.bias.detach().float() if self.bias is not None else None
        scale = torch.ones_like(shift) if shift is not None else None
        out_f = sparse_conv_forward(feats.contiguous().float(), self.packed_weight(), K, self.in_channels,
                                    self.out_channels, rb, scale, shift)
        return self._wrap(input, out_f, rb)

    def _wrap(self, input, out_f, rb):
        out = SparseConvTensor(out_f, rb.out_indices, rb.out_shape, input.batch_size)
        out.indice_dict = input.indice_dict
        if self.subm:
            out._rulebooks = input._rulebooks  # same active set
        out.grid = input.grid
        return out


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         True, indice_key=indice_key)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                         indice_key=indice_key)


def is_spconv_module(module):
    return isinstance(module, SparseModule)


class SparseSequential(SparseModule):
    """modules.py:44-139: applies sparse modules to the tensor and dense modules to ``.features``
    (skipped when there is no active voxel, :133-135)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], dict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError(f"index {idx} is out of range")
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, input):
        from .norm import bn1d_relu
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            module = mods[i]
            if is_spconv_module(module):
                assert isinstance(input, SparseConvTensor)
                input = module(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:
                    if (isinstance(module, nn.BatchNorm1d) and module.training and i + 1 < len(mods) and
                            isinstance(mods[i + 1], nn.ReLU)):
                        # norm -> act of a make_sparse_convmodule block in training mode: one fused pass (norm.bn1d_relu)
                        input = input.replace_feature(bn1d_relu(module, input.features))
                        i += 1
                    else:
                        input = input.replace_feature(module(input.features))
            else:
                input = module(input)
            i += 1
        return input
