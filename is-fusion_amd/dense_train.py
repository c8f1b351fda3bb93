"""Dense 3x3 BEV convolutions + BatchNorm2d (batch statistics) + ReLU in TRAINING mode on the HIP kernels
(SURVEY.md 8f #2 / #4).

The reference trains these stacks on cuDNN: mmcv ConvModule (fusion_encoder.py:862-960: conv_fusion, conv_ins, conv_scene,
conv_heatmap, heatmap_head_1/2) and SECONDV2 (backbones/second.py:126-165).  Until round 6 the training step ran them on
MIOpen (bf16 igemm + Col2Im2dU + layout transposes + casts: 7.9 ms of a 48-ms step at 2 x 300 k points,
gpurun_out/r06_train).  Here a dense [B, H, W] grid is a sparse tensor with every cell active, as in the inference path
(dense_conv.py):

* forward  = the sparse-conv kernel over the cached arithmetic rulebook of the grid (isf_dense_grid_rulebook),
* dX       = the same kernel over the transposed rulebook with the per-tap transposed filters,
* dW       = isf_sparse_conv_backward_filter_f16x3 (f16 matrix cores) over the rulebook's pair lists,
* BN + ReLU = the fused BatchNorm kernels of norm.py on the [tokens, C] rows (channels-last BatchNorm2d IS BatchNorm1d
  over the token rows), synchronised across ranks for naiveSyncBN2d.

Activations travel through a stack as fp32 token rows ([B*H*W, C], token = (b*H + y)*W + x); a stack hands back an NCHW
*view* of its rows, so the token-major ops around it (SST, attention) pay no layout copy.  Arithmetic: f16x3 split
(fp32-class) by default, single-pass f16 under torch.autocast -- what the reference's autocast does to its convolutions
(bf16 there; f16 operands + fp32 accumulation here, like the sparse convolutions: spconv.AUTOCAST_HALF).

No CPU fallback: layers the kernels do not tile (channel counts outside 32 / 64 / 128 / 256 per <= 256-channel input group,
e.g. the 10-class heat-map conv) run on the stock module, everything else fails loudly without the HIP library.
"""
import weakref

import torch
from torch import nn

from . import _lib
from . import spconv as sp
from .dense_conv import grid_rulebook
from .norm import bn1d_relu

ENABLED = True        # False: every stack on the stock modules (the round-5 training path; A/B and cross-check)

_amp_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_amp_bwd = torch.amp.custom_bwd(device_type="cuda")

_rulebooks = {}
_packed = {}          # (id(weight), transpose) -> (version, data_ptr, [(offset, channels, fwd filters, dX filters)], weakref)


def dense_rulebook(device, B, H, W, stride):
    """spconv.Rulebook of a dense B x H x W grid under a 3x3 / pad 1 / `stride` convolution, cached per geometry together
    with what the backward passes hang on it (transposed table, pair lists)."""
    key = (str(device), B, H, W, stride)
    if key not in _rulebooks:
        nbr, nstride, oh, ow = grid_rulebook(device, B, H, W, stride, False)
        rb = sp.Rulebook(nbr, nstride, B * H * W, B * oh * ow, None, (oh, ow))
        _rulebooks[key] = rb
    return _rulebooks[key]


def supported(conv):
    """can this nn.Conv2d run on the kernels?"""
    if not (isinstance(conv, nn.Conv2d) and conv.kernel_size == (3, 3) and conv.padding == (1, 1) and
            conv.dilation == (1, 1) and conv.groups == 1 and conv.stride in ((1, 1), (2, 2)) and
            conv.padding_mode == "zeros" and conv.weight.is_cuda):
        return False
    cin, cout = conv.in_channels, conv.out_channels
    return cout in (32, 64, 128, 256) and cin >= 32 and cin % 256 in (0, 32, 64, 128)   # <= 256-channel input groups


def _groups(weight, transpose):
    """packed (forward, dX) filters per <= 256-channel input group of a Conv2d weight [Cout, Cin, 3, 3], once per
    parameter version (torch.optim steps bump Tensor._version; `.data` writes do not: spconv.drop_packed_pairs)."""
    key = (id(weight), bool(transpose))
    hit = _packed.get(key)
    if hit is not None and hit[3]() is weight and hit[0] == weight._version and hit[1] == weight.data_ptr():
        return hit[2]
    w = weight.detach().float()
    cout, cin = w.shape[:2]
    # taps enumerated (ky, kx); transpose: (kx, ky) = the convolution of the spatially transposed map on the
    # un-transposed tokens (dense_conv.PackedConvBN)
    w5 = (w.permute(3, 2, 1, 0) if transpose else w.permute(2, 3, 1, 0)).contiguous()      # [3, 3, Cin, Cout]
    groups = []
    off = 0
    while off < cin:
        c = min(256, cin - off)
        wk = w5[:, :, off:off + c].contiguous().view(1, 3, 3, c, cout)
        groups.append((off, c, sp.pack_filters_f16x3(wk), sp.pack_filters_f16x3(wk, transposed=True)))
        off += c
    if len(_packed) > 256:
        _packed.clear()
    _packed[key] = (weight._version, weight.data_ptr(), groups, weakref.ref(weight))
    return groups


def drop_packed():
    _packed.clear()


class DenseConvFunction(torch.autograd.Function):
    """rows [B*H*W, Cin] fp32, weight [Cout, Cin, 3, 3] -> rows [B*OH*OW, Cout] fp32 (no bias)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, rows, weight, rb, transpose, half):
        _lib.require_cuda(rows, weight)
        cout = weight.shape[0]
        groups = _groups(weight, transpose)
        mode = 1 if half else 0
        rows = rows.detach()
        saved = []
        out = None
        for off, c, pk, _ in groups:
            xs = sp.to_split(rows if len(groups) == 1 else rows[:, off:off + c])
            y = sp.from_split(sp.sparse_conv_split(xs, pk, 9, c, cout, rb, ordered=False, mode=mode), (rb.num_out, cout))
            out = y if out is None else out.add_(y)
            saved.append(xs)
        ctx.save_for_backward(*saved)
        ctx.weight_ref, ctx.version = weight, weight._version
        ctx.geom = (rb, bool(transpose), mode, tuple(weight.shape))
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_out):
        rb, transpose, mode, wshape = ctx.geom
        cout, cin = wshape[:2]
        weight = ctx.weight_ref
        assert weight._version == ctx.version, "conv weight modified between forward and backward"
        groups = _groups(weight, transpose)
        gs, sc = sp.grad_to_split(grad_out)
        grad_rows = grad_w = None
        if ctx.needs_input_grad[0]:
            nbr_t, st = sp.transposed_nbr(rb)
            rbt = rb.__dict__.get("_rbt")
            if rbt is None:
                rbt = rb._rbt = sp._TransposedRulebook(nbr_t, st, rb.num_out, rb.num_in)
            parts = [sp.from_split_scaled(sp.sparse_conv_split(gs, pkt, 9, cout, c, rbt, ordered=False, mode=mode),
                                          (rb.num_in, c), sc[1:]) for _, c, _, pkt in groups]
            grad_rows = parts[0] if len(parts) == 1 else torch.cat(parts, 1)
        if ctx.needs_input_grad[1]:
            # (+2: every tap list of a dense grid is full -- larger reduction chunks, a third of the partial-sum traffic)
            parts = [sp.sparse_conv_backward_filter_f16x3(xs, c, gs, cout, rb, sc[1:], (1, 3, 3, c, cout), mode=mode | 2)
                     for xs, (_, c, _, _) in zip(ctx.saved_tensors, groups)]
            g5 = (parts[0] if len(parts) == 1 else torch.cat(parts, 3)).view(3, 3, cin, cout)
            grad_w = (g5.permute(3, 2, 1, 0) if transpose else g5.permute(3, 2, 0, 1)).contiguous()
        return grad_rows, grad_w, None, None, None


def to_rows(x):
    """[B, C, H, W] (any strides) -> fp32 rows [B*H*W, C]; free when x is the NCHW view of token-major rows"""
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).float().contiguous()


def from_rows(rows, B, H, W):
    """NCHW VIEW of token-major rows (no copy)"""
    return rows.view(B, H, W, rows.shape[1]).permute(0, 3, 1, 2)


def _layers(seq):
    """nn.Sequential of (Conv2d [, BatchNorm2d] [, ReLU])* -> [(conv, bn | None, relu)] or None"""
    mods = list(seq)
    out = []
    i = 0
    while i < len(mods):
        conv = mods[i]
        if not isinstance(conv, nn.Conv2d):
            return None
        bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.modules.batchnorm._BatchNorm) else None
        j = i + (2 if bn is not None else 1)
        relu = j < len(mods) and isinstance(mods[j], nn.ReLU)
        out.append((conv, bn, relu))
        i = j + (1 if relu else 0)
    return out


def usable(seq):
    """does this stack run on the kernels in the current mode?"""
    if not (ENABLED and sp.TRAINING_KERNELS and sp.WGRAD_F16X3 and torch.is_grad_enabled()):
        return False
    ls = _layers(seq if isinstance(seq, nn.Sequential) else nn.Sequential(seq))
    return ls is not None and all(supported(c) for c, _, _ in ls)


def conv_stack(seq, x, transpose=False):
    """training-mode forward of an nn.Sequential of (Conv2d 3x3 [, BatchNorm2d] [, ReLU])* -- a ConvModule, a SECONDV2
    block, a single Conv2d -- on [B, C, H, W] (or a list of such maps = their channel concatenation); returns an NCHW view
    of token-major rows.  transpose=True: the stack
    applied to x.permute(0, 1, 3, 2), result permuted back (what the reference does around its instance branch,
    fusion_encoder.py:1093,1139) without transposing anything.  Stacks the kernels do not cover run on the stock modules."""
    single = not isinstance(seq, nn.Sequential)
    if isinstance(x, (list, tuple)):     # channel concatenation of several maps (conv_fusion's input, fusion_encoder.py:1163)
        if usable(seq):                  # written token-major in one pass per source: to_rows() below is then a view
            x = torch.cat([t.permute(0, 2, 3, 1).float() for t in x], 3).permute(0, 3, 1, 2)
        else:
            x = torch.cat(list(x), 1)
    if not usable(seq):
        if transpose:
            y = seq(x.permute(0, 1, 3, 2).contiguous())
            return y.permute(0, 1, 3, 2)
        return seq(x)
    layers = _layers(nn.Sequential(seq) if single else seq)
    B, _, H, W = x.shape
    half = bool(sp.AUTOCAST_HALF and torch.is_autocast_enabled())
    with torch.autocast("cuda", enabled=False):
        rows = to_rows(x)
        for conv, bn, relu in layers:
            rb = dense_rulebook(rows.device, B, H, W, conv.stride[0])
            rows = DenseConvFunction.apply(rows, conv.weight, rb, transpose, half)
            H, W = rb.out_shape
            if conv.bias is not None:
                rows = rows + conv.bias.float()
            if bn is not None:
                rows = bn1d_relu(bn, rows, relu=relu)
            elif relu:
                rows = torch.relu(rows)
        return from_rows(rows, B, H, W)
