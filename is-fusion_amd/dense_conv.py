"""Dense 3x3 BEV convolutions (Conv2d + BatchNorm2d(eval) + ReLU) on the f16x3 sparse-conv kernel.

The reference runs them through mmcv ConvModule / nn.Conv2d (MIOpen): fusion_encoder.py:862-960 (conv_fusion, conv_ins,
conv_scene, conv_heatmap, heatmap_head_1/2), backbones/second.py:126-165 (SECONDV2).  Here a dense [B, H, W] grid is a
sparse tensor with every cell active; activations travel between layers as split-format token matrices
(``SplitMap``), so a stack of convs never leaves that layout.  fp32-class accuracy (f16 hi/lo split, fp32 accumulate).

No CPU fallback; inference only.
"""
import ctypes

import torch

from . import _lib
from .norm import fold_bn
from .spconv import pack_filters_f16x3

SUPPORTED = (32, 64, 128, 256)


class SplitMap:
    """[B, C, H, W] feature map held as a split-format token matrix (token = (b*H + y)*W + x)."""

    def __init__(self, data, B, C, H, W):
        self.data, self.B, self.C, self.H, self.W = data, B, C, H, W

    @property
    def num_tokens(self):
        return self.B * self.H * self.W

    @staticmethod
    def from_nchw(x, channel_offset=0, channels=None):
        """a (<= 256-channel) slice of an fp32 map -> SplitMap"""
        _lib.require_cuda(x)
        x = x.contiguous().float()
        B, Ct, H, W = x.shape
        C = Ct - channel_offset if channels is None else channels
        out = torch.empty(B * H * W * C * 4, dtype=torch.uint8, device=x.device)
        _lib.check(_lib.load().isf_nchw_to_split(_lib.ptr(x), B, Ct, channel_offset, C, H * W, _lib.ptr(out), C, 0,
                                                 _lib.stream()), "isf_nchw_to_split")
        return SplitMap(out, B, C, H, W)

    def to_rows(self):
        """[B*H*W, C] fp32 token rows (token = (b*H + y)*W + x): what the row-major fused Linear reads"""
        from .spconv import from_split
        return from_split(self.data, (self.num_tokens, self.C))

    def to_nchw(self, channels=None):
        """[B, C, H, W] fp32; channels: keep only the first `channels` (a view of the converted map)"""
        out = torch.empty((self.B, self.C, self.H, self.W), dtype=torch.float32, device=self.data.device)
        _lib.check(_lib.load().isf_split_to_nchw(_lib.ptr(self.data), self.B, self.C, self.H * self.W, _lib.ptr(out),
                                                 _lib.stream()), "isf_split_to_nchw")
        return out if channels is None else out[:, :channels]


_grids = {}


def grid_rulebook(device, B, H, W, stride=1, transpose=False):
    """cached arithmetic neighbour table of a dense B x H x W grid for a 3x3 / pad 1 convolution"""
    key = (str(device), B, H, W, stride, bool(transpose))
    if key not in _grids:
        lib = _lib.load()
        ohw = (ctypes.c_int * 2)()
        _lib.check(lib.isf_dense_grid_rulebook(B, H, W, 3, 3, stride, 1, int(transpose), None, 0, ohw, None))
        n_out = B * ohw[0] * ohw[1]
        nstride = lib.isf_nbr_stride(n_out)
        nbr = torch.empty((9, nstride), dtype=torch.int32, device=device)
        _lib.check(lib.isf_dense_grid_rulebook(B, H, W, 3, 3, stride, 1, int(transpose), _lib.ptr(nbr), nstride, ohw,
                                               _lib.stream()), "isf_dense_grid_rulebook")
        _grids[key] = (nbr, nstride, ohw[0], ohw[1])
    return _grids[key]


class PackedConvBN:
    """Conv2d(3x3, pad 1, no bias) [+ BatchNorm2d(eval)] [+ ReLU] packed for the f16x3 kernel.  Input channels beyond 256
    are split into <= 256-channel groups whose partial sums chain through the kernel's residual input."""

    def __init__(self, conv, bn=None, relu=True):
        w = conv.weight.detach().float()
        _lib.require_cuda(w)
        assert tuple(w.shape[2:]) == (3, 3) and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
        self.c_out_real, self.c_in = w.shape[:2]
        self.stride = conv.stride[0]
        self.relu = relu
        dev = w.device
        if bn is not None:
            self.scale, self.shift = fold_bn(bn)
        else:
            self.scale = torch.ones(self.c_out_real, dtype=torch.float32, device=dev)
            self.shift = torch.zeros(self.c_out_real, dtype=torch.float32, device=dev)
        if conv.bias is not None:
            self.shift = (self.shift + conv.bias.detach().float() * self.scale).contiguous()
        self.c_out = self.c_out_real
        if self.c_out_real < SUPPORTED[0]:
            # fewer output channels than the narrowest MFMA column tile (the 10-class heat-map convs): zero columns up
            # to 32; the caller keeps channels [:c_out_real] of the result (SplitMap.to_nchw(channels=...))
            self.c_out = SUPPORTED[0]
            pad = self.c_out - self.c_out_real
            w = torch.cat([w, w.new_zeros((pad,) + tuple(w.shape[1:]))], 0)
            self.scale = torch.cat([self.scale, self.scale.new_ones(pad)]).contiguous()
            self.shift = torch.cat([self.shift, self.shift.new_zeros(pad)]).contiguous()
        if self.c_out not in SUPPORTED:
            raise _lib.IsfError(f"dense conv: {self.c_out} output channels not built {SUPPORTED}")
        self.zero_shift = torch.zeros(self.c_out, dtype=torch.float32, device=dev)
        self.groups = []   # (channel offset, channels, packed weights, packed weights for the transposed map)
        off = 0
        while off < self.c_in:
            c = min(256, self.c_in - off)
            if c not in SUPPORTED:
                raise _lib.IsfError(f"dense conv: input channel group of {c} not built {SUPPORTED}")
            wk = w[:, off:off + c].permute(2, 3, 1, 0).contiguous()            # [ky, kx, Cin, Cout]
            wt = w[:, off:off + c].permute(3, 2, 1, 0).contiguous()            # taps enumerated (kx, ky)
            self.groups.append((off, c, pack_filters_f16x3(wk.view(1, 3, 3, c, self.c_out)),
                                pack_filters_f16x3(wt.view(1, 3, 3, c, self.c_out))))
            off += c

    def __call__(self, inputs, transpose=False):
        """inputs: SplitMap (c_in <= 256) or list of SplitMaps covering the channel groups in order.
        transpose=True: the result equals conv(x.permute(0,1,3,2)).permute(0,1,3,2) -- the convolution the reference
        applies to the spatially transposed map, expressed on the un-transposed tokens."""
        maps = inputs if isinstance(inputs, (list, tuple)) else [inputs]
        assert len(maps) == len(self.groups), "one SplitMap per 256-channel group"
        m0 = maps[0]
        nbr, nstride, oh, ow = grid_rulebook(m0.data.device, m0.B, m0.H, m0.W, self.stride, False)
        n_out = m0.B * oh * ow
        lib = _lib.load()
        acc = None
        for gi, (m, (off, c, pk, pkt)) in enumerate(zip(maps, self.groups)):
            assert m.C == c and (m.B, m.H, m.W) == (m0.B, m0.H, m0.W)
            last = gi == len(self.groups) - 1
            out = torch.empty(n_out * self.c_out * 4, dtype=torch.uint8, device=m.data.device)
            _lib.check(lib.isf_sparse_conv_forward_f16x3(
                _lib.ptr(m.data), m.num_tokens, c, _lib.ptr(pkt if transpose else pk), 9, self.c_out, _lib.ptr(nbr),
                nstride, n_out, _lib.ptr(self.scale), _lib.ptr(self.shift if last else self.zero_shift),
                _lib.ptr(acc) if acc is not None else None, int(self.relu and last), _lib.ptr(out), 0, _lib.stream()),
                "isf_sparse_conv_forward_f16x3")
            acc = out
        return SplitMap(acc, m0.B, self.c_out, oh, ow)


def pack_sequential(seq):
    """nn.Sequential of (Conv2d, BatchNorm2d, ReLU)* -> list of PackedConvBN (None when a layer does not fit)."""
    mods = list(seq)
    out = []
    i = 0
    while i < len(mods):
        conv = mods[i]
        bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], torch.nn.BatchNorm2d) else None
        j = i + (2 if bn is not None else 1)
        relu = j < len(mods) and isinstance(mods[j], torch.nn.ReLU)
        if not isinstance(conv, torch.nn.Conv2d):
            return None
        out.append(PackedConvBN(conv, bn, relu))
        i = j + (1 if relu else 0)
    return out
