"""Parameter containers of the HSF / IGF modules with the REFERENCE's sub-module and parameter names, so that
``state_dict()`` keys equal those of mmdet3d/models/middle_encoders/fusion_encoder.py, models/sst/*,
models/backbones/{sst_v2,second}.py and a released IS-Fusion checkpoint loads unchanged.  The arithmetic lives in
``fusion_encoder.py`` (HIP kernels through the C ABI); the stock 3x3 convolutions stay on PyTorch-ROCm / MIOpen as
the north_star prescribes.
"""
import torch
from torch import nn


class ConvModule(nn.Sequential):
    """mmcv ConvModule(conv_cfg=Conv2d, norm_cfg=BN2d) as the path uses it: sub-modules ``conv`` (no bias), ``bn``,
    ``activate`` (fusion_encoder.py:862-869)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, padding=1, eps=1e-5, momentum=0.1):
        super().__init__()
        self.add_module("conv", nn.Conv2d(in_channels, out_channels, kernel_size, 1, padding, bias=False))
        self.add_module("bn", nn.BatchNorm2d(out_channels, eps=eps, momentum=momentum))
        self.add_module("activate", nn.ReLU(inplace=True))


class _SelfAttn(nn.Module):
    """holder named like nn.MultiheadAttention: in_proj_weight/in_proj_bias/out_proj.{weight,bias}"""

    def __init__(self, d):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d)
        nn.init.xavier_uniform_(self.in_proj_weight)


class WindowAttention(nn.Module):
    def __init__(self, d, nhead):
        super().__init__()
        self.nhead = nhead
        self.self_attn = _SelfAttn(d)


class EncoderLayer(nn.Module):
    """sst_basic_block_v2.py:77-126 (post-norm, LayerNorm, GELU)."""

    def __init__(self, d, nhead, dim_feedforward):
        super().__init__()
        self.win_attn = WindowAttention(d, nhead)
        self.linear1 = nn.Linear(d, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d)
        self.norm1 = nn.LayerNorm(d)
        self.norm2 = nn.LayerNorm(d)


class BasicShiftBlockV2(nn.Module):
    def __init__(self, d, nhead, dim_feedforward):
        super().__init__()
        self.encoder_list = nn.ModuleList([EncoderLayer(d, nhead, dim_feedforward) for _ in range(2)])


class SSTv2(nn.Module):
    """backbones/sst_v2.py:11-63 parameter layout (linear0 only when in_channel is given)."""

    def __init__(self, d_model, nhead, num_blocks, dim_feedforward, output_shape, in_channel=None):
        super().__init__()
        self.d_model, self.nhead, self.output_shape = d_model, nhead, output_shape
        if in_channel is not None:
            self.linear0 = nn.Linear(in_channel, d_model[0])
        self.block_list = nn.ModuleList([BasicShiftBlockV2(d_model[i], nhead[i], dim_feedforward[i])
                                         for i in range(num_blocks)])
        for name, p in self.named_parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)


class SSTInputLayerV2(nn.Module):
    """models/sst/sst_input_layer_v2.py: no parameters; on a dense grid window membership is arithmetic."""

    def __init__(self, window_shape, sparse_shape, drop_info=None, shuffle_voxels=True, pos_temperature=1000,
                 normalize_pos=False, pos_embed=None, **kwargs):
        super().__init__()
        self.window_shape, self.sparse_shape = window_shape, sparse_shape
        self.pos_temperature, self.pos_embed_channels = pos_temperature, pos_embed
        assert not normalize_pos


class PositionEmbeddingLearned(nn.Module):
    def __init__(self, input_channel, num_pos_feats):
        super().__init__()
        self.position_embedding_head = nn.Sequential(
            nn.Conv1d(input_channel, num_pos_feats, kernel_size=1), nn.BatchNorm1d(num_pos_feats),
            nn.ReLU(inplace=True), nn.Conv1d(num_pos_feats, num_pos_feats, kernel_size=1))


class MSDeformAttn(nn.Module):
    def __init__(self, d_model, n_levels, n_heads, n_points):
        super().__init__()
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)


class DeformableTransformerDecoderLayer(nn.Module):
    def __init__(self, d_model, d_ffn, n_levels, n_heads, n_points):
        super().__init__()
        self.cross_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.norm1 = nn.LayerNorm(d_model)
        self.self_attn = _SelfAttn(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.norm3 = nn.LayerNorm(d_model)


class InsContextAtt(nn.Module):
    def __init__(self, num_layers=1, embed_dims=128, bev_size=180, n_points=16):
        super().__init__()
        self.bev_size, self.num_layers, self.n_points = bev_size, num_layers, n_points
        self.layers = nn.ModuleList([DeformableTransformerDecoderLayer(embed_dims, embed_dims, 1, 8, n_points)
                                     for _ in range(num_layers)])
        self.query_pos_embed = PositionEmbeddingLearned(2, embed_dims)
        self.key_pos_embed = PositionEmbeddingLearned(2, embed_dims)


class Instane2SceneAtt(nn.Module):
    def __init__(self, d_model, nhead=8):
        super().__init__()
        self.nhead = nhead
        self.multihead_attn = _SelfAttn(d_model)
        self.norm = nn.LayerNorm(d_model)


class SECONDV2(nn.Module):
    """backbones/second.py:98-238: stock Conv2d(bias=False)+BN(eps 1e-3)+ReLU stacks, invoked stage-wise from the
    fusion encoder.  Convolutions stay on PyTorch-ROCm (MIOpen)."""

    def __init__(self, in_channels=128, out_channels=(128, 128, 256), layer_nums=(3, 5, 5), layer_strides=(2, 2, 2),
                 norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01), conv_cfg=dict(type="Conv2d", bias=False), **kw):
        super().__init__()
        eps, mom = norm_cfg.get("eps", 1e-3), norm_cfg.get("momentum", 0.01)
        in_filters = [in_channels, *out_channels[:-1]]
        blocks = []
        for i, n in enumerate(layer_nums):
            block = []
            first = [nn.Conv2d(in_filters[i], out_channels[i], 3, stride=layer_strides[i], padding=1, bias=False),
                     nn.BatchNorm2d(out_channels[i], eps=eps, momentum=mom), nn.ReLU(inplace=True)]
            if layer_strides[i] == 2:
                self.ds_layer = nn.Sequential(*first)
            else:
                block = first
            for _ in range(n):
                block += [nn.Conv2d(out_channels[i], out_channels[i], 3, padding=1, bias=False),
                          nn.BatchNorm2d(out_channels[i], eps=eps, momentum=mom), nn.ReLU(inplace=True)]
            blocks.append(nn.Sequential(*block))
        self.blocks = nn.ModuleList(blocks)

    dense_conv = "hip"   # "hip": f16x3 MFMA kernel of the sparse encoder on the dense grid; "stock": MIOpen

    def _packed(self, name, seq):
        from .dense_conv import pack_sequential
        from .fusion_ops import watch_parameters
        cache = self.__dict__.setdefault("_isf_packed", {})
        watch_parameters(self)
        dev = next(seq.parameters()).device
        if cache.get(name, (None,))[0] != dev:
            cache[name] = (dev, pack_sequential(seq))
        return cache[name][1]

    def _run(self, name, seq, x):
        """x: [B, C, H, W] fp32 or a dense_conv.SplitMap -> SplitMap (hip) / tensor (stock)"""
        from .dense_conv import SplitMap
        if self.dense_conv != "hip":
            return seq(x)
        m = x if isinstance(x, SplitMap) else SplitMap.from_nchw(x)
        for layer in self._packed(name, seq):
            m = layer(m)
        return m

    def forward(self, x, stage=None):
        """(tokens, coords, feature) like the reference; tokens/coords of the dense grid are implicit here, so the
        HIP fusion encoder consumes the [B, C, H, W] tensor directly: stage1 -> (ds_layer output, None, feature)."""
        from .dense_conv import SplitMap

        def nchw(t):
            return t.to_nchw() if isinstance(t, SplitMap) else t
        if self.training:
            raise RuntimeError("isfusion_amd.SECONDV2 is the inference path (eval mode)")
        if stage == "stage1":
            feat = self._run("b0", self.blocks[0], x[0] if isinstance(x, (list, tuple)) else x)
            return nchw(self._run("ds", self.ds_layer, feat)), None, nchw(feat)
        if stage == "stage2":
            return None, None, nchw(self._run("b1", self.blocks[1], x[0] if isinstance(x, (list, tuple)) else x))
        x1 = self._run("b0", self.blocks[0], x)
        return nchw(x1), nchw(self._run("b1", self.blocks[1], self._run("ds", self.ds_layer, x1)))


def seeded_state_dict(module, seed):
    """Deterministic, well-conditioned random values for every parameter / buffer of ``module`` (keyed by name, so
    it does not depend on construction order).  Used by the golden generators (loaded into the REFERENCE modules)
    and by the GPU tests (loaded into these modules).  Normalisation scales are drawn around 1 (found by module type:
    a BatchNorm inside an nn.Sequential has no telling name), so that signals neither die nor explode through the
    conv stacks and a parity check on the last feature map still exercises the whole data path."""
    import hashlib
    norm_weights = set()
    for name, m in module.named_modules():
        if isinstance(m, (nn.modules.batchnorm._BatchNorm, nn.LayerNorm, nn.GroupNorm)):
            norm_weights.add((name + "." if name else "") + "weight")
    sd = {}
    for k, v in module.state_dict().items():
        h = int(hashlib.sha256((str(seed) + "/" + k).encode()).hexdigest()[:8], 16)
        g = torch.Generator().manual_seed(h)
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros_like(v)
        elif k.endswith("running_var"):
            sd[k] = torch.rand(v.shape, generator=g) * 0.5 + 0.75
        elif k.endswith("running_mean"):
            sd[k] = torch.randn(v.shape, generator=g) * 0.1
        elif k in norm_weights:
            sd[k] = torch.rand(v.shape, generator=g) * 0.5 + 0.75
        elif v.dim() == 1:
            sd[k] = torch.randn(v.shape, generator=g) * 0.1
        else:
            fan_in = v[0].numel()
            sd[k] = torch.randn(v.shape, generator=g) * (1.0 / fan_in) ** 0.5
        if k.endswith("sampling_offsets.bias"):  # keep sampling points a few cells around the reference point
            sd[k] = torch.randn(v.shape, generator=g) * 2.0
    return sd


class SECONDFPN(nn.Module):
    """necks/second_fpn.py:10-93 as the IS-Fusion config builds it (use_conv_for_no_stride=True): per level a
    Conv2d(k = 1/stride) or ConvTranspose2d(k = stride) without bias + BN(eps 1e-3) + ReLU, concatenated, and the
    final `permute(0, 1, 3, 2)`.  Stock PyTorch-ROCm ops (a 1x1 conv and a 2x2 transposed conv per frame)."""

    def __init__(self, in_channels=(128, 256), out_channels=(256, 256), upsample_strides=(1, 2),
                 norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01), upsample_cfg=dict(type="deconv", bias=False),
                 conv_cfg=dict(type="Conv2d", bias=False), use_conv_for_no_stride=True, **kw):
        super().__init__()
        self.in_channels, self.out_channels = list(in_channels), list(out_channels)
        eps, mom = norm_cfg.get("eps", 1e-3), norm_cfg.get("momentum", 0.01)
        blocks = []
        for cin, cout, s in zip(in_channels, out_channels, upsample_strides):
            if s > 1 or (s == 1 and not use_conv_for_no_stride):
                up = nn.ConvTranspose2d(cin, cout, s, stride=s, bias=upsample_cfg.get("bias", False))
            else:
                k = int(round(1 / s))
                up = nn.Conv2d(cin, cout, k, stride=k, bias=conv_cfg.get("bias", False))
            blocks.append(nn.Sequential(up, nn.BatchNorm2d(cout, eps=eps, momentum=mom), nn.ReLU(inplace=True)))
        self.deblocks = nn.ModuleList(blocks)

    # "stock": the torch modules above (MIOpen).  "hip" (opt-in until it has run on hardware): both
    # levels are GEMMs over BEV tokens -- the 1x1 conv is a linear layer, the k = s transposed conv is ONE linear layer
    # with s*s*Cout outputs whose columns are dealt to the s x s sub-cells -- on the fused f16x3 linear kernel with
    # BatchNorm folded into weight / bias and ReLU in its epilogue.
    dense_conv = "stock"

    def _folded(self, i):
        """(weight [taps*Cout, Cin], bias [taps*Cout], stride) of level i with eval BatchNorm folded in"""
        up, bn = self.deblocks[i][0], self.deblocks[i][1]
        scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().float()
        shift = (bn.bias - bn.running_mean * scale).detach().float()
        w = up.weight.detach().float()
        if isinstance(up, nn.ConvTranspose2d):
            s = up.stride[0]
            assert up.kernel_size == (s, s) and up.stride == (s, s) and up.bias is None
            wt = w.permute(2, 3, 1, 0).reshape(s * s * w.shape[1], w.shape[0])     # row = (dy*s + dx)*Cout + co
            return wt * scale.repeat(s * s)[:, None], shift.repeat(s * s), s
        assert up.kernel_size == (1, 1) and up.stride == (1, 1) and up.bias is None
        return w[:, :, 0, 0] * scale[:, None], shift, 1

    def forward_tokens(self, x, linear_relu):
        """the "hip" data path with the GEMM injected: linear_relu(x [B, Cin, H, W], weight, bias) -> relu(tokens W^T +
        b) as [B*H*W, N] rows ((b, y, x) order).  (tests/test_host.py runs it with a torch GEMM against the modules.)"""
        ups = []
        for i in range(len(self.deblocks)):
            w, b, s = self._folded(i)
            B, _, H, W = x[i].shape
            y = linear_relu(x[i], w, b)                                   # [B*H*W, s*s*Cout]
            cout = y.shape[1] // (s * s)
            y = y.view(B, H, W, s, s, cout).permute(0, 5, 1, 3, 2, 4).reshape(B, cout, H * s, W * s)
            ups.append(y)
        out = torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]
        return [out.permute(0, 1, 3, 2).contiguous()]

    def forward(self, x, **kwargs):
        assert len(x) == len(self.in_channels)
        if self.dense_conv == "hip" and not self.training:
            from . import fusion_ops as ops
            cache = self.__dict__.setdefault("_isf_packed", {})
            ops.watch_parameters(self)

            def linear_relu(t, w, b):
                key = (w.shape[0], w.shape[1], t.device)
                if key not in cache:
                    cache[key] = ops.PackedLinear(w.to(t.device), b.to(t.device))
                return ops.linear(t.float(), cache[key], act=ops.ACT_RELU)

            with torch.no_grad():
                return self.forward_tokens(x, linear_relu)
        ups = [d(x[i]) for i, d in enumerate(self.deblocks)]
        out = torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]
        return [out.permute(0, 1, 3, 2).contiguous()]
