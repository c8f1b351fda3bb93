"""The HSF / IGF modules with the REFERENCE's sub-module and parameter names (``state_dict()`` keys equal those of
mmdet3d/models/middle_encoders/fusion_encoder.py, models/sst/*, models/backbones/{sst_v2,second}.py, so a released
IS-Fusion checkpoint loads unchanged) and the reference's ``forward()`` signatures.  The arithmetic lives in
``fusion_ops.py`` / ``fusion_encoder.py`` (HIP kernels through the C ABI).  Eval mode, GPU tensors only: a CPU tensor
or training mode raises.  SSTInputLayerV2 / SSTv2 implement the case the IS-Fusion path feeds them -- EVERY cell of the
BEV grid is a token (fusion_encoder.py:1167-1181), where window membership is arithmetic -- and raise on a sparse token
set instead of silently doing something else.
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

    def forward(self, voxel_info, **kwargs):
        """sst_v2.py:65-91: voxel_info from SSTInputLayerV2 -> [ [B, d, ny, nx] ] (the recovered BEV map)."""
        from . import fusion_ops as ops
        if self.training:
            raise RuntimeError("isfusion_amd.SSTv2 is the inference path (eval mode)")
        grid = voxel_info.get("dense_grid")
        if grid is None:
            raise _not_dense("SSTv2.forward")
        B, ny, nx = grid
        feats = voxel_info["voxel_feats"]
        ny_o, nx_o = self.output_shape
        assert (ny, nx) == (ny_o, nx_o) and ny == nx, "output_shape must be the (square) token grid"
        bev = feats.view(B, ny, nx, feats.size(1)).permute(0, 3, 1, 2)     # read channels-first inside the GEMMs
        return [ops.sstv2_forward(self, bev, voxel_info["window_shape"][0], voxel_info["pos_temperature"])]


def _not_dense(where):
    from . import _lib
    return _lib.IsfError(f"{where}: this build implements the dense-grid case of the IS-Fusion path (every BEV cell "
                         "is a token, in (b, y, x) order: fusion_encoder.py:1167-1173); a sparse token set needs the "
                         "reference's window batching and is not built")


class SSTInputLayerV2(nn.Module):
    """models/sst/sst_input_layer_v2.py:20-110.  No parameters.  On the dense grid of the IS-Fusion path nothing is
    dropped (max_tokens = the full window) and window membership / in-window position are arithmetic on (y, x), so the
    voxel_info handed to SSTv2 carries the grid instead of index tensors."""

    def __init__(self, window_shape, sparse_shape, drop_info=None, shuffle_voxels=True, pos_temperature=1000,
                 normalize_pos=False, pos_embed=None, **kwargs):
        super().__init__()
        self.window_shape, self.sparse_shape = window_shape, sparse_shape
        self.pos_temperature, self.pos_embed_channels = pos_temperature, pos_embed
        self.meta_drop_info = drop_info
        assert not normalize_pos

    def forward(self, voxel_feats, voxel_coors, batch_size=None):
        """voxel_feats [N, C]; voxel_coors [N, 4] (b, z, y, x) -> voxel_info dict (sst_input_layer_v2.py:63-110).
        N must be B * ny * nx with the rows in (b, y, x) order (checked on the device)."""
        from . import _lib
        _lib.require_cuda(voxel_feats, voxel_coors)
        nx, ny = int(self.sparse_shape[0]), int(self.sparse_shape[1])
        N = voxel_feats.size(0)
        B = int(batch_size) if batch_size is not None else N // max(nx * ny, 1)
        if B <= 0 or N != B * ny * nx:
            raise _not_dense("SSTInputLayerV2.forward")
        c = voxel_coors.long()
        r = torch.arange(N, device=c.device)
        ok = (c[:, 0] == r // (ny * nx)) & (c[:, 2] == (r // nx) % ny) & (c[:, 3] == r % nx)
        if not bool(ok.all()):
            raise _not_dense("SSTInputLayerV2.forward")
        win = self.window_shape
        for lvl in (self.meta_drop_info or {}).values() if isinstance(self.meta_drop_info, dict) else ():
            info = lvl.get(0, lvl) if isinstance(lvl, dict) else lvl
            if isinstance(info, dict) and info.get("max_tokens", win[0] * win[1]) < win[0] * win[1]:
                raise _lib.IsfError("SSTInputLayerV2: drop_info would drop tokens of a full window (max_tokens < "
                                    "window size): region batching with drops is not built")
        return dict(voxel_feats=voxel_feats.float().contiguous(), voxel_coors=c, dense_grid=(B, ny, nx),
                    window_shape=tuple(win), pos_temperature=float(self.pos_temperature))


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

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None):
        """fusion_encoder.py:560-600 (reference_points [..., 2]): -> (output [N, Lq, C], sampling_locations,
        attention_weights).  Projections on the fused linear kernel, sampling through isf_ms_deform_attn_forward (the
        mmcv op signature, any number of levels)."""
        from . import fusion_ops as ops
        if self.training:
            raise RuntimeError("isfusion_amd.MSDeformAttn is the inference path (eval mode)")
        N, Lq, C = query.shape
        Lin = input_flatten.shape[1]
        M, L, P = self.n_heads, self.n_levels, self.n_points
        assert reference_points.shape[-1] == 2, "reference boxes (last dim 4) are not on the IS-Fusion path"
        c = ops._cache(self, query.device)
        if "value" not in c:
            for name, lin in (("value", self.value_proj), ("off", self.sampling_offsets), ("aw", self.attention_weights),
                              ("out", self.output_proj)):
                c[name] = ops.PackedLinear(lin.weight, lin.bias)
        q2 = query.reshape(N * Lq, C).float().contiguous()
        value = ops.linear(input_flatten.reshape(N * Lin, C).float().contiguous(), c["value"])
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask.reshape(N * Lin, 1), 0.0)
        off = ops.linear(q2, c["off"]).view(N, Lq, M, L, P, 2)
        aw = torch.softmax(ops.linear(q2, c["aw"]).view(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
        shapes = input_spatial_shapes.to(query.device).long()
        normalizer = torch.stack([shapes[..., 1], shapes[..., 0]], -1).float()
        loc = reference_points[:, :, None, :, None, :].float() + off / normalizer[None, None, None, :, None, :]
        out = ops.ms_deform_attn(value.view(N, Lin, M, C // M), shapes, input_level_start_index.to(query.device).long(),
                                 loc, aw)
        out = ops.linear(out.view(N * Lq, C), c["out"]).view(N, Lq, C)
        return out, loc, aw


class DeformableTransformerDecoderLayer(nn.Module):
    def __init__(self, d_model, d_ffn, n_levels, n_heads, n_points, dropout=0.1):
        super().__init__()
        # dropout1..4 of the reference layer (fusion_encoder.py:604-668, p = 0.1 from :779): parameter-free, so the
        # state-dict keys are unchanged; applied by the training path (fusion_train.ins_context_att)
        self.dropout = dropout
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

    def forward(self, query_feats, query_pos, bev_pos=None, scene_feats=None, **kwargs):
        """fusion_encoder.py:795-830: query_feats [B, E, Q], query_pos [B, Q, 2] (x, y in cells), bev_pos = the
        create_2D_grid cell centres [1, H*W, 2] (recomputed here; checked when given), scene_feats [B, E, H, W] in the
        orientation the REFERENCE passes (it is transposed inside, :797) -> [B, E, Q]."""
        from . import fusion_ops as ops
        if self.training:
            raise RuntimeError("isfusion_amd.InsContextAtt is the inference path (eval mode)")
        if bev_pos is not None:
            assert bev_pos.shape[-2] == self.bev_size * self.bev_size, "bev_pos must be the bev_size x bev_size grid"
        scene_t = scene_feats.permute(0, 1, 3, 2).contiguous()
        return ops.ins_context_att(self, query_feats, query_pos, scene_t, self.bev_size)


class Instane2SceneAtt(nn.Module):
    def __init__(self, d_model, nhead=8, dropout=0.1):
        super().__init__()
        self.nhead = nhead
        self.dropout = dropout   # nn.Dropout on the residual branch (fusion_encoder.py:478, :492); training path only
        self.multihead_attn = _SelfAttn(d_model)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, query, key, query_scene, bs, bev_size, attn_mask=None):
        """fusion_encoder.py:480-502: query [B, E, H*W] (conv_ins output, flattened), key [B, E, Q] (instances),
        query_scene [B, E, H, W] -> [B, E, H, W]."""
        from . import fusion_ops as ops
        if self.training:
            raise RuntimeError("isfusion_amd.Instane2SceneAtt is the inference path (eval mode)")
        assert attn_mask is None, "attn_mask is not used on the IS-Fusion path"
        E = query.shape[1]
        return ops.instance_to_scene(self, query.reshape(bs, E, bev_size, bev_size), key, query_scene, bev_size)


class SECONDV2(nn.Module):
    """backbones/second.py:98-238: stock Conv2d(bias=False)+BN(eps 1e-3)+ReLU stacks, invoked stage-wise from the
    fusion encoder.  dense_conv = "hip": on the sparse-conv kernels over the dense grid (inference: dense_conv.py, training:
    dense_train.py); "stock": PyTorch-ROCm (MIOpen)."""

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
        from .fusion_ops import frozen, param_key
        cache = self.__dict__.setdefault("_isf_packed", {})
        dev = next(seq.parameters()).device
        key = (dev, None if name in cache and frozen(self) else param_key(seq))
        if name not in cache or cache[name][0][0] != dev or (key[1] is not None and cache[name][0][1] != key[1]):
            cache[name] = ((dev, key[1] if key[1] is not None else param_key(seq)), pack_sequential(seq))
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

    def forward(self, x, stage=None, keep_split=False):
        """(tokens, coords, feature) like the reference; tokens/coords of the dense grid are implicit here, so the
        HIP fusion encoder consumes the [B, C, H, W] tensor directly: stage1 -> (ds_layer output, None, feature).
        keep_split (engine-level hand-over, eval): the results stay dense_conv.SplitMap -- their readers (the SST's and the
        neck's fused Linears) take token rows, so no [B, C, H, W] map is made of them."""
        from .dense_conv import SplitMap

        def nchw(t):
            return t.to_nchw() if isinstance(t, SplitMap) and not keep_split else t
        if self.training:      # training: conv + BatchNorm (batch statistics) + ReLU with autograd -- on the sparse-conv
            from . import dense_train as dt      # kernels over the dense grid (dense_train.py), or the stock modules
            run = dt.conv_stack if self.dense_conv == "hip" else (lambda seq, t: seq(t))
            if stage == "stage1":
                feat = run(self.blocks[0], x[0] if isinstance(x, (list, tuple)) else x)
                return run(self.ds_layer, feat), None, feat
            if stage == "stage2":
                return None, None, run(self.blocks[1], x[0] if isinstance(x, (list, tuple)) else x)
            x1 = run(self.blocks[0], x)
            return x1, run(self.blocks[1], run(self.ds_layer, x1))
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
    final `permute(0, 1, 3, 2)`.  Training: stock PyTorch-ROCm ops (a 1x1 conv and a 2x2 transposed conv per frame)."""

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

    # "hip" (default in eval mode): both levels are GEMMs over BEV tokens -- the 1x1 conv is a linear
    # layer, the k = s transposed conv is ONE linear layer with s*s*Cout outputs whose columns are dealt to the s x s
    # sub-cells -- on the fused f16x3 linear kernel with BatchNorm folded into weight / bias and ReLU in its epilogue.
    # "stock": the torch modules above (MIOpen; 0.49 ms per forward at B = 2 against 0.2 ms, tools/glue_profile.py).
    dense_conv = "hip"

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

    def forward_tokens(self, x, linear_relu, cached=False):
        """the "hip" data path with the GEMM injected: linear_relu(x [B, Cin, H, W], weight, bias) -> relu(tokens W^T +
        b) as [B*H*W, N] rows ((b, y, x) order).  (tests/test_host.py runs it with a torch GEMM against the modules.)"""
        out = None
        c0 = 0
        for i in range(len(self.deblocks)):
            w, b, s = self._folded_cached(i) if cached else self._folded(i)
            B, _, H, W = x[i].shape
            y = linear_relu(x[i], w, b)                                   # [B*H*W, s*s*Cout], column = (dy*s + dx)*Cout + co
            cout = y.shape[1] // (s * s)
            if out is None:
                out = y.new_empty((B, sum(self.out_channels), W * s, H * s))
            # the concatenated map AND the reference's final permute(0, 1, 3, 2) in one copy per level:
            # out[b, c0 + co, w*s + dx, h*s + dy] = y[b, h, w, dy, dx, co]
            out[:, c0:c0 + cout].view(B, cout, W, s, H, s).copy_(y.view(B, H, W, s, s, cout).permute(0, 5, 2, 4, 1, 3))
            c0 += cout
        return [out]

    def _linear_relu(self):
        """the injected GEMM of forward_tokens on the fused linear kernel, weights packed once per parameter version"""
        from . import fusion_ops as ops
        cache = self.__dict__.setdefault("_isf_packed", {})
        pk = None if cache and ops.frozen(self) else ops.param_key(self)
        if pk is not None and cache.get("_key") != pk:
            cache.clear()
            cache["_key"] = pk

        def linear_relu(t, w, b):
            key = (w.shape[0], w.shape[1], t.device)
            if key not in cache:
                cache[key] = ops.PackedLinear(w.to(t.device), b.to(t.device))
            return ops.linear(t.float(), cache[key], act=ops.ACT_RELU)
        return linear_relu

    def _folded_cached(self, i):
        """_folded(i) once per parameter version (the cache _linear_relu() has just validated)"""
        cache = self.__dict__.setdefault("_isf_packed", {})
        if ("folded", i) not in cache:
            cache[("folded", i)] = self._folded(i)
        return cache[("folded", i)]

    @torch.no_grad()
    def forward_split(self, x):
        """engine-level hand-over to the detection head (ISFusionPtsPath.forward_pts): the levels as split-format token
        matrices (dense_conv.SplitMap, token = (b*H + y)*W + x of the UN-permuted map, one map per level = per
        <= 256-channel group of the concatenation).  No [B, 512, H, W] tensor, no permute copy, no NCHW -> split pass:
        the head applies its 3x3 convolutions with transposed taps instead (TransFusionHeadV2.forward_split)."""
        from .dense_conv import SplitMap
        from .spconv import to_split
        assert len(x) == len(self.in_channels) and not self.training
        linear_relu = self._linear_relu()
        maps = []
        for i in range(len(self.deblocks)):
            w, b, s = self._folded_cached(i)
            if isinstance(x[i], SplitMap):                                # token rows: the row-major GEMM (column-split
                B, H, W = x[i].B, x[i].H, x[i].W                          # launches for the 90 x 90 level), no NCHW map
                y = linear_relu(x[i].to_rows(), w, b)
            else:
                B, _, H, W = x[i].shape
                y = linear_relu(x[i], w, b)                               # [B*H*W, s*s*Cout], column = (dy*s + dx)*Cout + co
            cout = y.shape[1] // (s * s)
            if s > 1:                                                     # sub-cells to their tokens: 1-KiB runs
                y = y.view(B, H, W, s, s, cout).permute(0, 1, 3, 2, 4, 5).reshape(B * H * s * W * s, cout)
            maps.append(SplitMap(to_split(y), B, cout, H * s, W * s))
        assert all((m.H, m.W) == (maps[0].H, maps[0].W) for m in maps)
        return maps

    def forward(self, x, **kwargs):
        assert len(x) == len(self.in_channels)
        if self.dense_conv == "hip" and not self.training:
            with torch.no_grad():
                return self.forward_tokens(x, self._linear_relu(), cached=True)
        ups = [d(x[i]) for i, d in enumerate(self.deblocks)]
        out = torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]
        return [out.permute(0, 1, 3, 2).contiguous()]
