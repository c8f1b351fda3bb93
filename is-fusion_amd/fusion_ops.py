"""Host side of the HSF / IGF rows (A8, A10-A14): thin wrappers that hand device pointers to
libisf_hip.so.  No CPU fallback: every function raises when its tensors are not on a GPU or the library is missing.

Layout conventions: a dense BEV grid [B, C, S, S] becomes a token matrix [B*S*S, C] (row = (b*S + y)*S + x, the
order fusion_encoder.py:1167-1173 builds); the transposes between the two are stock torch ops.
"""

import ctypes

import torch

from . import _lib

# the HIP kernels compute in fp32: under torch.autocast (the reference trains with mixed precision) inputs are cast
# to fp32 on the way in and autocast is off inside forward / backward
_amp_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_amp_bwd = torch.amp.custom_bwd(device_type="cuda")

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2


# ---------------------------------------------------------------------------------------------------- linear
class PackedLinear:
    """nn.Linear weights split once into f16 hi/lo MFMA fragments (isf_pack_linear)."""

    def __init__(self, weight, bias=None, transposed=False):
        """transposed: pack weight^T ([in, out] given, e.g. the forward weight for dX = dY W) without a transposed copy"""
        _lib.require_cuda(weight)
        w = weight.detach().float().contiguous()
        self.out_features, self.in_features = (w.shape[1], w.shape[0]) if transposed else w.shape
        lib = _lib.load()
        self.packed = torch.empty(lib.isf_packed_linear_bytes(self.out_features, self.in_features), dtype=torch.uint8,
                                  device=w.device)
        pack = lib.isf_pack_linear_transposed if transposed else lib.isf_pack_linear
        _lib.check(pack(_lib.ptr(w), self.out_features, self.in_features, _lib.ptr(self.packed), _lib.stream()),
                   "isf_pack_linear")
        self.bias = None if bias is None else bias.detach().float().contiguous()


def linear(x, pl, *, table=None, index=None, act=ACT_NONE, residual=None, ln=None, out_nchw=None):
    """y = LN(act(x W^T + b + table[index]) + residual).
    x: [M, K] fp32 row-major, or a BEV map [B, K, H, W] read as its B*H*W tokens (no transpose pass);
    residual: [M, N] or [B, N, H, W] likewise; out_nchw=(B, H, W) writes y as [B, N, H, W] instead of [M, N]."""
    _lib.require_cuda(x)
    assert x.dtype == torch.float32
    x = x if x.is_contiguous() else x.contiguous()
    x_hw = res_hw = y_hw = 0
    if x.dim() == 4:
        assert x.size(1) == pl.in_features
        M, x_hw, ldx = x.size(0) * x.size(2) * x.size(3), x.size(2) * x.size(3), 0
    else:
        assert x.dim() == 2 and x.size(1) == pl.in_features
        M, ldx = x.size(0), x.stride(0)
    if out_nchw is not None:
        B, H, W = out_nchw
        assert B * H * W == M
        y = torch.empty((B, pl.out_features, H, W), dtype=torch.float32, device=x.device)
        y_hw, ldy = H * W, 0
    else:
        y = torch.empty((M, pl.out_features), dtype=torch.float32, device=x.device)
        ldy = y.stride(0)
    if residual is not None:
        residual = residual.contiguous()
        if residual.dim() == 4:
            assert residual.size(1) == pl.out_features and residual.numel() == M * pl.out_features
            res_hw = residual.size(2) * residual.size(3)
        else:
            assert tuple(residual.shape) == (M, pl.out_features)
    g = b = None
    eps = 0.0
    if ln is not None:
        g, b, eps = ln.weight.detach(), ln.bias.detach(), float(ln.eps)
    _lib.check(_lib.load().isf_linear_forward(
        _lib.ptr(x), M, pl.in_features, ldx, _lib.ptr(pl.packed), pl.out_features,
        _lib.ptr(pl.bias) if pl.bias is not None else None,
        _lib.ptr(table) if table is not None else None, _lib.ptr(index) if index is not None else None, act,
        _lib.ptr(residual) if residual is not None else None, _lib.ptr(g) if g is not None else None,
        _lib.ptr(b) if b is not None else None, eps, _lib.ptr(y), ldy, x_hw, res_hw, y_hw, _lib.stream()),
        "isf_linear_forward")
    return y


def drop_caches(module, *_):
    """Forget everything below `module` that was derived from its parameters: packed-weight / table caches, the
    SparseEncoder's C plan, the LidarBranch's folded VFE parameters and captured HIP graphs (their kernels hold
    pointers into the packed copies)."""
    for sub in module.modules():
        d = sub.__dict__
        d.pop("_isf_cache", None)
        d.pop("_isf_packed", None)
        if d.get("_plan") is not None:
            d["_plan"] = None
        if d.get("_vfe_cache") is not None:
            d["_vfe_cache"] = None
        if isinstance(d.get("_graphs"), dict):
            d["_graphs"].clear()


def param_key(module):
    """(version, address) of every parameter and buffer below `module`: changes whenever one of them is replaced or
    written in place -- by load_state_dict (mmcv's load_checkpoint recurses over _load_from_state_dict and never fires
    the post hooks), an optimizer step or a manual copy_()."""
    return tuple((t._version, t.data_ptr()) for t in list(module.parameters()) + list(module.buffers()))


def _unfreeze_on_load(module, *_):
    """load_state_dict pre-hook (fires inside every module's _load_from_state_dict, so also under mmcv's
    load_checkpoint): new weights are coming, the packed copies must be re-derived."""
    freeze(module.__dict__.get("_isf_freeze_root", module), False)


def freeze(module, flag=True):
    """Inference deployments: skip the per-call "did a parameter change?" scan of the caches below `module`.
    The skip ends by itself when weights can change: a load_state_dict anywhere below `module`, or a forward in
    training mode (see frozen()), clears it -- call freeze() again once the weights are final.

    Every change of the flag -- freezing, unfreezing, the load_state_dict hook -- DROPS the derived state below
    `module` (drop_caches): a frozen cache is used without looking at the parameters, so it must have been packed after
    the freeze; load_state_dict -> freeze() -> forward, or train() -> optimizer steps -> eval() -> freeze(), would
    otherwise reuse copies packed from the old weights (and replay HIP graphs that point into them)."""
    drop_caches(module)
    for sub in module.modules():
        sub.__dict__["_isf_frozen"] = bool(flag)
        if hasattr(sub, "_frozen"):          # SparseEncoder / LidarBranch keep their own flag
            sub._frozen = bool(flag)
        if flag and "_isf_freeze_root" not in sub.__dict__:
            sub.__dict__["_isf_freeze_root"] = module
            sub._register_load_state_dict_pre_hook(_unfreeze_on_load, with_module=True)
        elif flag:
            sub.__dict__["_isf_freeze_root"] = module
    return module


def frozen(module):
    """True when `module`'s caches may be used without checking the parameters.  A module seen in training mode loses
    the flag (an optimizer step is about to change its weights; a later eval() then re-validates by key)."""
    if not module.__dict__.get("_isf_frozen", False):
        return False
    if module.training:
        module.__dict__["_isf_frozen"] = False
        if hasattr(module, "_frozen"):
            module._frozen = False
        return False
    return True


def _cache(module, device):
    """Per-module cache of packed weights / tables derived from its parameters, dropped when the device or any
    parameter / buffer below the module changed (SparseEncoder._c_plan keys its plan the same way)."""
    c = module.__dict__.get("_isf_cache")
    key = None if (c is not None and frozen(module)) else param_key(module)
    if c is None or c.get("device") != device or (key is not None and c.get("_key") != key):
        c = {"device": device, "_key": key if key is not None else param_key(module)}
        module.__dict__["_isf_cache"] = c
    return c


def watch_parameters(module):
    """kept for callers that cache outside _cache(): nothing to register, validity is checked through param_key"""
    return module


def to_tokens(x):
    """[B, C, H, W] -> [B*H*W, C]"""
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


def from_tokens(t, B, H, W):
    return t.view(B, H, W, -1).permute(0, 3, 1, 2).contiguous()


# ------------------------------------------------------------------------------------------------ A10 / A11
def _window_tables(S, win, shift, d, temperature, device):
    """In-window token index per grid cell and the sinusoidal window position embedding
    (sst_ops.py:219-268 shifts, sst_input_layer_v2.py:224-290 embedding; 2-D, normalize_pos False)."""
    off = win // 2 if shift else win
    c = (torch.arange(S) + off) % win
    iy, ix = torch.meshgrid(c, c, indexing="ij")
    index = (iy * win + ix).reshape(-1).to(torch.int32)                     # [S*S]
    k = torch.arange(win * win)
    y = (k // win).double() - win / 2
    x = (k % win).double() - win / 2
    half = d // 2
    inv = torch.arange(half, dtype=torch.float32)
    inv = (torch.tensor(float(temperature)) ** (2 * (inv // 2) / half)).double()
    ex, ey = x[:, None].float() / inv.float(), y[:, None].float() / inv.float()
    ex = torch.stack([ex[:, ::2].sin(), ex[:, 1::2].cos()], dim=-1).flatten(-2)
    ey = torch.stack([ey[:, ::2].sin(), ey[:, 1::2].cos()], dim=-1).flatten(-2)
    pos = torch.cat([ex, ey], dim=-1)                                        # [win*win, d]
    return index.to(device), pos.to(device)


def _encoder_layer_cache(layer, S, win, shift, temperature, device, B):
    c = _cache(layer, device)
    key = ("win", S, win, shift, B)
    if key not in c:
        attn = layer.win_attn.self_attn
        d = attn.out_proj.in_features
        index, pos = _window_tables(S, win, shift, d, temperature, device)
        w, b = attn.in_proj_weight.detach().float(), attn.in_proj_bias.detach().float()
        # (x + pos) Wq = x Wq + pos Wq: the position term becomes a 36-row table added in the GEMM epilogue
        tab = torch.zeros((win * win, 3 * d), dtype=torch.float32, device=device)
        tab[:, :2 * d] = (pos.double() @ w[:2 * d].double().t()).float()
        c[key] = dict(index=index.repeat(B).contiguous(), table=tab.contiguous(), qkv=PackedLinear(w, b),
                      out=PackedLinear(attn.out_proj.weight, attn.out_proj.bias),
                      block=PackedWindowBlock(w, attn.out_proj.weight, layer.win_attn.nhead) if d == 128 else None,
                      in_bias=b.contiguous(), out_bias=attn.out_proj.bias.detach().float().contiguous(),
                      l1=PackedLinear(layer.linear1.weight, layer.linear1.bias),
                      l2=PackedLinear(layer.linear2.weight, layer.linear2.bias))
    return c[key]


class PackedWindowBlock:
    """in_proj / out_proj weights of an SST window-attention layer in the streaming order of isf_window_block_forward"""

    def __init__(self, in_proj_weight, out_proj_weight, nhead):
        _lib.require_cuda(in_proj_weight)
        w_in = in_proj_weight.detach().float().contiguous()
        w_out = out_proj_weight.detach().float().contiguous()
        d = w_out.size(0)
        lib = _lib.load()
        self.packed = torch.empty(lib.isf_packed_window_block_bytes(d), dtype=torch.uint8, device=w_in.device)
        _lib.check(lib.isf_pack_window_block(_lib.ptr(w_in), _lib.ptr(w_out), d, nhead, _lib.ptr(self.packed),
                                             _lib.stream()), "isf_pack_window_block")


def window_block(x, pw, in_proj_bias, table, out_proj_bias, ln, B, S, d, nhead, win, shift):
    """y = LN(x + out_proj(window attention(in_proj(x + pos)))) on [B*S*S, d] tokens, one kernel (A10/A11)."""
    _lib.require_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and tuple(x.shape) == (B * S * S, d)
    y = torch.empty_like(x)
    _lib.check(_lib.load().isf_window_block_forward(
        _lib.ptr(x), B, S, d, nhead, win, shift, _lib.ptr(pw.packed), _lib.ptr(in_proj_bias), _lib.ptr(table),
        _lib.ptr(out_proj_bias), _lib.ptr(ln.weight.detach()), _lib.ptr(ln.bias.detach()), float(ln.eps), _lib.ptr(y),
        _lib.stream()), "isf_window_block_forward")
    return y


def window_attention(qkv, B, S, d, nhead, win, shift):
    out = torch.empty((qkv.size(0), d), dtype=torch.float32, device=qkv.device)
    _lib.check(_lib.load().isf_window_attention_forward(_lib.ptr(qkv), B, S, d, nhead, win, shift, _lib.ptr(out),
                                                        _lib.stream()), "isf_window_attention_forward")
    return out


WINDOW_BLOCK_FUSED = True   # False: the three-launch form (qkv linear -> window attention -> out-projection linear)


def sstv2_forward(sst, bev, win, temperature=1000.0):
    """get_regions[i] + grid2region_att[i] (sst_v2.py:65-133) on a dense grid: [B, C, S, S] -> [B, d, S, S].
    Per encoder layer: the attention half in one kernel (isf_window_block_forward), the feed-forward half as two fused
    linears (GELU; residual + LayerNorm)."""
    from .dense_conv import SplitMap
    rows = None
    if isinstance(bev, SplitMap):        # a SECONDV2 stage's result in split form: token rows, no [B, C, S, S] map
        B, C, S = bev.B, bev.C, bev.H
        rows = bev.to_rows()
        dev = rows.device
    else:
        _lib.require_cuda(bev)
        B, C, S, _ = bev.shape
        dev = bev.device
    fused_io = (S * S) % 4 == 0          # channels-first loads / stores inside the GEMM need hw % 4 == 0
    d_model = sst.block_list[0].encoder_list[0].win_attn.self_attn.out_proj.in_features
    fused_block = WINDOW_BLOCK_FUSED and d_model == 128      # built (and measured faster) for the 128-wide level
    if hasattr(sst, "linear0"):
        c = _cache(sst, dev)
        if "linear0" not in c:
            c["linear0"] = PackedLinear(sst.linear0.weight, sst.linear0.bias)
        x = linear(rows if rows is not None else (bev.float() if fused_io else to_tokens(bev.float())), c["linear0"])
    elif rows is not None:
        x = rows
    elif fused_block:                    # the block kernel reads token rows
        x = to_tokens(bev.float())
    else:   # the first layer reads the map channels-first both as GEMM input and as residual
        x = bev.float().contiguous() if fused_io else to_tokens(bev.float())
    d = x.size(1)
    layers = [(shift, layer) for block in sst.block_list for shift, layer in enumerate(block.encoder_list)]
    for li, (shift, layer) in enumerate(layers):
        p = _encoder_layer_cache(layer, S, win, shift, temperature, dev, B)
        if fused_block:
            y = window_block(x, p["block"], p["in_bias"], p["table"], p["out_bias"], layer.norm1, B, S, d,
                             layer.win_attn.nhead, win, shift)
        else:
            qkv = linear(x, p["qkv"], table=p["table"], index=p["index"])
            att = window_attention(qkv, B, S, d, layer.win_attn.nhead, win, shift)
            y = linear(att, p["out"], residual=x, ln=layer.norm1)
        h = linear(y, p["l1"], act=ACT_GELU)
        last = li == len(layers) - 1 and fused_io
        x = linear(h, p["l2"], residual=y, ln=layer.norm2, out_nchw=(B, S, S) if last else None)
    return x if fused_io else from_tokens(x, B, S, S)


# ------------------------------------------------------------------------------------------------------ A8
_H2D_RING = {}   # (device, shape, dtype) -> [slot index, [(pinned buffer, event recorded behind its last copy)]]


def h2d_async(host_tensor, device, slots=4):
    """A small host tensor -> device WITHOUT draining the stream.  `.to(device)` from pageable memory is a staged copy
    the runtime performs only once the stream is idle: issued at the start of a forward it blocks the host until the
    previous forward has left the GPU, which then sits idle until the host has launched the next kernels (0.4 ms per
    forward at BASELINE configs[2], tools/host_lead.py / profiles/r03_host_lead.txt).  Here the bytes go through a ring
    of pinned buffers (a slot is reused only after the copy that last read it has run) and the copy is queued like a
    kernel."""
    t = host_tensor.detach().contiguous()
    device = torch.device(device)
    if device.type != "cuda":
        return t.to(device)
    key = (device, tuple(t.shape), t.dtype)
    ring = _H2D_RING.setdefault(key, [0, []])
    if len(ring[1]) < slots:
        ring[1].append((torch.empty(t.shape, dtype=t.dtype, pin_memory=True), torch.cuda.Event()))
        buf, ev = ring[1][-1]
    else:
        buf, ev = ring[1][ring[0] % slots]
        ring[0] += 1
        ev.synchronize()
    buf.copy_(t)
    out = buf.to(device, non_blocking=True)
    ev.record(torch.cuda.current_stream(device))
    return out


def p2g_camera_params(lidar2img, img_aug, lidar_aug, noise=None):
    """Fold the per-(sample, camera) 4x4 chain of img_point_sampling (fusion_encoder.py:1030-1047) into the 20
    floats isf_p2g_forward takes (float64 on the host, rounded once).  Batched over samples and cameras: a handful of
    host ops (the per-camera Python loop it replaces took 0.4 ms, during which the GPU sat idle)."""
    l2i = torch.as_tensor(lidar2img).detach().double().cpu()
    ia = torch.as_tensor(img_aug).detach().double().cpu()
    la = torch.as_tensor(lidar_aug).detach().double().cpu()
    B, ncam = l2i.shape[:2]
    rinv = torch.linalg.inv(la[:, :3, :3])                                   # [B, 3, 3]
    m = l2i[:, :, :3, :3] @ rinv[:, None]                                    # [B, ncam, 3, 3]
    v = l2i[:, :, :3, 3] - (m @ la[:, None, :3, 3, None]).squeeze(-1)        # [B, ncam, 3]
    if noise is not None:   # training-time jitter: one scalar per sample added to all camera-frame coordinates (:992-995)
        v = v + torch.as_tensor(noise, dtype=torch.float64).reshape(B, 1, 1)
    out = torch.cat([m.reshape(B, ncam, 9), v, ia[:, :, :2, :3].reshape(B, ncam, 6), ia[:, :, :2, 3]], dim=-1)
    return out.float().reshape(B * ncam, 20)


def p2g_sample(pillars, pillar_coors, img_feat, lidar2img, img_aug, lidar_aug, input_shape, bs, bev, num_cam=6, cam=None,
               out=None, split=False):
    """img_fv_to_bev (fusion_encoder.py:965-1013): pillars [M, T, >=3], pillar_coors [M, 4] (b, z, y, x),
    img_feat [bs*num_cam, C, H, W] -> [bs, C, bev, bev].  cam: p2g_camera_params(...) already on the device (the
    detector computes it before it queues the LiDAR branch, so the host work hides behind GPU work).
    split=True (C == 256): -> dense_conv.SplitMap, the form conv_fusion reads (isf_p2g_forward_split)."""
    _lib.require_cuda(img_feat)
    dev = img_feat.device
    pillars = pillars.float().contiguous()
    coors = pillar_coors.to(torch.int32).contiguous()
    nhwc = img_feat.float().permute(0, 2, 3, 1).contiguous()
    if cam is None:
        cam = h2d_async(p2g_camera_params(lidar2img, img_aug, lidar_aug), dev)
    C, H, W = img_feat.shape[1:]
    if split:
        from .dense_conv import SplitMap
        assert C == 256
        if out is not None:     # a caller-owned SplitMap (the detector's HIP-graph input)
            assert isinstance(out, SplitMap) and (out.B, out.C, out.H, out.W) == (bs, C, bev, bev)
            raw = out.data
            assert raw.numel() == bs * C * bev * bev * 4 and raw.is_contiguous()
        else:
            raw = torch.empty(bs * C * bev * bev * 4, dtype=torch.uint8, device=dev)
        _lib.check(_lib.load().isf_p2g_forward_split(_lib.ptr(pillars), pillars.size(2), pillars.size(1), _lib.ptr(coors),
                                                     pillars.size(0), _lib.ptr(nhwc), bs, num_cam, H, W, C, _lib.ptr(cam),
                                                     int(input_shape[0]), int(input_shape[1]), bev, _lib.ptr(raw),
                                                     _lib.stream()), "isf_p2g_forward_split")
        return out if out is not None else SplitMap(raw, bs, C, bev, bev)
    if out is None:
        out = torch.empty((bs, C, bev, bev), dtype=torch.float32, device=dev)
    assert tuple(out.shape) == (bs, C, bev, bev) and out.is_contiguous() and out.dtype == torch.float32
    _lib.check(_lib.load().isf_p2g_forward(_lib.ptr(pillars), pillars.size(2), pillars.size(1), _lib.ptr(coors),
                                           pillars.size(0), _lib.ptr(nhwc), bs, num_cam, H, W, C, _lib.ptr(cam),
                                           int(input_shape[0]), int(input_shape[1]), bev, _lib.ptr(out),
                                           _lib.stream()), "isf_p2g_forward")
    return out


# ----------------------------------------------------------------------------------------------------- A12
def instance_topk(heatmap, k=200, nms_kernel=3, pool1_classes=(8, 9), return_masked=False, as_int32=False):
    """fusion_encoder.py:1100-1131 -> top index % (H*W) [B, k] int64 (and the suppressed map when asked).
    as_int32: -> (top, raw, masked | None) with the kernel's own int32 index tensors (what isf_instance_gather /
    isf_head_query_init take), no conversion launches."""
    _lib.require_cuda(heatmap)
    assert nms_kernel == 3
    hm = heatmap.float().contiguous()
    B, K, H, W = hm.shape
    mask = 0
    for c in pool1_classes:
        mask |= 1 << c
    top = torch.empty((B, k), dtype=torch.int32, device=hm.device)
    raw = torch.empty((B, k), dtype=torch.int32, device=hm.device)
    masked = torch.empty((B, K * H * W), dtype=torch.float32, device=hm.device) if return_masked else None
    _lib.check(_lib.load().isf_instance_topk(_lib.ptr(hm), B, K, H, W, k, mask, _lib.ptr(top), _lib.ptr(raw),
                                             _lib.ptr(masked) if masked is not None else None, _lib.stream()),
               "isf_instance_topk")
    if as_int32:
        return top, raw, masked
    if return_masked:
        return top.long(), raw.long(), masked
    return top.long()


def head_query_init(top_index, top_raw, feat_tok, tok_of_cell, class_table, qpe_table, bev_pos, masked, num_classes):
    """TransFusionHeadV2.forward_single between the top-k and the first decoder layer in one launch
    (transfusion_head_v2.py:806-842, :888-890; isf_head_query_init).  top_index / top_raw [B, P] int32 ->
    query, qpe, x = query + qpe [B*P, E], query_pos [B, P, 2], top_index / query_labels [B, P] int64,
    query_heatmap_score [B, classes, P]."""
    _lib.require_cuda(feat_tok)
    B, P = top_index.shape
    E = feat_tok.shape[1]
    HW = bev_pos.shape[-2]
    dev = feat_tok.device
    f32 = dict(dtype=torch.float32, device=dev)
    query, qpe, x = (torch.empty((B * P, E), **f32) for _ in range(3))
    query_pos = torch.empty((B, P, 2), **f32)
    top64 = torch.empty((B, P), dtype=torch.int64, device=dev)
    labels = torch.empty((B, P), dtype=torch.int64, device=dev)
    score = torch.empty((B, num_classes, P), **f32)
    assert feat_tok.is_contiguous() and class_table.is_contiguous() and masked.is_contiguous() and bev_pos.is_contiguous()
    assert qpe_table is None or qpe_table.is_contiguous()
    _lib.check(_lib.load().isf_head_query_init(
        _lib.ptr(top_index), _lib.ptr(top_raw), B, P, HW, E, num_classes, _lib.ptr(feat_tok),
        _lib.ptr(tok_of_cell) if tok_of_cell is not None else None, _lib.ptr(class_table),
        _lib.ptr(qpe_table) if qpe_table is not None else None, _lib.ptr(bev_pos), _lib.ptr(masked), _lib.ptr(query),
        _lib.ptr(qpe), _lib.ptr(x), _lib.ptr(query_pos), _lib.ptr(top64), _lib.ptr(labels), _lib.ptr(score), _lib.stream()),
        "isf_head_query_init")
    return query, qpe, x, query_pos, top64, labels, score


def head_scatter_predictions(blocks, B, P, query_pos):
    """blocks: [(name, src [B*P, ld] fp32, col0, channels)] -> {name: [B, channels, P]} with `center` offset by
    query_pos [B, P, 2], and the new query positions [B, P, 2] (transfusion_head_v2.py:883-885;
    isf_head_scatter_predictions): one launch for every output of a decoder layer."""
    n = len(blocks)
    dev = blocks[0][1].device
    out = {name: torch.empty((B, ch, P), dtype=torch.float32, device=dev) for name, _, _, ch in blocks}
    center = [i for i, b in enumerate(blocks) if b[0] == "center"]
    qnext = torch.empty((B, P, 2), dtype=torch.float32, device=dev) if center else None
    vp = ctypes.c_void_p
    src = (vp * n)(*[_lib.ptr(b[1]) for b in blocks])
    dst = (vp * n)(*[_lib.ptr(out[b[0]]) for b in blocks])
    ld = (ctypes.c_int * n)(*[b[1].stride(0) for b in blocks])
    col0 = (ctypes.c_int * n)(*[b[2] for b in blocks])
    ch = (ctypes.c_int * n)(*[b[3] for b in blocks])
    _lib.check(_lib.load().isf_head_scatter_predictions(
        n, src, ld, col0, ch, dst, center[0] if center else -1, _lib.ptr(query_pos) if center else None,
        _lib.ptr(qnext) if center else None, B, P, _lib.stream()), "isf_head_scatter_predictions")
    return out, qnext


def decode_boxes(heatmap, query_score, query_labels, center, height, dim, rot, vel, cell, origin, post_center_range,
                 score_threshold):
    """TransFusionHeadV2.get_bboxes with nms_type=None incl. TransFusionBBoxCoder.decode(filter=True)
    (transfusion_head_v2.py:1286-1312, transfusion_bbox_coder.py:39-124) in one launch.  All inputs [B, *, P]
    (heatmap = logits); -> boxes [B, P, 7|9], scores [B, P], labels [B, P] int32 compacted per sample in proposal
    order, counts [B] int32 (device)."""
    _lib.require_cuda(heatmap)
    B, C, P = heatmap.shape
    ts = [t.float().contiguous() for t in (heatmap, query_score, center, height, dim, rot)]
    v = vel.float().contiguous() if vel is not None else None
    lab = query_labels.long().contiguous()
    dev = heatmap.device
    boxes = torch.empty((B, P, 9 if v is not None else 7), dtype=torch.float32, device=dev)
    scores = torch.empty((B, P), dtype=torch.float32, device=dev)
    labels = torch.empty((B, P), dtype=torch.int32, device=dev)
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    coder = (ctypes.c_float * 12)(cell[0], cell[1], origin[0], origin[1], *[float(r) for r in post_center_range],
                                  float(score_threshold or 0.0), 1.0 if score_threshold else 0.0)
    _lib.check(_lib.load().isf_decode_boxes(_lib.ptr(ts[0]), _lib.ptr(ts[1]), _lib.ptr(lab), _lib.ptr(ts[2]),
                                            _lib.ptr(ts[3]), _lib.ptr(ts[4]), _lib.ptr(ts[5]),
                                            _lib.ptr(v) if v is not None else None, B, C, P, P, coder, _lib.ptr(boxes),
                                            _lib.ptr(scores), _lib.ptr(labels), _lib.ptr(counts), _lib.stream()),
               "isf_decode_boxes")
    return boxes, scores, labels, counts


def gather_instances(x_scene, top_idx, bev_size):
    """x_ins [B, E, Q] and the (x, y) query positions of fusion_encoder.py:1133-1141 (create_2D_grid cell centres,
    axes swapped)."""
    B, E = x_scene.shape[:2]
    x_ins = x_scene.reshape(B, E, -1).gather(2, top_idx[:, None, :].expand(-1, E, -1))
    qx = (top_idx % bev_size).float() + 0.5
    qy = torch.div(top_idx, bev_size, rounding_mode="floor").float() + 0.5
    return x_ins, torch.stack([qx, qy], dim=-1)


# ----------------------------------------------------------------------------------------------------- A13
def attention(q, k, v, B, Lq, Lk, E, nhead, ldkv=None):
    out = torch.empty((B * Lq, E), dtype=torch.float32, device=q.device)
    _lib.check(_lib.load().isf_attention_forward(_lib.ptr(q), q.stride(0), _lib.ptr(k), _lib.ptr(v),
                                                 ldkv if ldkv is not None else k.stride(0), B, Lq, Lk, E, nhead,
                                                 _lib.ptr(out), out.stride(0), _lib.stream()), "isf_attention_forward")
    return out


def msda(value, offsets, logits, ref, B, Q, nhead, hd, npts, H, W):
    out = torch.empty((B * Q, nhead * hd), dtype=torch.float32, device=value.device)
    _lib.check(_lib.load().isf_msda_forward(_lib.ptr(value), _lib.ptr(offsets), _lib.ptr(logits), _lib.ptr(ref), B, Q,
                                            nhead, hd, npts, H, W, _lib.ptr(out), _lib.stream()), "isf_msda_forward")
    return out


class MultiScaleDeformableAttnFunction(torch.autograd.Function):
    """MultiScaleDeformableAttnFunction_fp32 (multi_scale_deformable_attn_function.py:90-163) over the two native ops with
    mmcv's argument lists: forward = isf_ms_deform_attn_forward, backward = isf_ms_deform_attn_backward (grad_value,
    grad_sampling_loc, grad_attn_weight zero-filled here and accumulated by the kernel, as the reference does :142-160)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, value, spatial_shapes, level_start_index, sampling_locations, attention_weights):
        _lib.require_cuda(value, sampling_locations, attention_weights)
        v = value.float().contiguous()
        loc, aw = sampling_locations.float().contiguous(), attention_weights.float().contiguous()
        ss = spatial_shapes.to(v.device).long().contiguous()
        ls = level_start_index.to(v.device).long().contiguous()
        B, S, M, D = v.shape
        Q, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
        assert tuple(loc.shape) == (B, Q, M, L, P, 2) and tuple(aw.shape) == (B, Q, M, L, P) and ss.shape == (L, 2)
        out = torch.empty((B, Q, M * D), dtype=torch.float32, device=v.device)
        _lib.check(_lib.load().isf_ms_deform_attn_forward(_lib.ptr(v), _lib.ptr(ss), _lib.ptr(ls), _lib.ptr(loc),
                                                          _lib.ptr(aw), B, S, M, D, Q, L, P, _lib.ptr(out),
                                                          _lib.stream()), "isf_ms_deform_attn_forward")
        ctx.save_for_backward(v, ss, ls, loc, aw)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_output):
        v, ss, ls, loc, aw = ctx.saved_tensors
        B, S, M, D = v.shape
        Q, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
        go = grad_output.float().contiguous()
        gv, gl, gw = torch.zeros_like(v), torch.zeros_like(loc), torch.zeros_like(aw)
        _lib.check(_lib.load().isf_ms_deform_attn_backward(_lib.ptr(v), _lib.ptr(ss), _lib.ptr(ls), _lib.ptr(loc),
                                                           _lib.ptr(aw), _lib.ptr(go), B, S, M, D, Q, L, P, _lib.ptr(gv),
                                                           _lib.ptr(gl), _lib.ptr(gw), _lib.stream()),
                   "isf_ms_deform_attn_backward")
        return gv, None, None, gl, gw


def ms_deform_attn(value, spatial_shapes, level_start_index, sampling_locations, attention_weights):
    """MultiScaleDeformableAttnFunction.apply with mmcv's argument list
    (multi_scale_deformable_attn_function.py:84-163; im2col_step has no counterpart): value [B, S, M, D],
    spatial_shapes [L, 2] long (h, w), level_start_index [L] long, sampling_locations [B, Q, M, L, P, 2],
    attention_weights [B, Q, M, L, P] -> [B, Q, M*D]; differentiable in value, locations and weights."""
    return MultiScaleDeformableAttnFunction.apply(value, spatial_shapes, level_start_index, sampling_locations,
                                                  attention_weights)


def ingroup_indices(group_inds):
    """IngroupIndicesFunction / get_inner_win_inds_cuda (ops/sst/sst_ops.py:197-211): group_inds [N] long ->
    out_inds [N] long, out_inds[i] = number of earlier elements of the same group (deterministic)."""
    _lib.require_cuda(group_inds)
    g = group_inds.long().contiguous()
    out = torch.empty_like(g)
    _lib.check(_lib.load().isf_ingroup_indices(_lib.ptr(g), g.numel(), _lib.ptr(out), _lib.stream()),
               "isf_ingroup_indices")
    return out


class MSDAFunction(torch.autograd.Function):
    """MultiScaleDeformableAttnFunction of the reference (multi_scale_deformable_attn_function.py; one level), taking
    the raw sampling offsets and attention logits: forward = isf_msda_forward, backward = isf_msda_backward
    -> gradients for value, offsets and logits (validated on an MI355X: tests/test_gpu_train.py, tests/test_gpu_widened.py)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, value, offsets, logits, ref, B, Q, nhead, hd, npts, H, W):
        _lib.require_cuda(value, offsets, logits, ref)
        args = [t.detach().float().contiguous() for t in (value, offsets, logits, ref)]
        ctx.save_for_backward(*args)
        ctx.dims = (B, Q, nhead, hd, npts, H, W)
        return msda(*args, B, Q, nhead, hd, npts, H, W)

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_out):
        value, offsets, logits, ref = ctx.saved_tensors
        B, Q, nhead, hd, npts, H, W = ctx.dims
        g = grad_out.contiguous().float()
        gv, goff, glog = torch.empty_like(value), torch.empty_like(offsets), torch.empty_like(logits)
        _lib.check(_lib.load().isf_msda_backward(_lib.ptr(value), _lib.ptr(offsets), _lib.ptr(logits), _lib.ptr(ref),
                                                 _lib.ptr(g), B, Q, nhead, hd, npts, H, W, _lib.ptr(gv), _lib.ptr(goff),
                                                 _lib.ptr(glog), _lib.stream()), "isf_msda_backward")
        return (gv, goff, glog) + (None,) * 8


class AttentionFunction(torch.autograd.Function):
    """softmax(q k^T / sqrt(hd)) v of nn.MultiheadAttention (fusion_encoder.py:371-470) with a gradient: forward =
    isf_attention_forward, backward = isf_attention_backward (validated on an MI355X: tests/test_gpu_train.py, tests/test_gpu_widened.py).
    q [B*Lq, E], k / v [B*Lk, E] row-major."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, q, k, v, B, Lq, Lk, E, nhead, dropout_p=0.0, seed=0):
        """dropout_p > 0 (training): nn.MultiheadAttention's dropout on the attention probabilities
        (fusion_encoder.py:458), decided per (sample, head, query, key) by a hash of `seed` that the backward pass
        recomputes (isf_attention_forward_dropout / _backward_dropout; attention_keep_mask restates it)."""
        _lib.require_cuda(q, k, v)
        q, k, v = [t.detach().float().contiguous() for t in (q, k, v)]
        if dropout_p > 0.0:
            out = torch.empty((B * Lq, E), dtype=torch.float32, device=q.device)
            _lib.check(_lib.load().isf_attention_forward_dropout(
                _lib.ptr(q), q.stride(0), _lib.ptr(k), _lib.ptr(v), k.stride(0), B, Lq, Lk, E, nhead, float(dropout_p),
                int(seed), _lib.ptr(out), out.stride(0), _lib.stream()), "isf_attention_forward_dropout")
        else:
            out = attention(q, k, v, B, Lq, Lk, E, nhead)
        ctx.save_for_backward(q, k, v, out)
        ctx.dims = (B, Lq, Lk, E, nhead)
        ctx.drop = (float(dropout_p), int(seed))
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_out):
        q, k, v, out = ctx.saved_tensors
        B, Lq, Lk, E, nhead = ctx.dims
        g = grad_out.contiguous().float()
        gq, gk, gv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        p, seed = ctx.drop
        if p > 0.0:
            _lib.check(_lib.load().isf_attention_backward_dropout(
                _lib.ptr(q), q.stride(0), _lib.ptr(k), _lib.ptr(v), k.stride(0), _lib.ptr(out), _lib.ptr(g), g.stride(0),
                B, Lq, Lk, E, nhead, p, seed, _lib.ptr(gq), gq.stride(0), _lib.ptr(gk), _lib.ptr(gv), gk.stride(0),
                _lib.stream()), "isf_attention_backward_dropout")
        else:
            _lib.check(_lib.load().isf_attention_backward(
                _lib.ptr(q), q.stride(0), _lib.ptr(k), _lib.ptr(v), k.stride(0), _lib.ptr(out), _lib.ptr(g), g.stride(0),
                B, Lq, Lk, E, nhead, _lib.ptr(gq), gq.stride(0), _lib.ptr(gk), _lib.ptr(gv), gk.stride(0), _lib.stream()),
                "isf_attention_backward")
        return (gq, gk, gv) + (None,) * 7


def attention_keep_mask(seed, B, nhead, Lq, Lk, p, device="cpu"):
    """[B, nhead, Lq, Lk] bool: the keep / drop decisions of isf_attention_forward_dropout (AttnDrop in isf_common.h) restated
    in torch -- the finaliser of MurmurHash3 over seed ^ (b*heads + head) << 48 | query << 24 | key, upper 32 bits >= p * 2^32.
    int64 arithmetic wraps like uint64; logical right shifts are emulated by masking the sign extension."""
    def shr33(x):
        return (x >> 33) & 0x7FFFFFFF

    def as_i64(v):
        v &= (1 << 64) - 1
        return v - (1 << 64) if v >= (1 << 63) else v
    bh = torch.arange(B * nhead, dtype=torch.int64, device=device).view(B, nhead, 1, 1)
    i = torch.arange(Lq, dtype=torch.int64, device=device).view(1, 1, Lq, 1)
    j = torch.arange(Lk, dtype=torch.int64, device=device).view(1, 1, 1, Lk)
    x = (bh << 48) | (i << 24) | j
    x = x ^ as_i64(int(seed))
    x = x ^ shr33(x)
    x = x * as_i64(0xff51afd7ed558ccd)
    x = x ^ shr33(x)
    x = x * as_i64(0xc4ceb9fe1a85ec53)
    x = x ^ shr33(x)
    hi = (x >> 32) & 0xFFFFFFFF
    t = float(p) * 4294967296.0
    thresh = 4294967295 if t >= 4294967295.0 else int(t)
    return hi >= thresh


class WindowAttentionFunction(torch.autograd.Function):
    """the window attention core of SST's BasicShiftBlockV2 on a dense grid (sst_basic_block_v2.py:41-75) with a
    gradient: forward = isf_window_attention_forward, backward = isf_window_attention_backward.
    qkv [B*S*S, 3d] -> [B*S*S, d]."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, qkv, B, S, d, nhead, win, shift):
        _lib.require_cuda(qkv)
        qkv = qkv.detach().float().contiguous()
        ctx.save_for_backward(qkv)
        ctx.dims = (B, S, d, nhead, win, shift)
        return window_attention(qkv, B, S, d, nhead, win, shift)

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_out):
        (qkv,) = ctx.saved_tensors
        B, S, d, nhead, win, shift = ctx.dims
        g = grad_out.contiguous().float()
        gqkv = torch.empty_like(qkv)
        _lib.check(_lib.load().isf_window_attention_backward(_lib.ptr(qkv), _lib.ptr(g), B, S, d, nhead, win, shift,
                                                             _lib.ptr(gqkv), _lib.stream()),
                   "isf_window_attention_backward")
        return (gqkv,) + (None,) * 6


def _pos_embed(mod, xy):
    """PositionEmbeddingLearned (fusion_encoder.py:173-189): Conv1d(k=1) / BN1d / ReLU / Conv1d(k=1) on [B, N, 2] ->
    [B, N, E].  In eval mode the two kernel-size-1 convolutions are what they are -- two small GEMMs with the BN folded
    (library GEMMs: K = 2 and K = E are far below the MFMA tile of the fused linear kernel) -- instead of four MIOpen
    launches with layout transposes; training mode runs the stock modules (batch statistics)."""
    seq = mod.position_embedding_head
    if seq.training:
        return seq(xy.transpose(1, 2).contiguous()).transpose(1, 2).contiguous()
    from .norm import fold_bn
    c1, bn, _, c2 = seq
    scale, shift = fold_bn(bn)
    w1 = c1.weight[:, :, 0] * scale[:, None]                               # [E, 2]
    b1 = (c1.bias if c1.bias is not None else 0.0) * scale + shift
    h = torch.relu(torch.addmm(b1, xy.reshape(-1, xy.shape[-1]).float(), w1.t()))
    out = torch.addmm(c2.bias, h, c2.weight[:, :, 0].t()) if c2.bias is not None else h @ c2.weight[:, :, 0].t()
    return out.view(*xy.shape[:-1], -1)


def _ins_context_pack(mod, dev, bev_size):
    """per-module cache of InsContextAtt: position tables of the create_2D_grid lattice + packed weights"""
    c = _cache(mod, dev)
    E = mod.layers[0].self_attn.out_proj.in_features
    if "key_pos" not in c:
        g = torch.linspace(0, bev_size - 1, bev_size, device=dev) + 0.5
        bx, by = torch.meshgrid(g, g, indexing="ij")
        bev_pos = torch.stack([bx, by], 0).view(1, 2, -1).permute(0, 2, 1)
        c["key_pos"] = _pos_embed(mod.key_pos_embed, bev_pos / bev_size)[0].contiguous()          # [HW, E]
        c["query_pos_table"] = _pos_embed(mod.query_pos_embed, bev_pos / bev_size)[0].contiguous()  # [HW, E]
        c["layers"] = []
        for l in mod.layers:
            sa, ca = l.self_attn, l.cross_attn
            w, b = sa.in_proj_weight.detach().float(), sa.in_proj_bias.detach().float()
            c["layers"].append(dict(
                qk=PackedLinear(w[:2 * E], b[:2 * E]), v=PackedLinear(w[2 * E:], b[2 * E:]),
                out=PackedLinear(sa.out_proj.weight, sa.out_proj.bias),
                value=PackedLinear(ca.value_proj.weight, ca.value_proj.bias),
                # value_proj(scene + key_pos) = value_proj(scene) + key_pos W^T: the position term is input independent
                # -> a per-cell table added in the GEMM epilogue (the sum was a transposing 33-MB pass per forward)
                vtable=(c["key_pos"].double() @ ca.value_proj.weight.detach().double().t()).float().contiguous(),
                off=PackedLinear(ca.sampling_offsets.weight, ca.sampling_offsets.bias),
                aw=PackedLinear(ca.attention_weights.weight, ca.attention_weights.bias),
                oproj=PackedLinear(ca.output_proj.weight, ca.output_proj.bias),
                l1=PackedLinear(l.linear1.weight, l.linear1.bias), l2=PackedLinear(l.linear2.weight, l.linear2.bias)))
    return c


def mined_instances(mod, top32, scene, bev_size):
    """fusion_encoder.py:1133-1141 + the preamble of InsContextAtt.forward (:800-812) in one launch
    (isf_instance_gather): top32 [B, Q] int32 = cells of the transposed map as isf_instance_topk returns them, scene
    [B, E, S, S] the un-transposed map, mod = the InsContextAtt module (its query_pos_embed table).  -> dict(top [B, Q]
    int64, cells [B, Q] int64, tokens / qpe / tokens_pos [B*Q, E], query_pos [B, Q, 2], ref [B*Q, 2])."""
    _lib.require_cuda(scene)
    dev = scene.device
    B, Q = top32.shape
    E = scene.shape[1]
    assert scene.shape[2] == bev_size and scene.shape[3] == bev_size
    c = _ins_context_pack(mod, dev, bev_size)
    scene = scene.float().contiguous()
    f32 = dict(dtype=torch.float32, device=dev)
    r = dict(top=torch.empty((B, Q), dtype=torch.int64, device=dev), cells=torch.empty((B, Q), dtype=torch.int64, device=dev),
             tokens=torch.empty((B * Q, E), **f32), qpe=torch.empty((B * Q, E), **f32),
             tokens_pos=torch.empty((B * Q, E), **f32), query_pos=torch.empty((B, Q, 2), **f32),
             ref=torch.empty((B * Q, 2), **f32))
    _lib.check(_lib.load().isf_instance_gather(_lib.ptr(top32), B, Q, bev_size, E, _lib.ptr(scene),
                                               _lib.ptr(c["query_pos_table"]), _lib.ptr(r["top"]), _lib.ptr(r["cells"]),
                                               _lib.ptr(r["tokens"]), _lib.ptr(r["qpe"]), _lib.ptr(r["tokens_pos"]),
                                               _lib.ptr(r["query_pos"]), _lib.ptr(r["ref"]), _lib.stream()),
               "isf_instance_gather")
    return r


def ins_context_att(mod, x_ins, query_pos, scene, bev_size, query_cells=None, mined=None):
    """InsContextAtt.forward (fusion_encoder.py:795-830), eval mode.  x_ins [B, E, Q], query_pos [B, Q, 2] (x, y),
    scene [B, E, H, W] = the reference's `x_scene.permute(0, 1, 3, 2)` (:806; the caller holds the map in that
    orientation already) -> [B, E, Q] (token-major [B*Q, E] with mined=).  query_cells [B, Q] long (optional): the queries sit on cell centres of the
    create_2D_grid lattice, query_pos = bev_pos[query_cells] -- then their position embedding is a row of a per-cell
    table computed once (the same MLP on the same inputs) instead of two GEMMs + glue per forward.  mined = the dict of
    mined_instances() (then x_ins / query_pos / query_cells are not read)."""
    _lib.require_cuda(scene)
    dev = scene.device
    c = _ins_context_pack(mod, dev, bev_size)
    if mined is not None:
        (B, Q), E = mined["top"].shape, scene.shape[1]
    else:
        B, E, Q = x_ins.shape
    H, W = scene.shape[2:]
    ck = ("cell_index", B, H * W)
    if ck not in c:
        c[ck] = torch.arange(H * W, device=dev, dtype=torch.int32).repeat(B)
    cell = c[ck]
    # read channels-first inside the GEMM (no token copy) when the kernel's 4-row groups stay inside a sample
    scene = scene.float().contiguous() if (H * W) % 4 == 0 else to_tokens(scene.float())
    qk_first = None
    if mined is not None:   # mined_instances() has gathered tokens, positions and embeddings already
        out, ref, qpe, qk_first = mined["tokens"], mined["ref"], mined["qpe"], mined["tokens_pos"]
    else:
        out = x_ins.transpose(1, 2).reshape(B * Q, E).contiguous()
        ref = (query_pos / bev_size).reshape(B * Q, 2).contiguous()
        if query_cells is not None:
            qpe = c["query_pos_table"][query_cells.reshape(-1)]
        else:
            qpe = _pos_embed(mod.query_pos_embed, ref.view(B, Q, 2)).reshape(B * Q, E)
    for l, p in zip(mod.layers, c["layers"]):
        nhead, npts = l.cross_attn.n_heads, l.cross_attn.n_points
        qk_in = qk_first if qk_first is not None else out + qpe
        qk_first = None
        qk = linear(qk_in, p["qk"])
        v = linear(out, p["v"])
        # q = columns [0, E), k = columns [E, 2E) of qk (row stride 2E); v has row stride E, and the C entry takes
        # one stride for k and v -> give k its own buffer (200 rows)
        att = attention(qk, qk[:, E:].contiguous(), v, B, Q, Q, E, nhead)
        out = linear(att, p["out"], residual=out, ln=l.norm2)
        q = out + qpe
        value = linear(scene, p["value"], table=p["vtable"], index=cell)
        off = linear(q, p["off"])
        aw = linear(q, p["aw"])
        t2 = msda(value, off, aw, ref, B, Q, nhead, E // nhead, npts, H, W)
        out = linear(t2, p["oproj"], residual=out, ln=l.norm1)
        h = linear(out, p["l1"], act=ACT_RELU)
        out = linear(h, p["l2"], residual=out, ln=l.norm3)
    if mined is not None:
        return out   # token-major [B*Q, E]: what instance_to_scene reads (the caller of mined= is the encoder)
    return out.view(B, Q, E).transpose(1, 2).contiguous()


# ----------------------------------------------------------------------------------------------------- A14
def channel_attention(query_scene, query_ins):
    _lib.require_cuda(query_scene)
    qs, qi = query_scene.float().contiguous(), query_ins.float().contiguous()
    B, C, H, W = qs.shape
    assert H == W and qi.shape == qs.shape
    out = torch.empty_like(qs)
    _lib.check(_lib.load().isf_channel_attention_forward(_lib.ptr(qs), _lib.ptr(qi), B * C, H, _lib.ptr(out),
                                                         _lib.stream()), "isf_channel_attention_forward")
    return out


def instance_to_scene(mod, query, x_ins, scene_feats, bev_size):
    """Instane2SceneAtt.forward (fusion_encoder.py:480-502), eval mode: query [B, E, H, W] = conv_ins(bev),
    x_ins [B, E, Q] (or token-major [B*Q, E], as ins_context_att(mined=...) returns it), scene_feats [B, E, H, W] (the
    Grid-to-Region output)."""
    _lib.require_cuda(query)
    B, E, H, W = query.shape
    tokens = x_ins.dim() == 2
    Q = x_ins.size(0) // B if tokens else x_ins.size(2)
    c = _cache(mod, query.device)
    if "q" not in c:
        a = mod.multihead_attn
        w, b = a.in_proj_weight.detach().float(), a.in_proj_bias.detach().float()
        c["q"] = PackedLinear(w[:E], b[:E])
        c["kv"] = PackedLinear(w[E:], b[E:])
        c["out"] = PackedLinear(a.out_proj.weight, a.out_proj.bias)
    fused_io = (H * W) % 4 == 0
    xq = query.float().contiguous() if fused_io else to_tokens(query.float())   # read channels-first in the GEMMs
    xk = x_ins if tokens else x_ins.transpose(1, 2).reshape(B * Q, E).contiguous()
    qp = linear(xq, c["q"])
    kv = linear(xk, c["kv"])
    att = attention(qp, kv, kv[:, E:], B, H * W, Q, E, mod.nhead, ldkv=2 * E)   # k | v share rows of kv
    y = linear(att, c["out"], residual=xq, ln=mod.norm, out_nchw=(B, H, W) if fused_io else None)
    query_ins = y if fused_io else from_tokens(y, B, H, W)
    return channel_attention(scene_feats, query_ins)
