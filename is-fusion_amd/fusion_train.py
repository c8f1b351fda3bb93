"""Training-mode (differentiable) forms of the HSF / IGF rows -- SURVEY.md section 8f #2.

The inference path (fusion_ops.py) runs every transformer piece through kernels with fused epilogues and cached packed
weights.  Training needs gradients, so here the same arithmetic is composed from autograd Functions whose FORWARD is
the HIP kernel and whose BACKWARD is

  * fused linear            dX on the same f16x3 MFMA GEMM (transposed weight), dW = dY^T X and db as plain library
                            GEMMs / reductions (rocBLAS through torch.matmul: the rules' "plain library GEMM");
  * window / small-key attention, MSDA      the hand-written backward kernels (isf_attention_bwd.hip, isf_fusion.hip);
  * per-channel map attention               forward isf_channel_attention_forward, backward batched GEMMs;
  * Point-to-Grid                           isf_p2g_backward (scatter with fp32 atomics);

bias / position-table / GELU / residual / LayerNorm epilogues, the 3x3 convolutions + BatchNorm (stock, as the north_star
prescribes) and the top-k instance mining (indices: no gradient, as in the reference) are stock differentiable torch ops.
The reference does the same composition with torch modules (sst_basic_block_v2.py:77-126, fusion_encoder.py:480-502,
:560-600, :795-830).  Mixed precision: bf16 / fp16 inputs are accepted and promoted -- the kernels compute fp32-class
(>= the reference's autocast arithmetic); gradients come back in the input dtype.

Training-time stochastic ops of the reference: the residual / feed-forward Dropouts (p = 0.1) of
DeformableTransformerDecoderLayer (dropout1..4, fusion_encoder.py:604-668) and Instane2SceneAtt (:478, :492) and the
Point-to-Grid `random_noise` jitter (:992-995) are applied in training mode.  NOT applied: the dropout on the attention
PROBABILITIES inside the two nn.MultiheadAttention modules (:614, :476 -> :458) -- the flash-style HIP attention core
never materialises the probability matrix; reproducing it needs an in-kernel random stream (not built).

No CPU fallback: tensors must live on a GPU."""
import torch
import torch.nn.functional as F

from . import _lib
from . import fusion_ops as ops

# the HIP kernels compute in fp32: under torch.autocast (the reference trains with mixed precision) inputs are cast
# to fp32 on the way in and autocast is off inside forward / backward
_amp_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_amp_bwd = torch.amp.custom_bwd(device_type="cuda")


def _weight_grad(g, x):
    """dW = g^T x for g [M, N], x [M, K] with M in the tens of thousands and N, K <= 1024: the library GEMM picks ONE
    256 x 256 macro tile for the whole reduction (7 ms for 64800 x 128 x 128: tools/train_step.py profile), so the rows
    are cut into chunks reduced side by side (batched GEMM) and summed."""
    M = g.shape[0]
    chunk = 2048
    S = M // chunk
    if S < 4:
        return g.t().matmul(x)
    main = S * chunk
    gw = torch.bmm(g[:main].view(S, chunk, -1).transpose(1, 2), x[:main].view(S, chunk, -1)).sum(0)
    if main < M:
        gw = gw + g[main:].t().matmul(x[main:])
    return gw


class _PackedGrad(torch.autograd.Function):
    """identity whose backward hands on a CONTIGUOUS gradient"""

    @staticmethod
    def forward(ctx, x):
        return x

    @staticmethod
    def backward(ctx, g):
        return g.contiguous()


def _contiguous_inputs(mod, args):
    return tuple(a.contiguous() if torch.is_tensor(a) else a for a in args)


def _packed_output_grad(mod, args, out):
    return _PackedGrad.apply(out) if mod.training else out


def pack_stock_convs(module):
    """Stock Conv2d / ConvTranspose2d layers of the training path see packed tensors only: their inputs are made
    contiguous and so is the gradient arriving at their outputs.  The token-major HIP ops around them hand over
    permuted VIEWS ([B, H, W, C] storage seen as [B, C, H, W]), for which MIOpen falls back to its
    `naive_conv_ab_nonpacked_*` kernels -- 20-40 ms per call, 80 % of a training step before this hook.
    Registered once per conv (module-level functions: the model stays picklable), handles kept in
    `module._isf_conv_hooks`; `unpack_stock_convs` removes them.  The output hook is the identity in eval mode."""
    if getattr(module, "_isf_convs_packed", False):
        return module
    handles = []
    for m in module.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)) and not getattr(m, "_isf_packed_io", False):
            handles.append(m.register_forward_pre_hook(_contiguous_inputs))
            handles.append(m.register_forward_hook(_packed_output_grad))
            m._isf_packed_io = True
    module.__dict__["_isf_conv_hooks"] = handles
    module.__dict__["_isf_convs_packed"] = True
    return module


def unpack_stock_convs(module):
    """remove the hooks of pack_stock_convs (e.g. before exporting an inference model)"""
    for h in module.__dict__.pop("_isf_conv_hooks", []):
        h.remove()
    for m in module.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
            m.__dict__.pop("_isf_packed_io", None)
    module.__dict__["_isf_convs_packed"] = False
    return module


class LinearFunction(torch.autograd.Function):
    """y = x W^T (+ b).  x [M, K], weight [N, K] (nn.Linear layout), bias [N] or None."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, weight, bias):
        _lib.require_cuda(x, weight)
        xd, wd = x.detach().float().contiguous(), weight.detach().float().contiguous()
        y = ops.linear(xd, ops.PackedLinear(wd, bias.detach() if bias is not None else None))
        ctx.save_for_backward(xd, wd)
        ctx.has_bias = bias is not None
        ctx.in_dtype = x.dtype
        return y

    @staticmethod
    @_amp_bwd
    def backward(ctx, gy):
        xd, wd = ctx.saved_tensors
        g = gy.contiguous().float()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gs, sc = _lib.grad_rescale(g)           # gradients are tiny: keep the GEMM's f16 halves in range (exact)
            gx = (ops.linear(gs, ops.PackedLinear(wd, transposed=True)) * sc[1]).to(ctx.in_dtype)   # dX = dY W  (HIP GEMM)
        if ctx.needs_input_grad[1]:
            gw = _weight_grad(g, xd)                                                      # dW = dY^T X (library GEMM)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.sum(0)
        return gx, gw, gb


def linear(x, lin, **kw):
    """nn.Linear `lin` applied to [M, K] rows with a gradient; K % 32 == 0 and N % 16 == 0 (the kernel's tiles)."""
    return LinearFunction.apply(x, lin.weight, lin.bias)


def linear_w(x, weight, bias=None):
    return LinearFunction.apply(x, weight, bias)


class ChannelAttentionFunction(torch.autograd.Function):
    """out = query_scene + softmax(query_scene query_ins^T) query_ins per (batch, channel) map
    (fusion_encoder.py:495-500).  Forward = isf_channel_attention_forward; backward recomputes the probabilities and
    runs four batched GEMMs."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, qs, qi):
        qs, qi = qs.detach().float().contiguous(), qi.detach().float().contiguous()
        ctx.save_for_backward(qs, qi)
        return ops.channel_attention(qs, qi)

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        qs, qi = ctx.saved_tensors
        B, C, H, W = qs.shape
        a, b, go = qs.view(B * C, H, W), qi.view(B * C, H, W), g.contiguous().float().view(B * C, H, W)
        p = torch.softmax(torch.bmm(a, b.transpose(1, 2)), -1)          # [BC, H, H]
        gp = torch.bmm(go, b.transpose(1, 2))                            # d loss / d P
        gs = p * (gp - (gp * p).sum(-1, keepdim=True))                   # through the softmax
        ga = go + torch.bmm(gs, b)                                       # query_scene: identity + scores
        gb = torch.bmm(p.transpose(1, 2), go) + torch.bmm(gs.transpose(1, 2), a)
        return ga.view(B, C, H, W), gb.view(B, C, H, W)


class P2GFunction(torch.autograd.Function):
    """img_fv_to_bev (fusion_encoder.py:965-1013) with a gradient towards the camera feature map."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, img_feat, pillars, pillar_coors, cam, input_shape, bs, bev, num_cam):
        _lib.require_cuda(img_feat)
        dev = img_feat.device
        pil = pillars.detach().float().contiguous()
        coors = pillar_coors.to(torch.int32).contiguous()
        nhwc = img_feat.detach().float().permute(0, 2, 3, 1).contiguous()
        C, H, W = img_feat.shape[1:]
        out = torch.empty((bs, C, bev, bev), dtype=torch.float32, device=dev)
        _lib.check(_lib.load().isf_p2g_forward(_lib.ptr(pil), pil.size(2), pil.size(1), _lib.ptr(coors), pil.size(0),
                                               _lib.ptr(nhwc), bs, num_cam, H, W, C, _lib.ptr(cam), int(input_shape[0]),
                                               int(input_shape[1]), bev, _lib.ptr(out), _lib.stream()), "isf_p2g_forward")
        ctx.save_for_backward(pil, coors, cam)
        ctx.dims = (bs, num_cam, H, W, C, int(input_shape[0]), int(input_shape[1]), bev)
        ctx.in_dtype = img_feat.dtype
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        pil, coors, cam = ctx.saved_tensors
        bs, num_cam, H, W, C, ih, iw, bev = ctx.dims
        go = g.contiguous().float()
        gi = torch.empty((bs * num_cam, H, W, C), dtype=torch.float32, device=go.device)
        _lib.check(_lib.load().isf_p2g_backward(_lib.ptr(pil), pil.size(2), pil.size(1), _lib.ptr(coors), pil.size(0), bs,
                                                num_cam, H, W, C, _lib.ptr(cam), ih, iw, bev, _lib.ptr(go), _lib.ptr(gi),
                                                _lib.stream()), "isf_p2g_backward")
        return (gi.permute(0, 3, 1, 2).to(ctx.in_dtype),) + (None,) * 7


def p2g_sample(pillars, pillar_coors, img_feat, lidar2img, img_aug, lidar_aug, input_shape, bs, bev, num_cam=6,
               noise=None):
    """noise: per-sample camera-frame offsets of the reference's training-time `random_noise` jitter
    (fusion_encoder.py:992-995; ISFusionEncoder.forward_train draws them), or None"""
    cam = ops.p2g_camera_params(lidar2img, img_aug, lidar_aug, noise).to(img_feat.device)
    return P2GFunction.apply(img_feat, pillars, pillar_coors, cam, tuple(input_shape), bs, bev, num_cam)


# ------------------------------------------------------------------------------------------------ A10 / A11
class _WindowTableAdd(torch.autograd.Function):
    """qkv + tab[slot(token)] for the dense grid's window-position table ([win * win, 3d], slot(y, x) =
    ((y + off) % win) * win + (x + off) % win).  autograd's own backward of `tab[index]` is a sort-based
    indexing_backward_kernel over all B * S * S tokens (1 ms per layer at S = 180: four of them were the second largest
    kernel of a training step, profiles/r05_train_step_mid.txt); the slots are a regular pattern, so the table gradient
    is a strided sum of the token gradient: fold [B, S, S, C] over (B, S / win, S / win) and roll by the window offset."""

    @staticmethod
    def forward(ctx, qkv, tab, index, B, S, win, off):
        ctx.geom = (B, S, win, off)
        return qkv + tab[index.long().repeat(B)]

    @staticmethod
    def backward(ctx, g):
        B, S, win, off = ctx.geom
        C = g.size(1)
        gt = None
        if ctx.needs_input_grad[1]:
            t = g.reshape(B, S // win, win, S // win, win, C).sum(dim=(0, 1, 3))     # [y % win, x % win, C]; reshape: autograd
            gt = torch.roll(t, shifts=(off % win, off % win), dims=(0, 1)).reshape(win * win, C)   # may hand over a strided g
        return (g if ctx.needs_input_grad[0] else None), gt, None, None, None, None, None


def sstv2_forward(sst, bev, win, temperature=1000.0):
    """get_regions[i] + grid2region_att[i] (sst_v2.py:65-133, sst_basic_block_v2.py:77-126) on the dense grid with
    gradients: [B, C, S, S] -> [B, d, S, S]."""
    _lib.require_cuda(bev)
    B, C, S, _ = bev.shape
    x = ops.to_tokens(bev.float())
    if hasattr(sst, "linear0"):
        x = linear(x, sst.linear0)
    d = x.size(1)
    for block in sst.block_list:
        for shift, layer in enumerate(block.encoder_list):
            attn = layer.win_attn.self_attn
            index, pos = ops._window_tables(S, win, shift, d, temperature, bev.device)
            w, b = attn.in_proj_weight, attn.in_proj_bias
            # (x + pos) Wq = x Wq + pos Wq: position term as a 36-row table (differentiable towards the weights)
            tab = torch.cat([pos @ w[:2 * d].t(), pos.new_zeros((pos.size(0), d))], 1)
            # three d-column GEMMs (the dX GEMM contracts over the output columns: at most 256 per call)
            qkv = torch.cat([linear_w(x, w[i * d:(i + 1) * d], b[i * d:(i + 1) * d]) for i in range(3)], 1)
            if S % win == 0:   # the config's grids (180, 90 with 6 x 6 windows): structured table gradient
                qkv = _WindowTableAdd.apply(qkv, tab, index, B, S, win, win // 2 if shift else win)
            else:
                qkv = qkv + tab[index.long().repeat(B)]
            att = ops.WindowAttentionFunction.apply(qkv, B, S, d, layer.win_attn.nhead, win, shift)
            y = F.layer_norm(x + linear(att, attn.out_proj), (d,), layer.norm1.weight, layer.norm1.bias, layer.norm1.eps)
            h = F.gelu(linear(y, layer.linear1))
            x = F.layer_norm(y + linear(h, layer.linear2), (d,), layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)
    # an NCHW VIEW of the token rows: the consumers (dense_train.conv_stack, to_tokens, flatten(2)) read rows again
    return x.view(B, S, S, d).permute(0, 3, 1, 2)


# ----------------------------------------------------------------------------------------------------- A13
def _mha(q_in, k_in, v_in, attn, B, Lq, Lk, nhead, dropout_p=0.0):
    """nn.MultiheadAttention arithmetic on row-major [B*L, E] tokens with the HIP attention core.  dropout_p > 0 (training):
    the module's dropout on the attention probabilities (fusion_encoder.py:458), one fresh seed per call from torch's CPU
    generator (torch.manual_seed makes a run repeatable)."""
    E = q_in.size(1)
    w, b = attn.in_proj_weight, attn.in_proj_bias
    q = linear_w(q_in, w[:E], b[:E])
    k = linear_w(k_in, w[E:2 * E], b[E:2 * E])
    v = linear_w(v_in, w[2 * E:], b[2 * E:])
    if dropout_p > 0.0 and Lk <= 512 and E == 16 * nhead:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        att = ops.AttentionFunction.apply(q, k, v, B, Lq, Lk, E, nhead, float(dropout_p), seed)
    else:
        att = ops.AttentionFunction.apply(q, k, v, B, Lq, Lk, E, nhead)
    return linear(att, attn.out_proj)


def ins_context_att(mod, x_ins, query_pos, scene, bev_size):
    """InsContextAtt.forward (fusion_encoder.py:795-830) with gradients; scene in the orientation ops.ins_context_att
    takes (already transposed)."""
    dev = scene.device
    B, E, Q = x_ins.shape
    H, W = scene.shape[2:]
    g = torch.linspace(0, bev_size - 1, bev_size, device=dev) + 0.5
    bx, by = torch.meshgrid(g, g, indexing="ij")
    bev_pos = torch.stack([bx, by], 0).view(1, 2, -1).permute(0, 2, 1)
    key_pos = ops._pos_embed(mod.key_pos_embed, bev_pos / bev_size)[0]
    src = (scene.float().flatten(2).transpose(1, 2) + key_pos[None]).reshape(B * H * W, E)
    out = x_ins.float().transpose(1, 2).reshape(B * Q, E)
    ref = (query_pos / bev_size).reshape(B * Q, 2).contiguous()
    qpe = ops._pos_embed(mod.query_pos_embed, ref.view(B, Q, 2)).reshape(B * Q, E)
    for l in mod.layers:
        ca = l.cross_attn
        nhead, npts = ca.n_heads, ca.n_points
        p, tr = getattr(l, "dropout", 0.0), l.training      # dropout1..4 of the reference layer (:604-668, p = 0.1)
        qk_in = out + qpe
        out = F.layer_norm(out + F.dropout(_mha(qk_in, qk_in, out, l.self_attn, B, Q, Q, nhead, p if tr else 0.0), p, tr), (E,),
                           l.norm2.weight, l.norm2.bias, l.norm2.eps)                                       # dropout2
        q = out + qpe
        value = linear(src, ca.value_proj)
        off, aw = linear(q, ca.sampling_offsets), linear(q, ca.attention_weights)
        t2 = ops.MSDAFunction.apply(value.view(B, H * W, E), off, aw, ref, B, Q, nhead, E // nhead, npts, H, W)
        out = F.layer_norm(out + F.dropout(linear(t2, ca.output_proj), p, tr), (E,), l.norm1.weight, l.norm1.bias,
                           l.norm1.eps)                                                                     # dropout1
        h = F.dropout(F.relu(linear(out, l.linear1)), p, tr)                                                # dropout3
        out = F.layer_norm(out + F.dropout(linear(h, l.linear2), p, tr), (E,), l.norm3.weight, l.norm3.bias,
                           l.norm3.eps)                                                                     # dropout4
    return out.view(B, Q, E).transpose(1, 2).contiguous()


# ----------------------------------------------------------------------------------------------------- A14
def instance_to_scene(mod, query, x_ins, scene_feats, bev_size):
    """Instane2SceneAtt.forward (fusion_encoder.py:480-502) with gradients"""
    B, E, H, W = query.shape
    Q = x_ins.size(2)
    xq = ops.to_tokens(query.float())
    xk = x_ins.float().transpose(1, 2).reshape(B * Q, E)
    pd = getattr(mod, "dropout", 0.0)
    att = F.dropout(_mha(xq, xk, xk, mod.multihead_attn, B, H * W, Q, mod.nhead, pd if mod.training else 0.0), pd,
                    mod.training)                                               # self.dropout (:478, :492)
    y = F.layer_norm(xq + att, (E,), mod.norm.weight, mod.norm.bias, mod.norm.eps)
    return ChannelAttentionFunction.apply(scene_feats, ops.from_tokens(y, B, H, W))
