"""CPU checks of the index algebra / gradient formulas the backward kernels implement, written when
no GPU was available: each test re-states, lane by lane in numpy, what a kernel in is-fusion_amd/csrc does and compares
with an independent computation.  They pin the maths and the MFMA operand / accumulator lane mapping -- the plumbing
(launch geometry, pointers) is what tests/test_gpu_next.py checks on hardware."""
import numpy as np
import torch


def mfma_16x16x4(a_lane, b_lane, acc):
    """v_mfma_f32_16x16x4_f32 for one wave: a_lane[l] = A[m = l & 15][k = l >> 4], b_lane[l] = B[k = l >> 4][n = l & 15],
    acc[l][r] = C[m = 4 * (l >> 4) + r][n = l & 15] (the layout isf_spconv.hip's forward kernel is validated with)."""
    A = np.zeros((16, 4), np.float64)
    Bm = np.zeros((4, 16), np.float64)
    for l in range(64):
        A[l & 15, l >> 4] = a_lane[l]
        Bm[l >> 4, l & 15] = b_lane[l]
    C = A @ Bm
    for l in range(64):
        for r in range(4):
            acc[l, r] += C[4 * (l >> 4) + r, l & 15]


def test_wgrad_mfma_lane_mapping():
    """wgrad_mfma_kernel (isf_spconv_bwd.hip): 64 x 64 block of dW[k] from 16-byte loads of x and dy"""
    rng = np.random.default_rng(0)
    cin, cout, n_in, n_out = 96, 80, 50, 37           # not multiples of 64: exercises the channel masks
    x = rng.normal(size=(n_in, cin))
    dy = rng.normal(size=(n_out, cout))
    nbr = rng.integers(-1, n_in, n_out)               # one tap
    want = np.zeros((cin, cout))
    for o in range(n_out):
        if nbr[o] >= 0:
            want += np.outer(x[nbr[o]], dy[o])
    got = np.full((cin, cout), np.nan)
    co_blocks = (cout + 63) // 64
    for blk in range(((cin + 63) // 64) * co_blocks):
        ci_base, co_base = (blk // co_blocks) * 64, (blk % co_blocks) * 64
        acc = np.zeros((4, 4, 64, 4))
        for r0 in range(0, n_out, 4):
            a = np.zeros((64, 4))
            b = np.zeros((64, 4))
            for lane in range(64):
                sub, kslot = lane & 15, lane >> 4
                row = r0 + kslot
                src = nbr[row] if row < n_out else -1
                if src >= 0:
                    if ci_base + 4 * sub < cin:
                        a[lane] = x[src, ci_base + 4 * sub: ci_base + 4 * sub + 4]
                    if co_base + 4 * sub < cout:
                        b[lane] = dy[row, co_base + 4 * sub: co_base + 4 * sub + 4]
            for s in range(4):
                for t in range(4):
                    mfma_16x16x4(a[:, s], b[:, t], acc[s, t])
        for lane in range(64):
            sub, kslot = lane & 15, lane >> 4
            for s in range(4):
                for r in range(4):
                    ci = ci_base + 4 * (4 * kslot + r) + s
                    co = co_base + 4 * sub
                    if ci < cin and co < cout:
                        got[ci, co:co + 4] = [acc[s, t, lane, r] for t in range(4)]
    assert not np.isnan(got).any(), "some dW element is never written"
    assert np.abs(got - want).max() < 1e-9


def test_transposed_rulebook_gives_input_gradient():
    """dX through the forward formulation over nbr_t (isf_transpose_rulebook + isf_sparse_conv_backward_input)"""
    rng = np.random.default_rng(1)
    K, n_in, n_out, cin, cout = 5, 30, 22, 6, 4
    nbr = np.full((K, n_out), -1)
    for k in range(K):                                 # a tap maps an input row to at most one output row
        outs = rng.choice(n_out, 12, replace=False)
        ins = rng.choice(n_in, 12, replace=False)
        nbr[k, outs] = ins
    w = rng.normal(size=(K, cin, cout))
    dy = rng.normal(size=(n_out, cout))
    want = np.zeros((n_in, cin))
    for k in range(K):
        for o in range(n_out):
            if nbr[k, o] >= 0:
                want[nbr[k, o]] += w[k] @ dy[o]
    nbr_t = np.full((K, n_in), -1)
    for k in range(K):
        for o in range(n_out):
            if nbr[k, o] >= 0:
                assert nbr_t[k, nbr[k, o]] == -1
                nbr_t[k, nbr[k, o]] = o
    wt = w.transpose(0, 2, 1)                          # [K, cout, cin]
    got = np.zeros((n_in, cin))
    for j in range(n_in):                              # y[j] = sum_k x[nbr_t[k][j]] @ Wt[k]: the forward kernel
        for k in range(K):
            if nbr_t[k, j] >= 0:
                got[j] += dy[nbr_t[k, j]] @ wt[k]
    assert np.abs(got - want).max() < 1e-12


def test_attention_backward_row_and_column_formulas():
    """attention_bwd_rows_kernel / attention_bwd_cols_kernel: probabilities recomputed from L = logsumexp and
    D = dO . O; every dQ / dK / dV row from one pass -- vs torch autograd"""
    g = torch.Generator().manual_seed(0)
    Lq, Lk, hd = 13, 9, 16
    q, k, v, go = [torch.randn(s, generator=g, dtype=torch.float64) for s in ((Lq, hd), (Lk, hd), (Lk, hd), (Lq, hd))]
    scale = 1.0 / hd ** 0.5
    qa, ka, va = q.clone().requires_grad_(), k.clone().requires_grad_(), v.clone().requires_grad_()
    out = ((qa @ ka.T) * scale).softmax(-1) @ va
    out.backward(go)
    s = (q @ k.T) * scale
    L = torch.logsumexp(s, 1)
    D = (go * out.detach()).sum(1)
    dq = torch.zeros_like(q)
    for i in range(Lq):                                # rows kernel: wave i, lanes over keys
        for j in range(Lk):
            p = torch.exp(s[i, j] - L[i])
            dq[i] += p * (go[i] @ v[j] - D[i]) * k[j]
    dq *= scale
    dk, dv = torch.zeros_like(k), torch.zeros_like(v)
    for j in range(Lk):                                # cols kernel: wave j, lanes over queries
        for i in range(Lq):
            p = torch.exp(s[i, j] - L[i])
            dv[j] += p * go[i]
            dk[j] += p * (go[i] @ v[j] - D[i]) * q[i]
    dk *= scale
    assert (dq - qa.grad).abs().max() < 1e-12 and (dk - ka.grad).abs().max() < 1e-12
    assert (dv - va.grad).abs().max() < 1e-12


def test_msda_backward_formulas():
    """msda_backward_kernel: d(value) bilinear scatter, d(offset) = a * dO . ds/d(pixel), d(logit) through the softmax
    -- one (query, head) in numpy vs torch autograd through the same sampling rule"""
    g = torch.Generator().manual_seed(3)
    H = W = 7
    D, P = 16, 16
    value = torch.randn((H, W, D), generator=g, dtype=torch.float64)
    off = torch.randn((P, 2), generator=g, dtype=torch.float64) * 2
    logits = torch.randn((P,), generator=g, dtype=torch.float64)
    ref = torch.tensor([0.45, 0.6], dtype=torch.float64)
    go = torch.randn((D,), generator=g, dtype=torch.float64)

    def forward(value, off, logits):
        a = logits.softmax(0)
        out = torch.zeros(D, dtype=torch.float64)
        for p in range(P):
            w_im = (ref[0] + off[p, 0] / W) * W - 0.5
            h_im = (ref[1] + off[p, 1] / H) * H - 0.5
            if not (h_im > -1 and w_im > -1 and h_im < H and w_im < W):
                continue
            h0, w0 = int(torch.floor(h_im)), int(torch.floor(w_im))
            lh, lw = h_im - h0, w_im - w0
            s = torch.zeros(D, dtype=torch.float64)
            for dh, dw, wt in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw), (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
                if 0 <= h0 + dh <= H - 1 and 0 <= w0 + dw <= W - 1:
                    s = s + wt * value[h0 + dh, w0 + dw]
            out = out + a[p] * s
        return out

    va, oa, la = value.clone().requires_grad_(), off.clone().requires_grad_(), logits.clone().requires_grad_()
    forward(va, oa, la).backward(go)
    # the kernel's formulas
    a = logits.softmax(0)
    gv = torch.zeros_like(value)
    goff = torch.zeros_like(off)
    ga = torch.zeros(P, dtype=torch.float64)
    for p in range(P):
        w_im = float((ref[0] + off[p, 0] / W) * W - 0.5)
        h_im = float((ref[1] + off[p, 1] / H) * H - 0.5)
        if not (h_im > -1 and w_im > -1 and h_im < H and w_im < W):
            continue
        h0, w0 = int(np.floor(h_im)), int(np.floor(w_im))
        lh, lw = h_im - h0, w_im - w0
        vv = {}
        for dh, dw, wt in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw), (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
            ok = 0 <= h0 + dh <= H - 1 and 0 <= w0 + dw <= W - 1
            vv[dh, dw] = value[h0 + dh, w0 + dw] if ok else torch.zeros(D, dtype=torch.float64)
            if ok:
                gv[h0 + dh, w0 + dw] += wt * a[p] * go
        s = (1 - lh) * (1 - lw) * vv[0, 0] + (1 - lh) * lw * vv[0, 1] + lh * (1 - lw) * vv[1, 0] + lh * lw * vv[1, 1]
        dsw = (1 - lh) * (vv[0, 1] - vv[0, 0]) + lh * (vv[1, 1] - vv[1, 0])
        dsh = (1 - lw) * (vv[1, 0] - vv[0, 0]) + lw * (vv[1, 1] - vv[0, 1])
        ga[p] = go @ s
        goff[p, 0] = a[p] * (go @ dsw)
        goff[p, 1] = a[p] * (go @ dsh)
    glog = a * (ga - (a * ga).sum())
    assert (gv - va.grad).abs().max() < 1e-12
    assert (goff - oa.grad).abs().max() < 1e-12
    assert (glog - la.grad).abs().max() < 1e-12


def test_decode_boxes_kernel_algebra_matches_reference_golden(golden):
    """decode_boxes_kernel (isf_decode.hip) lane by lane: first-maximum class score, box arithmetic, centre-range /
    threshold mask, ballot + popcount compaction in proposal order -- vs the reference's get_bboxes goldens"""
    from fusion_common import HEAD_CODERS, HEAD_CONFIGS
    g = golden("head_ref.npz")
    for name, cfg in HEAD_CONFIGS.items():
        P = cfg["num_proposals"]
        hm, qs = g[name + ".heatmap"], g[name + ".query_heatmap_score"]
        lab = g[name + ".labels"]
        for cname, c in HEAD_CODERS.items():
            cell = np.float32(c["out_size_factor"] * c["voxel_size"][0])
            lo, hi = np.float32(c["post_center_range"][:3]), np.float32(c["post_center_range"][3:])
            for b in range(cfg["B"]):
                boxes, scores, labels, kept = np.zeros((P, 9), np.float32), np.zeros(P, np.float32), np.zeros(P, int), 0
                for p0 in range(0, P, 64):
                    keep = np.zeros(64, bool)
                    lane_box, lane_best, lane_arg = {}, {}, {}
                    for lane in range(64):
                        p = p0 + lane
                        if p >= P:
                            continue
                        best, arg = np.float32(0), 0
                        for cc in range(hm.shape[1]):
                            s = np.float32(1) / (np.float32(1) + np.exp(-hm[b, cc, p])) * qs[b, cc, p] * \
                                np.float32(1.0 if cc == lab[b, p] else 0.0)
                            if cc == 0 or s > best:
                                best, arg = s, cc
                        bx = np.zeros(9, np.float32)
                        bx[0] = g[name + ".center"][b, 0, p] * cell + np.float32(c["pc_range"][0])
                        bx[1] = g[name + ".center"][b, 1, p] * cell + np.float32(c["pc_range"][1])
                        bx[3:6] = np.exp(g[name + ".dim"][b, :, p])
                        bx[2] = g[name + ".height"][b, 0, p] - bx[5] * np.float32(0.5)
                        bx[6] = np.arctan2(g[name + ".rot"][b, 0, p], g[name + ".rot"][b, 1, p])
                        bx[7:9] = g[name + ".vel"][b, :, p]
                        k = bool((bx[:3] >= lo).all() and (bx[:3] <= hi).all())
                        if c["score_threshold"]:
                            k = k and best > np.float32(c["score_threshold"])
                        keep[lane], lane_box[lane], lane_best[lane], lane_arg[lane] = k, bx, best, arg
                    for lane in range(64):
                        if keep[lane]:
                            at = kept + int(keep[:lane].sum())      # popcount of the ballot below this lane
                            boxes[at], scores[at], labels[at] = lane_box[lane], lane_best[lane], lane_arg[lane]
                    kept += int(keep.sum())
                ref = g[f"{name}.{cname}.{b}.boxes"]
                assert kept == ref.shape[0], (name, cname, b)
                assert np.array_equal(labels[:kept], g[f"{name}.{cname}.{b}.box_labels"])
                assert np.abs(scores[:kept] - g[f"{name}.{cname}.{b}.scores"]).max(initial=0) < 1e-6
                assert np.abs(boxes[:kept] - ref).max(initial=0) < 2e-5


def test_window_attention_backward_formulas_with_partial_windows():
    """window_attention_bwd_kernel: query role (L, D, dQ over the window's valid keys) then key role (dK, dV over its
    valid queries) for one window that hangs over the grid edge -- vs torch autograd"""
    g = torch.Generator().manual_seed(5)
    T, hd = 36, 16
    valid = torch.zeros(T, dtype=torch.bool)
    valid[[i for i in range(T) if (i // 6) >= 3 and (i % 6) >= 2]] = True       # shifted corner window: 3 x 4 cells
    q, k, v, go = [torch.randn((T, hd), generator=g, dtype=torch.float64) for _ in range(4)]
    scale = 1.0 / hd ** 0.5
    idx = valid.nonzero().flatten()
    qa, ka, va = [t[idx].clone().requires_grad_() for t in (q, k, v)]
    (((qa @ ka.T) * scale).softmax(-1) @ va).backward(go[idx])
    L, D = torch.zeros(T, dtype=torch.float64), torch.zeros(T, dtype=torch.float64)
    dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    for i in range(T):                                   # query role
        if not valid[i]:
            continue
        s = torch.tensor([float(q[i] @ k[j]) * scale if valid[j] else -np.inf for j in range(T)], dtype=torch.float64)
        m = s.max()
        L[i] = m + torch.log(torch.exp(s - m).sum())
        for j in range(T):
            D[i] += torch.exp(s[j] - L[i]) * (go[i] @ v[j])
        for j in range(T):
            dq[i] += torch.exp(s[j] - L[i]) * ((go[i] @ v[j]) - D[i]) * k[j]
    for j in range(T):                                   # key role
        if not valid[j]:
            continue
        for i in range(T):
            if not valid[i]:
                continue
            p = torch.exp((q[i] @ k[j]) * scale - L[i])
            dv[j] += p * go[i]
            dk[j] += p * ((go[i] @ v[j]) - D[i]) * q[i]
    assert (dq[idx] * scale - qa.grad).abs().max() < 1e-12
    assert (dk[idx] * scale - ka.grad).abs().max() < 1e-12
    assert (dv[idx] - va.grad).abs().max() < 1e-12


def test_assemble_points_block_partition_and_compaction(golden):
    """isf_assemble_points (isf_input.hip): per-file block prefix, per-block binary search for the owning file, keep
    flags, exclusive scan of block counts, packed write, per-sample offsets -- emulated block by block vs the
    reference pipeline's goldens (a batch of three samples, one without sweeps, plus an empty file)"""
    from input_common import INPUT_CONFIGS, PC_RANGE, sweep_inputs
    g = golden("input_ref.npz")
    BLK = 256
    names = ["a", "key_only", "b"]
    files, raw = [], []                                   # descriptors in the loader's order
    row = 0

    def add(arr, sample, is_sweep, lag=0.0, rot=None, trans=None):
        nonlocal row
        files.append(dict(first=row, n=arr.shape[0], sample=sample, is_sweep=is_sweep, lag=np.float32(lag),
                          rot=np.eye(3) if rot is None else rot, trans=np.zeros(3) if trans is None else trans))
        raw.append(arr)
        row += arr.shape[0]

    for b, n in enumerate(names):
        key, sweeps, ts = sweep_inputs(*INPUT_CONFIGS[n])
        add(key, b, False)
        if n == "a":
            add(np.zeros((0, 5), np.float32), b, True)    # an empty file in the middle
        for sw in sweeps:
            add(sw["points"], b, True, ts - sw["timestamp"] / 1e6, sw["sensor2lidar_rotation"],
                sw["sensor2lidar_translation"])
    raw = np.concatenate(raw)
    B = len(names)
    # host side of isf_assemble_points
    blocks, prev, first_block = 0, 0, [0] * (B + 1)
    for f in files:
        for bb in range(prev + 1, f["sample"] + 1):
            first_block[bb] = blocks
        prev = f["sample"]
        f["block_begin"] = blocks
        blocks += (f["n"] + BLK - 1) // BLK
    for bb in range(prev + 1, B + 1):
        first_block[bb] = blocks
    live = len(files)
    while live > 0 and files[live - 1]["n"] == 0:
        live -= 1
    rng = np.float32(PC_RANGE)

    def block(blk):
        lo, hi = 0, live - 1                              # last file with block_begin <= blk
        while lo < hi:
            mid = (lo + hi + 1) >> 1
            if files[mid]["block_begin"] <= blk:
                lo = mid
            else:
                hi = mid - 1
        f = files[lo]
        start = (blk - f["block_begin"]) * BLK
        n = min(BLK, f["n"] - start)
        assert n > 0, "a block must own points"
        p = raw[f["first"] + start: f["first"] + start + n].copy()
        if f["is_sweep"]:
            xyz = (p[:, :3].astype(np.float64) @ f["rot"].T).astype(np.float32)
            p[:, :3] = (xyz.astype(np.float64) + f["trans"]).astype(np.float32)
            p[:, 4] = f["lag"]
        else:
            p[:, 4] = 0
        keep = ((p[:, 0] > rng[0]) & (p[:, 1] > rng[1]) & (p[:, 2] > rng[2]) &
                (p[:, 0] < rng[3]) & (p[:, 1] < rng[4]) & (p[:, 2] < rng[5]))
        return p, keep

    counts = np.array([block(blk)[1].sum() for blk in range(blocks)])
    offsets = np.concatenate([[0], np.cumsum(counts)[:-1]])
    out = np.zeros((raw.shape[0], 5), np.float32)
    for blk in range(blocks):
        p, keep = block(blk)
        rank = np.cumsum(keep) - keep                     # ballot + popcount prefix, wave counts added in order
        out[offsets[blk] + rank[keep]] = p[keep]
    sample_offsets = [int(offsets[fb]) if fb < blocks else int(offsets[-1] + counts[-1]) for fb in first_block]
    for b, n in enumerate(names):
        ref = g[f"{n}.test.points"]
        got = out[sample_offsets[b]:sample_offsets[b + 1]]
        assert got.shape == ref.shape and np.array_equal(got, ref), n


def test_wgrad16_lds_image_is_a_bijection_and_conflict_free():
    """isf_spconv_wgrad16.hip stages a wave's 32-pair x BC-channel operand tile channel-major in LDS (wg16_addr: channel
    record c ^ ((c >> 3) & 1), row-octet slot kg ^ g((c >> 2) & 3) ^ h((c >> 4) & 3)).  Restated here lane by lane: every
    (row, channel, half) lands on its own bytes, an MFMA fragment read (lane = (m, kg): 8 consecutive rows of one
    channel) finds them, the four 16-lane groups a ds_read_b128 is served in hit 16 different 16-byte slots of a 256-byte
    bank row and the 16 contiguous lanes a ds_write_b64 is served in 16 different 8-byte slots of 128 bytes
    (MI355X_MICROARCH.md, LDS table)."""
    def g(x): return (-x) & 3
    def h(x): return ((x & 1) << 1) | ((x >> 1) & 1)
    def addr(c, kg): return ((c ^ ((c >> 3) & 1)) << 6) + (((kg ^ g((c >> 2) & 3) ^ h((c >> 4) & 3)) & 3) << 4)
    read_groups = [[0, 1, 2, 3, 12, 13, 14, 15] + list(range(20, 28)), list(range(4, 12)) + [16, 17, 18, 19, 28, 29, 30, 31]]
    read_groups += [[l + 32 for l in grp] for grp in read_groups]
    for BC in (32, 64):
        mem = {}
        lane_of = lambda lane: ((lane & 7, (lane >> 3) & 7) if BC == 64 else (lane & 3, (lane >> 2) & 7))
        for lane in range(64):
            cg, q = lane_of(lane)
            for it in range(BC // 32):
                hl = it if BC == 64 else lane >> 5
                for c in range(8 * cg, 8 * cg + 8):
                    a = hl * BC * 64 + addr(c, q >> 1) + (q & 1) * 8
                    for j in range(4):
                        assert a + 2 * j not in mem
                        mem[a + 2 * j] = (4 * q + j, c, hl)
        assert len(mem) == BC * 64
        # the kernel's store path (Wg16Lane): two base addresses per item + immediate record offsets + an even / odd
        # data swap for odd channel octets must hit exactly wg16_addr's bytes
        for lane in range(64):
            cg, q = lane_of(lane)
            swap = cg & 1
            lds0 = (addr(8 * cg, q >> 1) & ~0x1c0)
            lds1 = (addr(8 * cg + 4, q >> 1) & ~0x1c0) + 256
            for d in range(4):
                a = lds0 if d < 2 else lds1
                for rec, ch in (((2 * d) & 3, 2 * d + (1 if swap else 0)), ((2 * d + 1) & 3, 2 * d + (0 if swap else 1))):
                    assert a + 64 * rec == addr(8 * cg + ch, q >> 1), (BC, lane, d)
        for hl in range(2):
            for mt in range(BC // 16):
                for lane in range(64):
                    a = hl * BC * 64 + addr(mt * 16 + (lane & 15), lane >> 4)
                    assert [mem[a + 2 * jj] for jj in range(8)] == [(8 * (lane >> 4) + jj, mt * 16 + (lane & 15), hl)
                                                                    for jj in range(8)]
                for grp in read_groups:
                    assert len({(addr(mt * 16 + (l & 15), l >> 4) // 16) % 16 for l in grp}) == 16
        for it in range(BC // 32):
            for e in range(8):
                for g0 in range(0, 64, 16):
                    slots = set()
                    for lane in range(g0, g0 + 16):
                        cg, q = lane_of(lane)
                        hl = it if BC == 64 else lane >> 5
                        slots.add(((hl * BC * 64 + addr(8 * cg + e, q >> 1) + (q & 1) * 8) // 8) % 16)
                    assert len(slots) == 16


def test_window_table_gradient_fold_equals_autograd():
    """fusion_train._WindowTableAdd: the gradient of the 36-row window-position table as a strided fold of the token
    gradient (instead of autograd's sort-based index backward over all tokens) -- both shifts, several grid sizes"""
    from isfusion_amd import fusion_ops as ops
    from isfusion_amd import fusion_train as ft
    torch.manual_seed(0)
    for S, win, shift in ((12, 6, 0), (12, 6, 1), (18, 6, 1), (30, 6, 0)):
        B, d = 2, 8
        index, _ = ops._window_tables(S, win, shift, d, 1000.0, "cpu")
        off = win // 2 if shift else win
        tab = torch.randn(win * win, 3 * d, dtype=torch.float64, requires_grad=True)
        qkv = torch.randn(B * S * S, 3 * d, dtype=torch.float64, requires_grad=True)
        w = torch.randn(B * S * S, 3 * d, dtype=torch.float64)
        a = qkv + tab[index.long().repeat(B)]
        (a * w).sum().backward()
        g1, q1 = tab.grad.clone(), qkv.grad.clone()
        tab.grad = qkv.grad = None
        b = ft._WindowTableAdd.apply(qkv, tab, index, B, S, win, off)
        (b * w).sum().backward()
        assert torch.allclose(a, b) and torch.allclose(tab.grad, g1) and torch.allclose(qkv.grad, q1), (S, win, shift)


def test_rows_linear_and_bn_fallback_gradients_on_the_cpu():
    """voxel_encoder._RowsLinear (chunked weight gradient of the DynamicVFE linears) against F.linear's autograd, and
    norm.bn1d_relu's stock fallback (CPU tensors / eval mode: nn.BatchNorm1d -> add -> relu, op for op the reference's
    composition) against the same composition written out"""
    from isfusion_amd import norm
    from isfusion_amd.voxel_encoder import _RowsLinear
    torch.manual_seed(0)
    x = torch.randn(20000, 11, dtype=torch.float64, requires_grad=True)
    w = torch.randn(64, 11, dtype=torch.float64, requires_grad=True)
    g = torch.randn(20000, 64, dtype=torch.float64)
    y = _RowsLinear.apply(x, w)
    y.backward(g)
    got = (y.detach(), x.grad.clone(), w.grad.clone())
    x.grad = w.grad = None
    y2 = torch.nn.functional.linear(x, w)
    y2.backward(g)
    for a, b in zip(got, (y2.detach(), x.grad, w.grad)):
        assert torch.allclose(a, b, rtol=1e-12, atol=1e-12)
    for train in (True, False):
        bn = torch.nn.BatchNorm1d(8).train(train)
        bn2 = torch.nn.BatchNorm1d(8).train(train)
        bn2.load_state_dict(bn.state_dict())
        xa = torch.randn(50, 8, requires_grad=True)
        xb = xa.detach().clone().requires_grad_()
        res = torch.randn(50, 8)
        out = norm.bn1d_relu(bn, xa, residual=res, relu=True)          # CPU tensors: the stock composition
        ref = torch.relu(bn2(xb) + res)
        out.sum().backward()
        ref.sum().backward()
        assert torch.allclose(out, ref) and torch.allclose(xa.grad, xb.grad)
        assert torch.allclose(bn.running_mean, bn2.running_mean) and torch.allclose(bn.running_var, bn2.running_var)
