"""GPU end-to-end test (-m gpu): ISFusionPtsPath.extract_pts_feat (isfusion.py:103-121 without the neck) -- raw sweeps
+ camera feature maps -> multi-scale BEV features -- against the composition of the CPU oracles (C restatement for the
LiDAR branch and the pillar voxelization, torch restatement for HSF / IGF / SECONDV2)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_extract_pts_feat_matches_oracle_composition(dev, oracle_mod):
    from isfusion_amd import synthetic
    from isfusion_amd.detector import ISFusionPtsPath
    from isfusion_amd.fusion_modules import seeded_state_dict
    from isfusion_amd.norm import fold_bn
    from oracle import fusion_ops as orc
    B = 2
    torch.manual_seed(0)
    net = ISFusionPtsPath().eval()
    net._lidar.randomize_weights_(0).randomize_bn_(1)
    net.fusion_encoder.load_state_dict(seeded_state_dict(net.fusion_encoder, 100))
    net.pts_backbone.load_state_dict(seeded_state_dict(net.pts_backbone, 200))
    net.pts_neck.load_state_dict(seeded_state_dict(net.pts_neck, 250))
    net.pts_bbox_head.load_state_dict(seeded_state_dict(net.pts_bbox_head, 300))
    net = net.to(dev)
    pts = [synthetic.lidar_sweeps(4321 + i, 6000) for i in range(B)]
    inp = synthetic.fusion_inputs(77, B)
    img_feats = tuple(torch.from_numpy(a).to(dev) for a in inp["img_feats"])
    kw = dict(lidar2img=torch.from_numpy(inp["lidar2img"]), img_aug_matrix=torch.from_numpy(inp["img_aug_matrix"]),
              lidar_aug_matrix=torch.from_numpy(inp["lidar_aug_matrix"]))
    metas = [dict(input_shape=inp["input_shape"]) for _ in range(B)]
    feats, hm = net.extract_pts_feat([torch.from_numpy(p).to(dev) for p in pts], img_feats, metas,
                                     return_heatmap=True, **kw)
    assert [tuple(f.shape) for f in feats] == [(B, 128, 180, 180), (B, 256, 90, 90)]

    # ---- oracle composition on the CPU
    oracle = oracle_mod
    vs, rg = net.voxel_size, net.pc_range
    coors = np.concatenate([np.concatenate([np.full((p.shape[0], 1), b, np.int32),
                                            oracle.dynamic_voxelize(p, vs, rg)], 1) for b, p in enumerate(pts)])
    vfe = net.pts_voxel_encoder
    bn1 = [t.cpu().numpy() for t in fold_bn(vfe.vfe_layers[0].norm)]
    bn2 = [t.cpu().numpy() for t in fold_bn(vfe.vfe_layers[1].norm)]
    vf, vc, _ = oracle.dynamic_vfe(np.concatenate(pts), coors, vs, rg,
                                   vfe.vfe_layers[0].linear.weight.detach().cpu().numpy(), bn1,
                                   vfe.vfe_layers[1].linear.weight.detach().cpu().numpy(), bn2)
    bev, _ = oracle.sparse_encoder_forward(net.pts_middle_encoder.plan_to_numpy(), vf, vc, B)
    pil, pco = [], []
    for b, p in enumerate(pts):
        v, c, n = oracle.hard_voxelize(p, net.pillar_size, rg, 12, 60000)
        pil.append(v)
        pco.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
    pillars, pcoors = torch.from_numpy(np.concatenate(pil)), torch.from_numpy(np.concatenate(pco))
    sd = {k: v.float().cpu() for k, v in net.fusion_encoder.state_dict().items()}
    sdb = {"bb." + k: v.float().cpu() for k, v in net.pts_backbone.state_dict().items()}
    with torch.no_grad():
        img_bev = orc.p2g_sample(pillars[..., :3], pcoors, torch.from_numpy(inp["img_feats"][1]), kw["lidar2img"],
                                 kw["img_aug_matrix"], kw["lidar_aug_matrix"], inp["input_shape"], B, 180)
        bev_feats = orc.conv_module(torch.cat([img_bev, torch.from_numpy(bev)], 1), sd, "conv_fusion")
        g0 = orc.sstv2_forward(bev_feats, sd, "grid2region_att.0")
        ret, rhm, rtop = orc.instance_fusion(bev_feats, g0, sd, B, 180, 200)
        nxt, f0 = orc.secondv2_stage(ret, sdb, "bb", "stage1")
        g1 = orc.sstv2_forward(nxt, sd, "grid2region_att.1")
        _, f1 = orc.secondv2_stage(g1, sdb, "bb", "stage2")
    assert (hm.cpu() - rhm).abs().max().item() < 1e-3
    assert torch.equal(net.fusion_encoder.last_top_idx.cpu(), rtop), "mined instance cells differ"
    # north_star tolerance on BEV features
    assert (feats[0].cpu() - f0).abs().max().item() < 1e-3
    assert (feats[1].cpu() - f1).abs().max().item() < 1e-3

    # ---- neck (stock torch ops, run on the CPU with the same weights) + detection head
    from isfusion_amd.fusion_modules import SECONDFPN
    head_out = net.forward_pts([torch.from_numpy(p).to(dev) for p in pts], img_feats, metas, **kw)[0][0]
    neck = SECONDFPN().eval()
    neck.dense_conv = "stock"
    neck.load_state_dict({k: v.cpu() for k, v in net.pts_neck.state_dict().items()})
    with torch.no_grad():
        x = neck([f0, f1])[0]
        sdh = {k: v.float().cpu() for k, v in net.pts_bbox_head.state_dict().items()}
        ho = orc.transfusion_head_forward(x, sdh, 200)
    # the 512-channel neck output has magnitude O(1); the head sums 9*512 products per heat-map logit
    assert (head_out["dense_heatmap"].cpu() - ho["dense_heatmap"]).abs().max().item() < \
        2e-3 * max(1.0, ho["dense_heatmap"].abs().max().item())
    # Proposal selection: a LiDAR BEV map is empty over large areas, so many heat-map cells carry EXACTLY equal scores
    # and the reference's argsort breaks those ties arbitrarily (the HIP kernel: ascending flat index).  Compare what
    # is defined: the sorted top-200 scores, and per-proposal outputs for the proposals both sides selected.
    s_hip = head_out["query_heatmap_score"].cpu().max(1).values.sort(descending=True).values
    s_orc = ho["query_heatmap_score"].max(1).values.sort(descending=True).values
    assert (s_hip - s_orc).abs().max().item() < 2e-3
    HW = 180 * 180
    lab_o = orc.instance_topk(ho["dense_heatmap"], 200)[2] // HW
    lab_h, cell_h = net.pts_bbox_head.query_labels.cpu(), net.pts_bbox_head.last_top_index.cpu()
    matched = 0
    for b in range(B):
        where = {(int(c), int(n)): j for j, (c, n) in enumerate(zip(lab_o[b].tolist(), ho["top_idx"][b].tolist()))}
        for i, key in enumerate(zip(lab_h[b].tolist(), cell_h[b].tolist())):
            j = where.get((int(key[0]), int(key[1])))
            if j is None:
                continue   # a tied cell the other side did not pick
            for k in ("center", "height", "dim", "rot", "vel", "heatmap"):
                err = (head_out[k][b, :, i].cpu() - ho[k][b, :, j]).abs().max().item()
                # ~1e-3 input differences pass through 4608-term convs, two attentions and LayerNorms: only the scale
                # is asserted (a wiring / orientation mistake gives O(1) .. O(100))
                assert err < 5e-2, (k, b, i, j, err)
            matched += 1
    assert matched >= 20, f"only {matched} proposals selected by both sides"


def test_engine_neck_head_handover_matches_the_module_path(dev):
    """ISFusionPtsPath.forward_pts hands the neck's levels to the head as split-format token matrices of the un-permuted
    BEV map (SECONDFPN.forward_split -> TransFusionHeadV2.forward_split: transposed-tap convolutions, permuted cell /
    token indices) instead of the reference's permuted [B, 512, X, Y] tensor; the public module forwards
    (pts_neck(...) -> pts_bbox_head(...)) keep that tensor.  Both must give the same head outputs."""
    from isfusion_amd.detector import ISFusionPtsPath
    from isfusion_amd.fusion_modules import seeded_state_dict
    net = ISFusionPtsPath().eval()
    net.pts_neck.load_state_dict(seeded_state_dict(net.pts_neck, 250))
    net.pts_bbox_head.load_state_dict(seeded_state_dict(net.pts_bbox_head, 300))
    net = net.to(dev)
    g = torch.Generator().manual_seed(11)
    B = 2
    # dense random maps: heat-map scores are all distinct by far more than the two paths' rounding difference (they add
    # the nine taps in different orders), so both must select the same 200 cells
    feats = [torch.randn((B, 128, 180, 180), generator=g).to(dev), torch.randn((B, 256, 90, 90), generator=g).to(dev)]
    with torch.no_grad():
        via_modules = net.pts_bbox_head(net.pts_neck(feats))[0][0]
        top_modules = net.pts_bbox_head.last_top_index.clone()
        via_engine = net.pts_bbox_head.forward_split(net.pts_neck.forward_split(feats))[0]
        top_engine = net.pts_bbox_head.last_top_index.clone()
    assert set(via_engine) == set(via_modules)
    scale = max(1.0, via_modules["dense_heatmap"].abs().max().item())
    assert (via_engine["dense_heatmap"] - via_modules["dense_heatmap"]).abs().max().item() < 1e-4 * scale
    assert torch.equal(top_engine, top_modules)
    for key in ("center", "height", "dim", "rot", "vel", "heatmap", "query_heatmap_score"):
        a, w = via_engine[key], via_modules[key]
        assert a.shape == w.shape and (a - w).abs().max().item() < 1e-3 * max(1.0, w.abs().max().item()), key


def test_hip_graph_tail_reproduces_the_eager_forward(dev):
    """ISFusionPtsPath.enable_graph(): conv_fusion .. head captured once per batch size and replayed as one HIP graph,
    the LiDAR branch / pillar voxelization / Point-to-Grid eager and writing into the graph's input buffers.  Same
    kernels on the same data => the head outputs are bit-identical to the eager forward_pts, for two different frame
    sets in a row (the replay must see the new inputs) and after an eager call in between."""
    from isfusion_amd import synthetic
    from isfusion_amd.detector import ISFusionPtsPath
    from isfusion_amd.fusion_modules import seeded_state_dict
    B = 2
    net = ISFusionPtsPath().eval()
    net._lidar.randomize_weights_(0).randomize_bn_(1)
    for mod, seed in ((net.fusion_encoder, 100), (net.pts_backbone, 200), (net.pts_neck, 250), (net.pts_bbox_head, 300)):
        mod.load_state_dict(seeded_state_dict(mod, seed))
    net = net.to(dev)
    sets = []
    for fs in range(2):
        pts = [torch.from_numpy(synthetic.lidar_sweeps(7000 + 10 * fs + i, 20000)).to(dev) for i in range(B)]
        inp = synthetic.fusion_inputs(80 + fs, B)
        img_feats = tuple(torch.from_numpy(a).to(dev) for a in inp["img_feats"])
        kw = dict(lidar2img=torch.from_numpy(inp["lidar2img"]), img_aug_matrix=torch.from_numpy(inp["img_aug_matrix"]),
                  lidar_aug_matrix=torch.from_numpy(inp["lidar_aug_matrix"]))
        sets.append((pts, img_feats, [dict(input_shape=inp["input_shape"]) for _ in range(B)], kw))
    keys = ("center", "height", "dim", "rot", "vel", "heatmap", "query_heatmap_score", "dense_heatmap")

    def run(i):
        pts, img_feats, metas, kw = sets[i]
        out = net.forward_pts(pts, img_feats, metas, **kw)[0][0]
        return {k: out[k].clone() for k in keys}, net.pts_bbox_head.last_top_index.clone()

    eager = [run(0), run(1)]
    assert not torch.equal(eager[0][0]["dense_heatmap"], eager[1][0]["dense_heatmap"])
    with pytest.raises(Exception):
        net.enable_graph()
        run(0)                       # weights not declared final: refused
    net.freeze().enable_graph()
    for i in (0, 1, 0, 1):
        got, top = run(i)
        assert torch.equal(top, eager[i][1])
        for k in keys:
            assert torch.equal(got[k], eager[i][0][k]), (i, k)
    net.enable_graph(False)
    got, _ = run(1)                  # eager again (frozen caches)
    assert all(torch.equal(got[k], eager[1][0][k]) for k in keys)


def test_back_to_back_forwards_without_host_sync(dev):
    """The bench loop queues forwards without ever synchronising; the host then runs ahead of the GPU (the LiDAR branch
    of forward n + 1 is being launched while the tail of forward n executes, by a whole tail in graph mode).  Twelve
    full-size forwards (B = 2 x 300 k points, alternating frame sets) queued back to back must reproduce the
    synchronised results bit for bit -- eager and as HIP-graph replays.  Regression (tools/graph_fault.py,
    profiles/r04_graph_fault.txt): with the legacy NULL stream as the launch stream the replays ended in a GPU memory
    fault at the fourth unsynchronised forward when the pillar voxelization ran on a side stream; graph mode now moves a
    caller on the NULL stream onto a private launch stream (side-stream voxelization restored), and a caller on its own
    stream runs where it is."""
    from isfusion_amd import synthetic
    from isfusion_amd.detector import ISFusionPtsPath
    from isfusion_amd.fusion_modules import seeded_state_dict
    B = 2
    net = ISFusionPtsPath().eval()
    net._lidar.randomize_weights_(0).randomize_bn_(1)
    for mod, seed in ((net.fusion_encoder, 100), (net.pts_backbone, 200), (net.pts_neck, 250), (net.pts_bbox_head, 300)):
        mod.load_state_dict(seeded_state_dict(mod, seed))
    net = net.to(dev).freeze()
    sets = []
    for fs in range(2):
        pts = [torch.from_numpy(synthetic.lidar_sweeps(7100 + 10 * fs + i, 300000)).to(dev) for i in range(B)]
        inp = synthetic.fusion_inputs(90 + fs, B)
        img_feats = tuple(torch.from_numpy(a).to(dev) for a in inp["img_feats"])
        kw = dict(lidar2img=torch.from_numpy(inp["lidar2img"]), img_aug_matrix=torch.from_numpy(inp["img_aug_matrix"]),
                  lidar_aug_matrix=torch.from_numpy(inp["lidar_aug_matrix"]))
        sets.append((pts, img_feats, [dict(input_shape=inp["input_shape"]) for _ in range(B)], kw))
    keys = ("center", "height", "dim", "rot", "vel", "heatmap", "dense_heatmap")

    def run(i):
        pts, img_feats, metas, kw = sets[i % 2]
        out = net.forward_pts(pts, img_feats, metas, **kw)[0][0]
        return {k: out[k].clone() for k in keys}

    want = []
    for i in range(2):
        want.append(run(i))
        torch.cuda.synchronize()
    own = torch.cuda.Stream(device=dev)
    for graph, stream in ((False, None), (True, None), (True, own)):
        net.enable_graph(graph)
        if stream is not None:
            stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream if stream is not None else torch.cuda.current_stream()):
            got = [run(i) for i in range(12)]          # no host sync in between
        torch.cuda.synchronize()
        for i, g in enumerate(got):
            for k in keys:
                assert torch.equal(g[k], want[i % 2][k]), (graph, stream is not None, i, k)


def test_soak_unsynchronised_forwards_on_several_non_null_streams(dev):
    """the graph-mode fault was narrowed to the legacy NULL stream, not root-caused; before trusting the
    stream hop, soak it.  450 full-path forwards (B = 2 x 60 k points, two alternating frame sets) queued WITHOUT any host
    synchronisation on three streams of torch's pool in turn -- 150 eager, 150 as HIP-graph replays with the pillar
    voxelization on the launch stream (the round-5 default), 150 as replays with it on the side stream (the set-up that
    faulted on the NULL stream) -- every one of them bit-identical to the synchronised reference."""
    from isfusion_amd import synthetic
    from isfusion_amd.detector import ISFusionPtsPath
    from isfusion_amd.fusion_modules import seeded_state_dict
    B = 2
    net = ISFusionPtsPath().eval()
    net._lidar.randomize_weights_(0).randomize_bn_(1)
    for mod, seed in ((net.fusion_encoder, 100), (net.pts_backbone, 200), (net.pts_neck, 250), (net.pts_bbox_head, 300)):
        mod.load_state_dict(seeded_state_dict(mod, seed))
    net = net.to(dev).freeze()
    sets = []
    for fs in range(2):
        pts = [torch.from_numpy(synthetic.lidar_sweeps(8100 + 10 * fs + i, 60000)).to(dev) for i in range(B)]
        inp = synthetic.fusion_inputs(190 + fs, B)
        img_feats = tuple(torch.from_numpy(a).to(dev) for a in inp["img_feats"])
        kw = dict(lidar2img=torch.from_numpy(inp["lidar2img"]), img_aug_matrix=torch.from_numpy(inp["img_aug_matrix"]),
                  lidar_aug_matrix=torch.from_numpy(inp["lidar_aug_matrix"]))
        sets.append((pts, img_feats, [dict(input_shape=inp["input_shape"]) for _ in range(B)], kw))
    keys = ("center", "heatmap", "dense_heatmap")

    def run(i):
        pts, img_feats, metas, kw = sets[i % 2]
        out = net.forward_pts(pts, img_feats, metas, **kw)[0][0]
        return torch.stack([out[k].double().abs().sum() for k in keys])      # a checksum per forward (device side)

    want = []
    for i in range(2):
        want.append(run(i).clone())
        torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    for graph, side in ((False, False), (True, False), (True, True)):
        net.enable_graph(graph, pillar_side_stream=side)
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        sums = []
        for k in range(150):
            st = streams[(k // 5) % 3]                      # five in a row per stream, then the next stream
            with torch.cuda.stream(st):
                sums.append(run(k))
            if k % 5 == 4:                                  # order the streams behind each other, on the device only
                streams[((k // 5) + 1) % 3].wait_stream(st)
        torch.cuda.synchronize()
        for k, c in enumerate(sums):
            assert torch.equal(c, want[k % 2]), (graph, side, k)
    net.enable_graph(False)


def test_lidar_branch_batches_in_flight_on_two_streams_reproduce_serial_bits(dev):
    """bench.py's "pipelined" leg: consecutive LiDAR-branch calls alternate between two HIP streams (workspaces, count
    mailboxes and geometry side streams are per (device, stream)), so that one batch's voxelization / VFE / geometry runs
    beside the previous batch's convolutions.  Nothing is synchronised between the calls; every output equals the serial
    call's bit for bit, for batches of different sizes following each other on the same stream (workspace reuse)."""
    import isfusion_amd as m
    from isfusion_amd import synthetic
    lb = m.LidarBranch().randomize_weights_(0).randomize_bn_(1).eval().to(dev).freeze()
    sets = [[torch.from_numpy(synthetic.lidar_sweeps(700 + 10 * s + i, n)).to(dev) for i in range(b)]
            for s, (n, b) in enumerate(((60000, 2), (20000, 3), (120000, 2), (3000, 1)))]
    want = [lb(p).clone() for p in sets]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
    got = []
    for k in range(12):
        with torch.cuda.stream(streams[k % 2]):
            got.append((k % len(sets), lb(sets[k % len(sets)])))
        if k % 3 == 2:                      # three in a row on alternating streams, then a different pairing
            streams.reverse()
    torch.cuda.synchronize()
    for k, (s, out) in enumerate(got):
        assert torch.equal(out, want[s]), (k, s)


def test_lib_stream_is_torchs_current_stream(dev):
    """_lib.stream() (the raw-handle fast path) == torch.cuda.current_stream().cuda_stream on the default stream, inside a
    torch.cuda.stream() context, and after it"""
    from isfusion_amd import _lib
    assert _lib.stream() == torch.cuda.current_stream().cuda_stream
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        assert _lib.stream() == side.cuda_stream == torch.cuda.current_stream().cuda_stream
    assert _lib.stream() == torch.cuda.current_stream().cuda_stream
