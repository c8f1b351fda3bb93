"""bench.py -- IS-Fusion LiDAR-branch forward throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[1] -- "isfusion_0075voxel LiDAR-only branch
(voxelize + spconv backbone -> BEV), synthetic 300k-pt sweeps, batch=4" per GPU.  One "step" = one forward
of the whole LiDAR branch (dynamic voxelization -> DynamicVFE -> 21-layer SparseEncoder -> dense BEV
[4,512,180,180]) over one batch whose points are already resident in HBM.  Weights: random init of the
isfusion_0075voxel architecture (no checkpoint / dataset access), eval-mode BN, fp32.

Multi-GPU: frames are independent units => every rank runs its own batch of 4 frames, no data-path
collective (scaling "weak"); timing = barrier + synchronize on both sides, max over ranks.

Prints ONE JSON line on rank 0 with the contract fields plus
  "roofline":     dominant kernel (sparse-conv template instantiation with the largest share of GPU time),
                  achieved = algorithmic flops (2*pairs*Cin*Cout) / mean hipEvent kernel time, vs what the matrix
                  pipe delivers per algorithmic flop (dense f16 peak / 3 passes of the split arithmetic, or the fp32
                  MFMA peak with --fp32; MI355X_MICROARCH.md) -- plus the per-kernel table;
  "cpu_baseline": the CPU oracle (C port of the reference algorithm, conv loop on all host threads) on one frame.

--config 3 measures BASELINE configs[2] instead (full HSF + IGF point-cloud path, batch 2, camera features precomputed)
with the same contract and a stage table; the default (--config 2) is the headline.

Diagnostics (never the headline; the line is labelled): --f16 (single-pass f16 kernels, reduced precision) and the
--conv-diag knock-out kernels (garbage results, timing only; tools/conv_knockout.sh).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32-input MFMA peak (v_mfma_f32_16x16x4_f32)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense f16/bf16 MFMA peak (never the 2:1-sparsity figure)
POINTS_PER_FRAME = 300000
BATCH_PER_GPU = 4
CFG_ID = 2


def frames_for_rank(rank, world, batch):
    """Global frame ids processed by `rank`: disjoint across ranks, `batch` per rank (weak scaling)."""
    return [rank * batch + i for i in range(batch)]


def make_frames(rank, world, batch, num_points, frame_set=0):
    """Seeded synthetic sweeps of this rank.  ISF_BENCH_FRAME_CACHE=<dir> (tuning sweeps only: tools/conv_knockout.sh
    runs bench.py two dozen times) keeps the generated frames on disk; the frames are the same either way."""
    import numpy as np
    from isfusion_amd import synthetic
    cache = os.environ.get("ISF_BENCH_FRAME_CACHE", "")
    frames = []
    for f in frames_for_rank(rank, world, batch):
        seed = 1234 + 1000 * CFG_ID + f + 100000 * frame_set
        path = os.path.join(cache, f"frame_{seed}_{num_points}.npy") if cache else ""
        if path and os.path.exists(path):
            frames.append(np.load(path))
            continue
        pts = synthetic.lidar_sweeps(seed, num_points)
        if path:
            os.makedirs(cache, exist_ok=True)
            np.save(path, pts)
        frames.append(pts)
    return frames


def traffic_file():
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_traffic.json")))
    return files[-1] if files else ""


def pmc_traffic(cin, cout):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (newest profiles/r*_traffic.json: FETCH_SIZE, corrected x2 for gfx950, + WRITE_SIZE; counters cannot be collected
    from inside the timed process).  None when no summary is committed for that kernel."""
    path = traffic_file()
    try:
        per = json.load(open(path))["per_launch"]
    except (OSError, ValueError, KeyError):
        return None
    nt = min(cout // 16, 8)
    for k, v in per.items():
        if (k.startswith(f"spconv_f16x3_kernel<{cin}, {nt},") or k.startswith(f"spconv_dma_kernel<{cin}, {nt},")) and \
                "fetch_bytes" in v and "write_bytes" in v:
            return round(v["fetch_bytes"] + v["write_bytes"])
    return None


def conv_layer_bytes_flops(kind, cin, cout, K, n_in, n_out, pairs, s=4, dense_tables=False):
    """Compulsory traffic of one sparse-conv layer; s = bytes per element (4: fp32 / split rows,
    2: the f16 storage mode).  The rulebook term is the table the kernel actually has to read, per output row: 4 K bytes
    for the dense neighbour table (108 for 27 taps), 4 (K / 3) + 4 for the line-compressed table of the narrow layers
    (40) -- for the benchmark geometry within a few bytes per row of SURVEY's pairs * 8 (8.7 pairs per row at level 0,
    15.5 at level 3), and never less than what is read (the dense table was under-charged on the narrow layers)."""
    lines = cin <= 64 and cout <= 64 and not dense_tables and K % 3 == 0
    table = n_out * (4 * (K // 3) + 4 if lines else 4 * K)
    by = n_in * cin * s + n_out * cout * s + table + K * cin * cout * s
    fl = 2.0 * pairs * cin * cout
    return by, fl


def cpu_baseline(num_points, seed_frame):
    """Oracle ("port") timed on this host: 1 frame of `num_points` points through voxelize + VFE + encoder."""
    import numpy as np
    import torch
    import isfusion_amd as m
    import oracle
    from isfusion_amd import synthetic
    from isfusion_amd.norm import fold_bn
    oracle.build()
    lb = m.LidarBranch().randomize_weights_(0).randomize_bn_(1).eval()
    pts = synthetic.lidar_sweeps(seed_frame, num_points)
    vs, rg = lb.voxel_size, lb.point_cloud_range
    vfe = lb.pts_voxel_encoder
    bn1 = [t.numpy() for t in fold_bn(vfe.vfe_layers[0].norm)]
    bn2 = [t.numpy() for t in fold_bn(vfe.vfe_layers[1].norm)]
    plan = lb.pts_middle_encoder.plan_to_numpy()
    t0 = time.perf_counter()
    coors = np.concatenate([np.zeros((num_points, 1), np.int32), oracle.dynamic_voxelize(pts, vs, rg)], 1)
    vf, vc, _ = oracle.dynamic_vfe(pts, coors, vs, rg, vfe.vfe_layers[0].linear.weight.detach().numpy(), bn1,
                                   vfe.vfe_layers[1].linear.weight.detach().numpy(), bn2)
    bev, outs = oracle.sparse_encoder_forward(plan, vf, vc, 1)
    dt = time.perf_counter() - t0
    return dt, int(vf.shape[0])



# ------------------------------------------------------------------------------------------- --config 3
def cpu_baseline_fusion(seed):
    """Oracle ("port": C for the LiDAR branch, torch-CPU restatement for HSF / IGF + backbone stages) on ONE frame of
    30 k points with the full-size 180 x 180 fusion grid; bounded so that the default run stays within minutes."""
    import numpy as np
    import torch
    import oracle
    from isfusion_amd import synthetic
    from isfusion_amd.detector import ISFusionPtsPath
    from isfusion_amd.fusion_modules import seeded_state_dict
    from isfusion_amd.norm import fold_bn
    from oracle import fusion_ops as orc
    oracle.build()
    net = ISFusionPtsPath().eval()
    net._lidar.randomize_weights_(0).randomize_bn_(1)
    net.fusion_encoder.load_state_dict(seeded_state_dict(net.fusion_encoder, 100))
    net.pts_backbone.load_state_dict(seeded_state_dict(net.pts_backbone, 200))
    P = 30000
    p = synthetic.lidar_sweeps(seed, P)
    inp = synthetic.fusion_inputs(seed, 1)
    vs, rg = net.voxel_size, net.pc_range
    vfe = net.pts_voxel_encoder
    bn1 = [t.numpy() for t in fold_bn(vfe.vfe_layers[0].norm)]
    bn2 = [t.numpy() for t in fold_bn(vfe.vfe_layers[1].norm)]
    sd = {k: v.float() for k, v in net.fusion_encoder.state_dict().items()}
    sdb = {"bb." + k: v.float() for k, v in net.pts_backbone.state_dict().items()}
    plan = net.pts_middle_encoder.plan_to_numpy()
    t0 = time.perf_counter()
    coors = np.concatenate([np.zeros((P, 1), np.int32), oracle.dynamic_voxelize(p, vs, rg)], 1)
    vf, vc, _ = oracle.dynamic_vfe(p, coors, vs, rg, vfe.vfe_layers[0].linear.weight.detach().numpy(), bn1,
                                   vfe.vfe_layers[1].linear.weight.detach().numpy(), bn2)
    bev, _ = oracle.sparse_encoder_forward(plan, vf, vc, 1)
    v, c, n = oracle.hard_voxelize(p, net.pillar_size, rg, 12, 60000)
    pcoors = torch.from_numpy(np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], 1))
    with torch.no_grad():
        img_bev = orc.p2g_sample(torch.from_numpy(v)[..., :3], pcoors, torch.from_numpy(inp["img_feats"][1]),
                                 torch.from_numpy(inp["lidar2img"]), torch.from_numpy(inp["img_aug_matrix"]),
                                 torch.from_numpy(inp["lidar_aug_matrix"]), inp["input_shape"], 1, 180)
        bev_feats = orc.conv_module(torch.cat([img_bev, torch.from_numpy(bev)], 1), sd, "conv_fusion")
        g0 = orc.sstv2_forward(bev_feats, sd, "grid2region_att.0")
        ret, _, _ = orc.instance_fusion(bev_feats, g0, sd, 1, 180, 200)
        nxt, f0 = orc.secondv2_stage(ret, sdb, "bb", "stage1")
        _, f1 = orc.secondv2_stage(orc.sstv2_forward(nxt, sd, "grid2region_att.1"), sdb, "bb", "stage2")
    return time.perf_counter() - t0, P


def main_rehearsal(args):
    """--backend gloo: the multi-rank plumbing of this file on the CPU -- self-launch, rendezvous on 127.0.0.1, disjoint
    frame ids per rank, barrier + max-over-ranks clock, ONE line from rank 0 -- around a stub step.  Not a measurement:
    the line says so.  tests/test_host.py runs `python bench.py --gpus 2 --backend gloo` and checks two processes ran."""
    import torch
    import torch.distributed as dist
    from isfusion_amd import launch
    rank, world, _ = launch.world_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = frames_for_rank(rank, world, args.batch)

    def barrier():
        if world > 1:
            dist.barrier()

    real = bool(getattr(args, "real_step", False))
    checksum = None
    if real:
        # --real-step: every rank runs the REAL LidarBranch forward of its own frames on cuda:0 (a 1-GPU box: the ranks
        # share the device; gloo carries only the barrier and the clock reduction).  What it proves before the first
        # 8-GPU run: two processes of libisf_hip.so coexist (arena, count mailbox, pinned rings), the self-launch /
        # rendezvous / one-line path works with the actual step inside it, and the line carries n_gpus = world.
        assert torch.cuda.is_available(), "--real-step needs a GPU"
        import isfusion_amd as m
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        lb = m.LidarBranch().randomize_weights_(0).randomize_bn_(1).eval().to(dev)
        pts = [torch.from_numpy(p).to(dev) for p in make_frames(rank, world, args.batch, args.points)]

        def step():
            return lb(pts)
    else:
        x = torch.ones(64, 64)

        def step():
            nonlocal x
            x = (x @ x) / 64.0       # the stub step
            return x
    for _ in range(args.warmup):
        step()
    if real:
        torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    if real:
        torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if real:
        checksum = [float(out.double().abs().sum().item()), bool(torch.isfinite(out).all().item())]
    pids, frames, sums = [os.getpid()], [mine], [checksum]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        pids, frames, sums = [None] * world, [None] * world, [None] * world
        dist.all_gather_object(pids, os.getpid())
        dist.all_gather_object(frames, mine)
        dist.all_gather_object(sums, checksum)
    if rank == 0:
        line = {"metric": "REHEARSAL (gloo, stub step): launch / clock / line plumbing only", "value": 0.0,
                "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 4), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "rehearsal",
                "config": {"workload": "stub", "parallelism": f"dp{world}", "backend": "gloo",
                           "rank_pids": pids, "rank_frames": frames,
                           "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}}
        if real:
            line["metric"] = ("REHEARSAL (gloo clock, REAL LidarBranch step, all ranks share cuda:0): coexistence of the "
                              "ranks' library state + the launch / clock / line path; not a scaling measurement")
            line["value"] = round(args.batch * world * args.steps / dt, 2)
            line["dtype"], line["data"] = "f32 (f16x3 split-precision MFMA, fp32 accumulate)", "synthetic"
            line["config"].update(workload=f"LidarBranch forward, {args.points}-pt sweeps, batch={args.batch} per rank",
                                  rank_checksums=sums)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def main_fusion(args):
    """--config 3: BASELINE configs[2] as the line of its own (same JSON contract as the headline)."""
    import torch
    import torch.distributed as dist
    if args.lib:   # a side build of the library (tools/probes/build_side_lib.sh)
        from isfusion_amd import _lib
        _lib.LIB_PATH = os.path.abspath(args.lib)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (libisf_hip.so has no CPU path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if args.launch_stream:
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    line = fusion_leg(args, rank, world, dev, args.batch, args.steps, args.warmup, not args.no_cpu_baseline)
    if rank == 0:
        line["launches_per_forward"] = line.pop("_count_launches")()
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def attention_rooflines(net, dev, B, timed, p2g_ms=None, pillars=None):
    """north_star: "MFMA utilisation (attention)" -- the three attention kernels of the full path timed on their cfg-3
    shapes (own weights of the built network, random activations; HIP events around 10 launches each), algorithmic flops
    per DESIGN.md section 4 against the matrix peak of the arithmetic each computes in.  + Point-to-Grid in GB/s."""
    import torch
    from isfusion_amd import fusion_ops as ops
    enc = net.fusion_encoder
    out = {}
    with torch.no_grad():
        # fused window block (d = 128, 180 x 180 grid, 6 x 6 windows, 8 heads): per token 2*d*3d (qkv) + 2*2*36*d (scores,
        # PV) + 2*d*d (out projection) flop
        sst = enc.grid2region_att[0]
        win = enc.get_regions[0].window_shape[0]
        layer = sst.block_list[0].encoder_list[0]
        d = layer.win_attn.self_attn.out_proj.in_features
        S = enc.bev_size
        pc = ops._encoder_layer_cache(layer, S, win, 0, float(enc.get_regions[0].pos_temperature), dev, B)
        if pc.get("block") is not None:
            x = torch.randn(B * S * S, d, device=dev)
            ms, _ = timed(lambda: ops.window_block(x, pc["block"], pc["in_bias"], pc["table"], pc["out_bias"], layer.norm1, B,
                                                   S, d, layer.win_attn.nhead, win, 0))
            fl = B * S * S * (2.0 * d * 3 * d + 4.0 * win * win * d + 2.0 * d * d)
            out["window_block_kernel<128,16>"] = dict(
                ms=round(ms, 4), gflop=round(fl / 1e9, 2), tflops=round(fl / ms / 1e9, 1), peak=round(MFMA_F16_PEAK_TFLOPS / 3, 1),
                frac=round(3 * fl / ms / 1e9 / MFMA_F16_PEAK_TFLOPS, 4), bound="mfma (f16x3)",
                shape=f"{B * S * S} tokens, d {d}, {win}x{win} windows: qkv + attention + out projection + LN in one launch")
        # instance-to-scene cross attention: 32400 scene queries x 200 instance keys, 8 heads of 16 (flash style, MFMA)
        E, Q, nh = 128, 200, 8
        q = torch.randn(B * S * S, E, device=dev)
        kv = torch.randn(B * Q, 2 * E, device=dev)
        ms, _ = timed(lambda: ops.attention(q, kv, kv[:, E:], B, S * S, Q, E, nh, ldkv=2 * E))
        fl = 4.0 * B * S * S * Q * E
        out["attention_mfma16_kernel (32400 x 200)"] = dict(
            ms=round(ms, 4), gflop=round(fl / 1e9, 2), tflops=round(fl / ms / 1e9, 1), peak=round(MFMA_F16_PEAK_TFLOPS / 3, 1),
            frac=round(3 * fl / ms / 1e9 / MFMA_F16_PEAK_TFLOPS, 4), bound="mfma (f16x3)",
            shape=f"B {B}: {S * S} queries x {Q} keys, {nh} heads of {E // nh}")
        # per-channel 180 x 180 map attention (fp32 MFMA): 4 R^3 flop per map, B * C maps
        C = 128
        a, b = torch.randn(B, C, S, S, device=dev), torch.randn(B, C, S, S, device=dev)
        ms, _ = timed(lambda: ops.channel_attention(a, b))
        fl = 4.0 * S ** 3 * B * C
        out["channel_attention_mfma_kernel"] = dict(
            ms=round(ms, 4), gflop=round(fl / 1e9, 2), tflops=round(fl / ms / 1e9, 1), peak=MFMA_F32_PEAK_TFLOPS,
            frac=round(fl / ms / 1e9 / MFMA_F32_PEAK_TFLOPS, 4), bound="mfma (fp32)", shape=f"{B * C} maps of {S} x {S}")
    if p2g_ms and pillars is not None:
        M, T = int(pillars.shape[0]), int(pillars.shape[1])
        by = M * T * 12.0 + B * 6 * 24 * 66 * 256 * 4.0 + B * 256 * S * S * 4.0   # pillar points + NHWC camera map + canvas
        out["p2g_kernel"] = dict(ms=round(p2g_ms, 4), algorithmic_gbs=round(by / p2g_ms / 1e6, 1), peak=HBM_PEAK_GBS,
                                 frac=round(by / p2g_ms / 1e6 / HBM_PEAK_GBS, 4), bound="hbm / L2 latency",
                                 shape=f"{M} pillars x {T} points, 6 cameras, canvas [{B}, 256, {S}, {S}]")
    return out


def voxel_scatter_roofline(lb, frames, n=10):
    """north_star: "HBM GB/s (voxel scatter)" -- dynamic voxelization + DynamicVFE (hash-free rank index, counting sort
    of the points by voxel, two fused VFE layers with segmented max) timed standalone on the step's frames (HIP events
    around `n` calls of the two C entries; inside the engine the same kernels run fused with the frame marking), against
    the algorithmic bytes of DESIGN.md section 4: voxelize P * (20 + 16), VFE 3 passes over 32-byte point records +
    3 x N x 64 floats of voxel rows."""
    import torch
    from isfusion_amd.voxelize import dynamic_voxelize_batched
    vfe = lb.pts_voxel_encoder

    def run():
        pts, coors = dynamic_voxelize_batched(frames, lb.voxel_size, lb.point_cloud_range)
        return pts, vfe(pts.float(), coors)
    with torch.no_grad():
        pts, (vf, vc) = run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    P, N = int(pts.shape[0]), int(vf.shape[0])
    by = P * (20.0 + 16.0) + 3.0 * P * 32.0 + 3.0 * N * 64 * 4.0
    return dict(ms=round(ms, 4), points=P, voxels=N, algorithmic_gbs=round(by / ms / 1e6, 1), peak=HBM_PEAK_GBS,
                frac=round(by / ms / 1e6 / HBM_PEAK_GBS, 4), bound="hbm + latency",
                kernels="dynamic_voxelize + vfe_prep / count / scan / order / mean / layer1 / layer2 (isf_voxelize.hip, "
                        "isf_vfe.hip), two host read-backs included")


def count_launches(fn):
    """Device-side operations (kernels, memsets, copies) one call of `fn` puts on the GPU, from torch.profiler's device
    events; None when the profiler is not usable on the box."""
    import torch
    try:
        from torch.profiler import ProfilerActivity, profile
        fn()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            fn()
            torch.cuda.synchronize()
        evs = [e for e in prof.events() if getattr(e, "device_type", None) is not None and "cuda" in str(e.device_type).lower()]
        return len(evs) if evs else None
    except Exception as e:
        sys.stderr.write(f"count_launches: profiler unavailable ({e})\n")
        return None


def fusion_leg(args, rank, world, dev, B, steps, warmup, with_cpu_baseline):
    """BASELINE configs[2]: full IS-Fusion HSF + IGF forward -- LiDAR branch, pillar voxelization, ISFusionEncoder
    (Point-to-Grid, conv_fusion, Grid-to-Region x2, instance mining / context / instance-to-scene), SECONDV2 stages,
    SECONDFPN neck, TransFusionHeadV2.forward -- on B x P-point sweeps + precomputed (random) 6-camera feature maps,
    fp32-class arithmetic.  Same JSON contract as the headline; stage table from HIP events in a separate pass.
    Returns the line (rank 0) -- printed by --config 3, attached as "cfg3" to the headline line by the default run."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from isfusion_amd import _lib, synthetic
    from isfusion_amd.detector import ISFusionPtsPath
    from isfusion_amd.fusion_modules import seeded_state_dict

    net = ISFusionPtsPath().eval()
    net._lidar.randomize_weights_(0).randomize_bn_(1)
    for mod, seed in ((net.fusion_encoder, 100), (net.pts_backbone, 200), (net.pts_neck, 250), (net.pts_bbox_head, 300)):
        mod.load_state_dict(seeded_state_dict(mod, seed))
    net = net.to(dev)
    net.freeze()              # inference deployment: weights are static, the caches skip their change scans
    if args.graph:
        net.enable_graph()    # everything behind Point-to-Grid replays as one HIP graph per batch size (measured: no
                              # gain -- 7.95 vs 7.89 ms, profiles/r03_call10_graph.txt -- so off by default)
    sets = []
    for fs in range(max(1, args.frame_sets)):
        pts = [torch.from_numpy(p).to(dev) for p in make_frames(rank, world, B, args.points, 10 + fs)]
        inp = synthetic.fusion_inputs(5 + fs + 100 * rank, B)
        img_feats = tuple(torch.from_numpy(x).to(dev) for x in inp["img_feats"])
        kw = dict(lidar2img=torch.from_numpy(inp["lidar2img"]), img_aug_matrix=torch.from_numpy(inp["img_aug_matrix"]),
                  lidar_aug_matrix=torch.from_numpy(inp["lidar_aug_matrix"]))
        metas = [dict(input_shape=inp["input_shape"]) for _ in range(B)]
        sets.append((pts, img_feats, metas, kw))

    def step(i):
        pts, img_feats, metas, kw = sets[i % len(sets)]
        return net.forward_pts(pts, img_feats, metas, **kw)

    def barrier():
        if world > 1:
            dist.barrier()

    for i in range(warmup):
        out = step(i)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out = step(i)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    heat = out[0][0]["heatmap"]
    assert torch.isfinite(heat).all()

    if rank == 0:
        # ---- stage table (separate pass, HIP events on the launch stream; not part of the timed region)
        def timed(fn, n=10):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                r = fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n, r
        pts, img_feats, metas, kw = sets[0]
        enc, bb = net.fusion_encoder, net.pts_backbone
        stages = {}
        with torch.no_grad():
            stages["lidar_branch"], x = timed(lambda: net._lidar(pts, want_stats=True, time_layers=True))
            st, tab = net._lidar.last_stats, net._lidar.conv_layer_table()
            stages["pillar_voxelize"], (pil, npts, pco) = timed(lambda: net.voxelize(pts, "pillar"))
            ekw = dict(kw, pts_metas=dict(pillars=pil, pillars_num_points=npts, pillar_coors=pco), img_metas=metas,
                       pts_backbone=bb)
            stages["p2g"], img_bev = timed(lambda: enc.img_fv_to_bev([img_feats[1]], B, **ekw))
            stages["conv_fusion"], bev = timed(lambda: enc.fuse(img_bev, x))
            stages["grid2region_0"], g0 = timed(lambda: enc.grid2region(0, bev))
            stages["instance_fusion"], (ret, _) = timed(lambda: enc.instance_fusion(bev, g0, B))
            stages["second_stage1"], (nxt, _, f0) = timed(lambda: bb([ret], "stage1"))
            stages["grid2region_1"], g1 = timed(lambda: enc.grid2region(1, nxt))
            stages["second_stage2"], (_, _, f1) = timed(lambda: bb([g1], "stage2"))
            if net.pts_neck.dense_conv == "hip" and net.pts_bbox_head.dense_conv == "hip":   # as forward_pts runs them
                stages["neck"], nk = timed(lambda: net.pts_neck.forward_split([f0, f1]))
                stages["head"], _ = timed(lambda: net.pts_bbox_head.forward_split(nk))
            else:
                stages["neck"], nk = timed(lambda: net.pts_neck([f0, f1]))
                stages["head"], _ = timed(lambda: net.pts_bbox_head(nk, img_feats, metas))
        stages = {k: round(v, 3) for k, v in stages.items()}
        # ---- roofline of the dominant kernel: the sparse-conv kernel family (LiDAR branch layers by template
        # instantiation; the 3x3 dense convs of conv_fusion / IGF / SECONDV2 run on the same kernels over a dense rulebook)
        groups = {}
        for i, (kind, cin, cout, K) in enumerate(tab):
            by, fl = conv_layer_bytes_flops(kind, cin, cout, K, st.num_in[i], st.num_out[i], st.pairs[i])
            g = groups.setdefault(f"spconv_mfma<cin={cin},cout={cout}>", dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
            g["ms"] += float(st.ms[i]); g["flops"] += fl; g["bytes"] += by; g["launches"] += 1
        # dense 3x3 convs: flops = 2 * 9 * Cin * Cout per output cell (every tap present except at the border)
        dense = dict(ms=stages["second_stage1"] + stages["second_stage2"], launches=12,
                     flops=2.0 * 9 * B * (180 * 180 * 128 * 128 * 6 + 90 * 90 * (128 * 256 + 5 * 256 * 256)))
        name, dom = max(groups.items(), key=lambda kv: kv[1]["ms"])
        t_s = dom["ms"] * 1e-3
        tfl = dom["flops"] / t_s / 1e12 if t_s > 0 else 0.0
        roof = dict(bound="mfma", achieved=round(tfl, 3), peak=round(MFMA_F16_PEAK_TFLOPS / 3, 1), unit="TFLOP/s",
                    frac=round(3 * tfl / MFMA_F16_PEAK_TFLOPS, 4), traffic=None,
                    arithmetic="f16x3 split MFMA: 3 f16 passes per fp32 product, fp32 accumulate; peak = 2500 / 3",
                    kernel=name + " (LiDAR branch, the largest single-kernel share of the step)",
                    launches_per_step=dom["launches"], avg_launch_ms=round(dom["ms"] / dom["launches"], 4),
                    algorithmic_bytes_per_launch=round(dom["bytes"] / dom["launches"]),
                    per_kernel={k: dict(ms=round(v["ms"], 4), tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 3)
                                        if v["ms"] > 0 else 0, launches=v["launches"]) for k, v in groups.items()},
                    dense_conv_stages=dict(ms=round(dense["ms"], 3), launches=dense["launches"],
                                           tflops=round(dense["flops"] / (dense["ms"] * 1e-3) / 1e12, 2),
                                           frac=round(3 * dense["flops"] / (dense["ms"] * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4),
                                           note="SECONDV2 stage 1 + 2: twelve 3x3 convs on the sparse-conv kernel over "
                                                "the dense-grid rulebook (includes their layout conversions)"),
                    stages_ms=stages, stages_sum_ms=round(sum(stages.values()), 3))
        try:
            roof["attention_kernels"] = attention_rooflines(net, dev, B, timed, stages.get("p2g"), pil)
        except Exception as e:                              # the line stands without the extra entries
            roof["attention_kernels"] = {"error": str(e)[:200]}
        line = {
            "metric": "nuScenes frames/sec forward (0.075 voxel), full HSF+IGF point-cloud path incl. neck + head forward",
            "value": round(B * world * steps / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 (f16x3 split-precision MFMA, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: full IS-Fusion HSF+IGF forward (LiDAR branch + pillar voxelize + "
                                   "ISFusionEncoder + SECONDV2 stages + SECONDFPN + TransFusionHeadV2.forward), 6-camera "
                                   f"feature maps precomputed (random), synthetic {args.points}-pt sweeps, batch={B}/GPU, "
                                   "random-init weights, eval BN, fp32",
                       "points_per_frame": args.points, "batch_per_gpu": B, "parallelism": f"dp{world}",
                       "frame_sets_rotated": len(sets)},
            "roofline": roof,
        }
        if world == 1 and with_cpu_baseline:
            cdt, cp = cpu_baseline_fusion(1234 + 1000 * 3)
            import oracle
            line["cpu_baseline"] = {
                "value": round(1.0 / cdt, 4), "unit": f"frames/s ({cp}-pt frame)", "cores": oracle.num_threads(),
                "kind": "port",
                "sample": f"oracle composition (C LiDAR branch with the conv loop on {oracle.num_threads()} host "
                          f"threads + torch-CPU restatement of HSF / IGF / SECONDV2 stages, full 180 x 180 grid, no neck / "
                          f"head) on 1 frame of {cp} points took {cdt:.2f} s",
                "sample_seconds": round(cdt, 2)}
        # the launch count comes from torch.profiler, and an initialised profiler slows every later launch of the process
        # (measured: the training leg behind it 48 -> 134 ms per step): the caller runs `count` after ALL timed legs
        line["launches_per_forward"] = None
        line["_count_launches"] = lambda: count_launches(lambda: step(0))
        line["config"]["hip_graph"] = ("off" if not args.graph else
                                       "ISFusionPtsPath.enable_graph(): conv_fusion .. head (shape-static per batch size) "
                                       "captured once and replayed; LiDAR branch, pillar voxelization and Point-to-Grid eager")
        line["config"]["frozen_caches"] = ("net.freeze(): inference deployment, the packed-weight caches skip their "
                                           "per-call parameter-change scan (about 0.4 ms of host time per forward)")
        return line
    return None


# ------------------------------------------------------------------------------ BASELINE configs[4] / configs[3] legs
def pipelined_leg(lb, frame_sets, steps, warmup, in_flight=2, **kw):
    """The headline workload with `in_flight` batches in flight: call k runs on HIP stream k % in_flight (the library's
    workspaces, mailboxes and geometry side streams are per (device, stream)), so the voxelization / VFE / level-0
    geometry of batch k + 1 -- a fifth of a step of small kernels that leave most of the chip idle -- overlap the
    convolutions of batch k.  Same kernels, same launches, same bits (checked against a serial call below); nothing
    is synchronised between the steps, all of them start and end inside the timed region.  Reported BESIDE the
    headline, whose per-kernel figures (roofline, rocprof summary) are only meaningful for serial launches."""
    import torch
    streams = [torch.cuda.Stream() for _ in range(in_flight)]
    cur = torch.cuda.current_stream()
    for st in streams:
        st.wait_stream(cur)
    outs = [None] * in_flight

    def run(k):
        with torch.cuda.stream(streams[k % in_flight]):
            outs[k % in_flight] = lb(frame_sets[k % len(frame_sets)], **kw)

    for k in range(warmup):
        run(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        run(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    last = (steps - 1)
    want = lb(frame_sets[last % len(frame_sets)], **kw)
    torch.cuda.synchronize()
    same = bool(torch.equal(want, outs[last % in_flight]))
    B = len(frame_sets[0])
    return {"metric": "LiDAR-branch frames per second with batches in flight on separate HIP streams",
            "value": round(steps * B / dt, 2), "unit": "frames/s", "ms_per_step": round(dt / steps * 1e3, 3),
            "steps": steps, "warmup": warmup, "in_flight": in_flight, "bit_identical_to_serial": same,
            "note": "steps overlap (batch k+1's voxelize / VFE / geometry beside batch k's convolutions); every step's "
                    "work is inside the timed region; the headline value above is the serial figure"}


def f16_stress_leg(args, rank, world, dev, steps=16, warmup=4, B=4, points=500000, voxel=0.05):
    """BASELINE configs[4] on one GPU, attached to the headline line as "cfg5_f16": the LiDAR branch at 0.05 m voxels
    (sparse shape [41, 2160, 2160], BEV 270 x 270) on 500 k-point sweeps in the f16 STORAGE mode (isf_encoder_options
    .precision = 2: f16 rows between the layers, f16 operands, fp32 accumulate -- the reference's indice_conv_half data
    types): frames/s and, per conv kernel, the algorithmic HBM rate (SURVEY 8d bytes with 2-byte elements) as a fraction
    of the 8 TB/s peak -- the "HBM-bound sparse-conv roofline run"."""
    import numpy as np
    import torch
    import isfusion_amd as m
    me = dict(m.ISFUSION_0075["pts_middle_encoder"])
    side = int(round(108.0 / voxel))
    me["sparse_shape"] = [41, side, side]
    lb = m.LidarBranch(voxel_size=[voxel, voxel, 0.2], pts_middle_encoder=me)
    lb = lb.randomize_weights_(0).randomize_bn_(1).eval().to(dev).freeze()
    sets = [[torch.from_numpy(q).to(dev) for q in make_frames(rank, world, B, points, 20 + fs)] for fs in range(2)]
    for i in range(warmup):
        out = lb(sets[i % 2], precision=2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out = lb(sets[i % 2], precision=2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    tab = lb.conv_layer_table()
    groups, tot_ms, tot_by = {}, 0.0, 0.0
    ns = 4
    for i in range(ns):                                   # per-layer HIP events: separate passes, both frame sets
        lb(sets[i % 2], precision=2, time_layers=True)
        st = lb.last_stats
        for j, (kind, cin, cout, K) in enumerate(tab):
            by, fl = conv_layer_bytes_flops(kind, cin, cout, K, st.num_in[j], st.num_out[j], st.pairs[j], 2)
            g = groups.setdefault(f"spconv<cin={cin},cout={cout}>", dict(ms=0.0, bytes=0.0, flops=0.0, launches=0))
            g["ms"] += float(st.ms[j]) / ns; g["bytes"] += by / ns; g["flops"] += fl / ns; g["launches"] += 1
            tot_ms += float(st.ms[j]) / ns; tot_by += by / ns
    for g in groups.values():
        g["launches"] //= ns
    per = {k: dict(ms=round(v["ms"], 4), launches=v["launches"], algorithmic_gbs=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                   hbm_frac=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                   tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                   mfma_frac=round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4)) for k, v in groups.items()
           if v["ms"] > 0}
    return {"metric": "nuScenes-shaped frames/sec forward, LiDAR branch at 0.05 m voxels (BASELINE configs[4] on 1 GPU)",
            "value": round(B * steps / dt, 2), "unit": "frames/s", "steps": steps, "warmup": warmup,
            "ms_per_step": round(dt / steps * 1e3, 3),
            "dtype": "f16 storage, f16 operands, fp32 accumulate (isf_encoder_options.precision = 2)",
            "config": {"workload": f"BASELINE configs[4] shape on one GPU: voxel {voxel} m, sparse shape [41, {side}, {side}], "
                                   f"{points}-pt synthetic sweeps, batch={B}, LiDAR branch only (cameras are out of scope); "
                                   "the 8-GPU leg is N replicas (bench.py --gpus 8)", "batch_per_gpu": B,
                       "points_per_frame": points},
            "conv_ms_per_step": round(tot_ms, 3), "all_conv_algorithmic_gbs": round(tot_by / (tot_ms * 1e-3) / 1e9, 1),
            "all_conv_hbm_frac": round(tot_by / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "voxels_level0": int(st.num_in[0]), "per_kernel": per}


def train_leg(args, rank, world, dev, steps=10, B=2, points=300000):
    """BASELINE configs[3] on one GPU, attached as "cfg4_train": the full point-cloud path's training step (forward with
    gradients through LiDAR branch, fusion encoder, SECONDV2 stages and neck + stand-in loss + backward + SGD step) under
    torch.autocast(bfloat16), B frames per GPU.  ms per step from the wall clock around `steps` synchronised steps;
    launches and the device-time share from one further step under torch.profiler (None when the profiler is not
    usable on the box).  The 8-GPU leg wraps the same module in DDP (tools/train_step.py --gpus 8: RCCL all-reduce of the
    gradients only).  Round 5: the leg runs at the headline's sweep size (2 x 300 000 points, the config's
    batch of 2 per GPU) for 10 timed steps after 2 warm-ups; rounds 2-4 reported 60 000-point sweeps x 3 steps."""
    import torch
    from isfusion_amd import synthetic
    from isfusion_amd.detector import ISFusionPtsPath
    from isfusion_amd.fusion_modules import seeded_state_dict
    net = ISFusionPtsPath().train()
    net._lidar.randomize_weights_(0).randomize_bn_(1)
    for mod, seed in ((net.fusion_encoder, 100), (net.pts_backbone, 200), (net.pts_neck, 250)):
        mod.load_state_dict(seeded_state_dict(mod, seed))
    for q in net.pts_bbox_head.parameters():
        q.requires_grad_(False)                       # head losses / target assignment: training control plane
    net = net.to(dev)
    opt = torch.optim.SGD([q for q in net.parameters() if q.requires_grad], lr=1e-4, momentum=0.9)
    pts = [torch.from_numpy(synthetic.lidar_sweeps(9000 + 100 * rank + i, points)).to(dev) for i in range(B)]
    inp = synthetic.fusion_inputs(7 + rank, B)
    img = tuple(torch.from_numpy(x).to(dev) for x in inp["img_feats"])
    kw = dict(lidar2img=torch.from_numpy(inp["lidar2img"]), img_aug_matrix=torch.from_numpy(inp["img_aug_matrix"]),
              lidar_aug_matrix=torch.from_numpy(inp["lidar_aug_matrix"]))
    metas = [dict(input_shape=inp["input_shape"]) for _ in range(B)]
    losses = []

    def one():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out, hm = net.forward_train_pts(pts, img, metas, **kw)
            loss = (out[0].float() ** 2).mean() + hm.float().sigmoid().mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    for _ in range(2):
        losses.append(float(one()))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses.append(float(one()))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    launches = kernel_ms = None
    try:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            one()
            torch.cuda.synchronize()
        evs = [e for e in prof.events() if getattr(e, "device_type", None) is not None and "cuda" in str(e.device_type).lower()]
        if evs:
            launches = len(evs)
            kernel_ms = sum(float(getattr(e, "device_time", 0.0) or getattr(e, "cuda_time", 0.0)) for e in evs) / 1e3
    except Exception as e:                             # the measurement above stands without the profile
        launches, kernel_ms = None, None
        sys.stderr.write(f"train_leg: profiler unavailable ({e})\n")
    assert all(v == v and abs(v) < 1e30 for v in losses), losses
    return {"metric": "training step of the full point-cloud path (BASELINE configs[3] on 1 GPU)", "value": round(ms, 2),
            "unit": "ms per step", "higher_is_better": False, "steps": steps, "warmup": 2,
            "dtype": "torch.autocast(bfloat16): the remaining stock ops (neck deconvs, the 10-class heat-map conv, torch "
                     "linears of the VFE) in bf16; the HIP sparse-conv autograd Functions -- the LiDAR encoder's and, since "
                     "round 6, the dense 3x3 conv + BatchNorm2d stacks (dense_train.py) -- run single-pass f16 MFMA with fp32 "
                     "accumulate under autocast (spconv.AUTOCAST_HALF, the reference's custom_fwd(cast_inputs=torch.half)); "
                     "the other HIP Functions compute in fp32-class f16x3 arithmetic",
            "config": {"workload": f"forward_train_pts + stand-in loss + backward + SGD step, batch={B}/GPU, {points}-pt "
                                   "synthetic sweeps, 6-camera feature maps precomputed (random); detection losses / target "
                                   "assignment are the reference's control plane (out of scope)", "batch_per_gpu": B,
                       "points_per_frame": points, "parallelism": "dp1 (8-GPU: tools/train_step.py --gpus 8, DDP over RCCL)"},
            "launches_per_step": launches, "device_kernel_ms_per_step": None if kernel_ms is None else round(kernel_ms, 2),
            "kernel_time_share": None if kernel_ms is None else round(kernel_ms / ms, 3),
            "losses": [round(v, 5) for v in losses]}


def compact_line(line, legs):
    """The ONE line the driver parses: the contract's keys, the `roofline` / `cpu_baseline` objects reduced to the fields the
    contract names (the full objects are on the "headline_detail" line printed just before), and scalar copies of every
    secondary leg's headline numbers at the top level, in front of the nested objects.  Kept under 2 KB."""
    roof = line["roofline"]
    keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launches_per_step", "avg_launch_ms",
            "algorithmic_bytes_per_launch", "conv_ms_per_step", "all_conv_tflops", "all_conv_algorithmic_gbs")
    small_roof = {k: roof[k] for k in keep if k in roof}
    vs = roof.get("voxel_scatter")
    if isinstance(vs, dict) and "frac" in vs:
        small_roof["voxel_scatter"] = {k: vs[k] for k in ("ms", "algorithmic_gbs", "frac") if k in vs}
    out = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data")}
    c3, c5, c4, pl = legs.get("cfg3"), legs.get("cfg5_f16"), legs.get("cfg4_train"), legs.get("pipelined")
    if c3:
        out.update(cfg3_fps=c3["value"], cfg3_ms=c3["ms_per_step"], cfg3_launches=c3.get("launches_per_forward"))
    if c5:
        out.update(cfg5_fps=c5["value"], cfg5_ms=c5["ms_per_step"], cfg5_hbm_frac=c5.get("all_conv_hbm_frac"))
    if c4:
        out.update(train_ms=c4["value"], train_launches=c4.get("launches_per_step"))
    if pl and isinstance(pl, dict) and "value" in pl:
        out.update(pipelined_fps=pl["value"])
    cfg = line["config"]
    out["config"] = {"workload": cfg["workload"][:300], "points_per_frame": cfg.get("points_per_frame"),
                     "batch_per_gpu": cfg.get("batch_per_gpu"), "parallelism": cfg.get("parallelism"),
                     "collectives": cfg.get("collectives", "")[:80]}
    out["roofline"] = small_roof
    cb = line.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: (v[:260] if isinstance(v, str) else v) for k, v in cb.items()}
    out["detail_lines"] = ["headline_detail"] + list(legs.keys())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3],
                    help="2 (default, the headline): BASELINE configs[1], LiDAR-only branch, batch 4; "
                         "3: BASELINE configs[2], full HSF + IGF forward (camera features precomputed), batch 2")
    ap.add_argument("--frame-sets", type=int, default=2,
                    help="distinct input batches rotated through the steps (so a step is not a cache-warm replay of "
                         "the previous one)")
    ap.add_argument("--points", type=int, default=POINTS_PER_FRAME)
    ap.add_argument("--batch", type=int, default=0, help="frames per GPU and step (default 4 for --config 2, 2 for 3)")
    ap.add_argument("--cpu-points", type=int, default=300000,
                    help="points of the CPU-baseline sample frame (one full 300 k-point frame: about 30 s on one "
                         "core, about 10 s on 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cfg3", action="store_true",
                    help="skip the BASELINE configs[2] leg the default run appends to the headline line as \"cfg3\"")
    ap.add_argument("--cfg3-steps", type=int, default=30)
    ap.add_argument("--no-cfg4", action="store_true",
                    help="skip the BASELINE configs[3] leg (1-GPU bf16-autocast training step) appended as \"cfg4_train\"")
    ap.add_argument("--no-cfg5", action="store_true",
                    help="skip the BASELINE configs[4] leg (0.05 m voxels, 500 k points, f16 storage) appended as \"cfg5_f16\"")
    ap.add_argument("--graph", action="store_true",
                    help="config 3: replay the shape-static tail (conv_fusion .. head) as one HIP graph instead of ~330 "
                         "eager launches")
    ap.add_argument("--fp32", action="store_true", help="force the fp32 MFMA conv kernels")
    ap.add_argument("--lib", default="",
                    help="A/B and probe builds: load this build of libisf_hip.so instead of the in-tree one (results are "
                         "not checked for finiteness: knock-out builds produce garbage)")
    ap.add_argument("--launch-stream", action="store_true",
                    help="run every leg on a private (non-NULL) HIP stream instead of torch's default stream")
    ap.add_argument("--no-pipelined", action="store_true",
                    help="skip the two-batches-in-flight leg appended as \"pipelined\"")
    ap.add_argument("--conv-diag", type=int, default=0, choices=[0, 2, 4, 6, 8, 16, 32, 48, 64, 96, 128, 192, 256, 512, 704] + [1024 * v for v in range(1, 8)] + [16384, 32768, 49152, 65536, 131072, 262144, 524288, 262144 + 32, 262144 + 64, 1048576, 2097152, 4194304, 8388608, 16777216, 33554432, 67108864, 134217728, 268435456, 536870912] + [512 + 1024 * v for v in range(1, 16)],
                    help="DIAGNOSTIC ONLY: knock-out timing modes of the sparse-conv kernel (isf_encoder_options.diagnostic; "
                         "results are garbage, the line is labelled)")
    ap.add_argument("--stage-rows", type=int, default=0,
                    help="LDS-staged input rows per conv tile (isf_encoder_options.stage_rows): 0 = the library's "
                         "per-layer default, -1 = staging off, N = N rows on the layers of --stage-mask")
    ap.add_argument("--stage-mask", type=lambda v: int(v, 0), default=0,
                    help="with --stage-rows N: bit i = conv layer i runs the staged kernel (0 = every layer)")
    ap.add_argument("--voxel", type=float, default=0.075,
                    help="DIAGNOSTIC: x / y voxel size; 0.05 with --points 500000 --f16 is BASELINE configs[4] (the HBM-bound "
                         "stress run); the headline is 0.075")
    ap.add_argument("--f16", action="store_true",
                    help="DIAGNOSTIC ONLY: f16 storage + single-pass f16 conv kernels (isf_encoder_options.precision 2: the "
                         "reference's indice_conv_half data types, BASELINE configs[4] dtype); reduced precision, never "
                         "the headline line")
    ap.add_argument("--real-step", action="store_true",
                    help="with --backend gloo: the REAL LidarBranch forward per rank, all ranks on cuda:0 (1-GPU rehearsal "
                         "of the multi-rank run; -m gpu test)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL (the measurement); gloo = CPU REHEARSAL of the multi-rank launch / clock / JSON "
                         "line with a stub step (tests/test_host.py), never a measurement")
    args = ap.parse_args()
    if args.batch <= 0:
        args.batch = BATCH_PER_GPU if args.config == 2 else 2
    # `python bench.py --gpus N` started plainly: become N ranks (one per GPU) under torch.distributed.run; fewer than N
    # devices is an error.  Under a launcher (WORLD_SIZE set) this is a no-op.
    from isfusion_amd import launch
    launch.self_launch(args.gpus, args.backend)
    if args.backend == "gloo":
        return main_rehearsal(args)
    if args.config == 3:
        return main_fusion(args)

    import torch
    import torch.distributed as dist
    import isfusion_amd as m
    if args.lib:
        from isfusion_amd import _lib
        _lib.LIB_PATH = os.path.abspath(args.lib)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (libisf_hip.so has no CPU path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if args.launch_stream:
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    # under a launcher (torch.distributed.run exports MASTER_PORT) the collective path runs even with ONE rank: RCCL
    # init, the barriers around the timed region, the max-reduce of the clock -- so that the only thing a multi-GPU run
    # adds to a tested path is N (tests/test_gpu_train.py)
    use_dist = world > 1 or ("MASTER_PORT" in os.environ and "WORLD_SIZE" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)  # RCCL on ROCm

    precision = 2 if args.f16 else 1 if args.fp32 else 0
    stage_kw = dict(stage_rows=args.stage_rows, stage_mask=args.stage_mask)
    if abs(args.voxel - 0.075) > 1e-9:   # BASELINE configs[4]: 0.05 m voxels -> sparse shape [41, 2160, 2160], BEV 270 x 270
        me = dict(m.ISFUSION_0075["pts_middle_encoder"])
        side = int(round(108.0 / args.voxel))
        me["sparse_shape"] = [41, side, side]
        lb = m.LidarBranch(voxel_size=[args.voxel, args.voxel, 0.2], pts_middle_encoder=me)
    else:
        lb = m.LidarBranch()
    lb = lb.randomize_weights_(0).randomize_bn_(1).eval().to(dev).freeze()
    frame_sets = [[torch.from_numpy(p).to(dev) for p in make_frames(rank, world, args.batch, args.points, fs)]
                  for fs in range(max(1, args.frame_sets))]
    frames = frame_sets[0]
    torch.cuda.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()

    for i in range(args.warmup):
        out = lb(frame_sets[i % len(frame_sets)], precision=precision, conv_diag=args.conv_diag, **stage_kw)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()

    # timed region: exactly K steps.  Per-layer hipEvents (recorded inside the library on the launch stream) cost the
    # step ~0.28 ms: an event between two launches holds the next kernel back ~12 us and reading the times needs a
    # host sync at the end of the step (tools/timeline_gaps.py, profiles/r02_call22_timeline_gaps.txt).  So the
    # events ride on every `stride`-th step of the timed region (an odd stride: both rotating frame sets are
    # sampled); the other steps run exactly as a caller runs them.
    nl = len(lb.conv_layer_table())
    stride = max(1, args.steps // 3) | 1   # three sampled steps (round 5: five; each costs the step ~0.3 ms)
    samples = []   # (ms, num_in, num_out, pairs) per layer of every sampled step
    t0 = time.perf_counter()
    for step in range(args.steps):
        sampled = step % stride == 0
        out = lb(frame_sets[step % len(frame_sets)], time_layers=sampled, precision=precision, conv_diag=args.conv_diag,
                 **stage_kw)
        if sampled:
            st = lb.last_stats
            samples.append(([st.ms[i] for i in range(nl)], [st.num_in[i] for i in range(nl)],
                            [st.num_out[i] for i in range(nl)], [st.pairs[i] for i in range(nl)]))
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # (the two-batches-in-flight leg runs LAST, after every other leg: the four extra streams it leaves behind -- two launch
    # streams + the library's geometry side streams -- make every later launch on torch's default stream slower; measured on
    # the configs[2] leg, same box, alternating: 6.71 / 6.75 ms on its own or without this leg before it, 6.92 / 6.92 with it:
    # gpurun_out/r06_cfg3_ctx)
    want_pipelined = world == 1 and not args.no_pipelined and not args.lib
    diag = (args.conv_diag & 15) != 0 or ((args.conv_diag >> 10) & 3) != 0   # timing diagnostics: results are garbage (16 = sharing off, 32 = uniform tiles: valid)
    assert diag or args.lib or torch.isfinite(out).all()

    if rank == 0:
        import numpy as np
        tab = lb.conv_layer_table()
        ns = len(samples)
        ms_layer = np.array([smp[0] for smp in samples]).mean(axis=0)       # per layer, per step
        groups = {}
        tot_bytes = tot_flops = 0.0
        for ms, n_in, n_out, pairs in samples:                               # per-step figures: averaged over the samples
            for i, (kind, cin, cout, K) in enumerate(tab):
                by, fl = conv_layer_bytes_flops(kind, cin, cout, K, n_in[i], n_out[i], pairs[i], 2 if args.f16 else 4,
                                                bool(args.conv_diag & (16384 | 128)))
                tot_bytes += by / ns
                tot_flops += fl / ns
                g = groups.setdefault(f"spconv_mfma<cin={cin},cout={cout}>", dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
                g["ms"] += float(ms[i]) / ns; g["flops"] += fl / ns; g["bytes"] += by / ns
        for i, (kind, cin, cout, K) in enumerate(tab):
            groups[f"spconv_mfma<cin={cin},cout={cout}>"]["launches"] += 1
        name, dom = max(groups.items(), key=lambda kv: kv[1]["ms"])
        cin_dom, cout_dom = [int(v) for v in __import__("re").findall(r"\d+", name)]
        t_s = dom["ms"] * 1e-3
        tflops = dom["flops"] / t_s / 1e12 if t_s > 0 else 0.0
        gbs = dom["bytes"] / t_s / 1e9 if t_s > 0 else 0.0
        # precision 1: the f16x3 split kernels EXECUTE 3 f16 MFMA passes per algorithmic fp32 multiply-add, so
        # the matrix-pipe roofline is the f16 peak against 3x the algorithmic flops
        split = st.precision == 1
        mult, peak = (3.0, MFMA_F16_PEAK_TFLOPS) if split else (1.0, MFMA_F32_PEAK_TFLOPS)
        if args.f16:
            mult = 1.0   # one f16 MFMA pass per product
        t_roof_mfma = mult * dom["flops"] / (peak * 1e12)
        t_roof_hbm = dom["bytes"] / (HBM_PEAK_GBS * 1e9)
        if t_roof_mfma >= t_roof_hbm:
            # achieved = ALGORITHMIC flops / launch time; the peak is what the matrix pipe can deliver per
            # algorithmic flop with this arithmetic: dense f16 peak / 3 passes (or the fp32 MFMA peak)
            roof = dict(bound="mfma", achieved=round(tflops, 3), peak=round(peak / mult, 1), unit="TFLOP/s",
                        frac=round(mult * tflops / peak, 4),
                        arithmetic=f"f16x3 split MFMA: 3 f16 passes per fp32 product, fp32 accumulate; peak = "
                                   f"{MFMA_F16_PEAK_TFLOPS:.0f} dense f16 TFLOP/s / 3" if split else "fp32 MFMA")
        else:
            roof = dict(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(gbs / HBM_PEAK_GBS, 4))
        roof.update(traffic=pmc_traffic(cin_dom, cout_dom), traffic_source="profiles/" + os.path.basename(traffic_file()) + " (rocprofv3 --pmc "
                    "FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, bytes per launch)",
                    algorithmic_bytes_per_launch=round(dom["bytes"] / dom["launches"]), kernel=name, launches_per_step=dom["launches"],
                    avg_launch_ms=round(dom["ms"] / dom["launches"], 4),
                    timing=f"per-layer HIP events inside the library on the launch stream, on {ns} of the {args.steps} "
                           f"timed steps (every {stride}th; both frame sets); per-step figures are their mean",
                    algorithmic_gbs=round(gbs, 1), algorithmic_tflops=round(tflops, 3),
                    conv_ms_per_step=round(float(ms_layer.sum()), 3),
                    all_conv_tflops=round(tot_flops / (ms_layer.sum() * 1e-3) / 1e12, 3) if ms_layer.sum() > 0 else 0,
                    all_conv_algorithmic_gbs=round(tot_bytes / (ms_layer.sum() * 1e-3) / 1e9, 1) if ms_layer.sum() > 0 else 0,
                    per_kernel={k: dict(ms=round(v["ms"], 4), tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 3)
                                        if v["ms"] > 0 else 0, gbs=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)
                                        if v["ms"] > 0 else 0, launches=v["launches"]) for k, v in groups.items()},
                    voxels_per_level=[int(samples[-1][1][0])] + [int(samples[-1][2][i]) for i, t in enumerate(tab) if t[0] == "spconv"])
        try:
            roof["voxel_scatter"] = voxel_scatter_roofline(lb, frame_sets[0])
        except Exception as e:                              # the line stands without the extra entry
            roof["voxel_scatter"] = {"error": str(e)[:200]}
        frames_total = args.batch * world * args.steps
        line = {
            "metric": "nuScenes frames/sec forward (0.075 voxel), LiDAR branch voxelize+spconv->BEV",
            "value": round(frames_total / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "DIAGNOSTIC (--conv-diag knock-out kernels: results are garbage, timing only)" if diag
            else "f16 storage, f16 operands, fp32 accumulate (DIAGNOSTIC: reduced precision, not the headline)" if args.f16
            else "f32 (f16x3 split-precision MFMA, fp32 accumulate)" if st.precision == 1 else "f32",
            "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: isfusion_0075voxel LiDAR-only branch (dynamic voxelize + "
                                    "DynamicVFE + 21-layer SparseEncoder -> BEV [B,512,180,180]), synthetic "
                                    f"nuScenes-shaped {args.points}-pt sweeps, batch={args.batch}/GPU, random-init "
                                    "weights, eval BN, fp32") if abs(args.voxel - 0.075) < 1e-9 and not args.f16 else
                                   (f"DIAGNOSTIC (BASELINE configs[4] shape when --voxel 0.05 --points 500000 --f16): LiDAR "
                                    f"branch at {args.voxel} m voxels, {args.points}-pt sweeps, batch={args.batch}/GPU, "
                                    + ("f16 storage" if args.f16 else "fp32-class")),
                       "points_per_frame": args.points, "batch_per_gpu": args.batch, "parallelism": f"dp{world}",
                       "collectives": f"RCCL {launch.rccl_version()} over xGMI (barrier + clock reduction only: the "
                                      f"forward has no data-path collective)" if use_dist else "none (1 rank)",
                       "frame_sets_rotated": len(frame_sets),
                       "frozen_caches": "lb.freeze(): inference deployment, the packed-weight caches skip their per-call "
                                        "parameter-change scan"},
            "roofline": roof,
        }
        # ---- secondary legs.  Each is printed on ITS OWN earlier line ({"leg": name, ...}); the final line carries short
        # scalar copies of their headline numbers in front of the nested objects and stays under 2 KB, so that a reader who
        # keeps only the top-level keys / the tail of the output still sees every configuration.
        legs = {}
        cfg3_count = None
        if world == 1 and not args.no_cfg3:
            # BASELINE configs[2] (full HSF + IGF forward, batch 2) measured by the same process, after the headline's
            # timed region: a driver-observed number for the second configuration
            if not want_pipelined:
                del lb, out, frame_sets, frames
            torch.cuda.empty_cache()
            c3 = fusion_leg(args, rank, world, dev, 2, args.cfg3_steps, max(3, args.warmup), False)
            legs["cfg3"] = {k: c3[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype",
                                                "config")}
            legs["cfg3"]["stages_ms"] = c3["roofline"].pop("stages_ms")
            legs["cfg3"]["roofline"] = c3["roofline"]
            cfg3_count = c3.pop("_count_launches")
        if world == 1 and not args.no_cfg5:
            # BASELINE configs[4] (0.05 m voxels, 500 k points, f16 storage) and configs[3] (bf16-autocast training step)
            # on this GPU, same process, after the headline: driver-observed numbers for the two remaining configurations
            torch.cuda.empty_cache()
            legs["cfg5_f16"] = f16_stress_leg(args, rank, world, dev)
        if world == 1 and not args.no_cfg4:
            torch.cuda.empty_cache()
            legs["cfg4_train"] = train_leg(args, rank, world, dev)
        if want_pipelined:
            torch.cuda.empty_cache()
            legs["pipelined"] = pipelined_leg(lb, frame_sets, args.steps, max(2, args.warmup // 2), 2, precision=precision,
                                              conv_diag=args.conv_diag, **stage_kw)
        # the CPU baseline after every GPU leg (128 OpenMP threads on the host the launch thread runs on)
        if world == 1 and not args.no_cpu_baseline:
            cdt, n0 = cpu_baseline(args.cpu_points, 1234 + 1000 * CFG_ID)
            import oracle
            cores = oracle.num_threads()
            n0_full = st.num_in[0] / args.batch
            line["cpu_baseline"] = {
                "value": round(1.0 / cdt * (n0 / n0_full), 4), "unit": "frames/s (300k-pt-frame equivalent)",
                "cores": cores, "kind": "port",
                "sample": f"oracle (C port, oracle/isf_oracle.c; the conv loop runs OpenMP over the pairs of a tap "
                          f"on {cores} host threads, the other stages are scalar) on 1 frame of {args.cpu_points} "
                          f"points ({n0} voxels) took {cdt:.2f} s; scaled linearly by level-0 voxels to a "
                          f"{args.points}-pt frame ({int(n0_full)} voxels)",
                "sample_seconds": round(cdt, 2)}
        if cfg3_count is not None:      # after every timed leg (the profiler slows the launches that follow it)
            legs["cfg3"]["launches_per_forward"] = cfg3_count()
        final = compact_line(line, legs)
        print(json.dumps({"leg": "headline_detail", "roofline": line["roofline"], "config": line["config"],
                          "cpu_baseline": line.get("cpu_baseline")}))
        for name, leg in legs.items():
            print(json.dumps(dict({"leg": name}, **leg)))
        sys.stdout.flush()
        print(json.dumps(final))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
