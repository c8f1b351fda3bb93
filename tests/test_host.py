"""CPU tests: the C-ABI library loads and exports every declared symbol (no compute calls), host-side
module logic (state-dict layout, plan flattening, weight-layout conversion), synthetic inputs, and the
multi-process (gloo, world_size 2) paths: sync-BN statistics and bench.py's rank sharding."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from isfusion_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    hdr = open(os.path.join(ROOT, "include", "isf_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(isf_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = _lib.load()  # sets restype/argtypes for every symbol -> AttributeError if one is missing
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.isf_version() > 0
    assert lib.isf_nbr_stride(1) == 128 and lib.isf_nbr_stride(129) == 256
    o = _lib.i3([0, 0, 0])
    assert lib.isf_conv_out_shape(_lib.i3([41, 1440, 1440]), _lib.i3([3, 3, 3]), _lib.i3([2, 2, 2]),
                                  _lib.i3([1, 1, 1]), o) == 0
    assert list(o) == [21, 720, 720]


def test_ops_refuse_cpu_tensors():
    import isfusion_amd as m
    from isfusion_amd._lib import IsfError
    with pytest.raises(IsfError):
        m.voxelization(torch.rand(10, 5), [0.5, 0.5, 0.5], [0, 0, 0, 1, 1, 1], -1, -1)
    with pytest.raises(IsfError):
        m.DynamicScatter([1, 1, 1], [0, 0, 0, 1, 1, 1], True)(torch.rand(4, 3), torch.zeros(4, 3, dtype=torch.int32))


def test_state_dict_keys_match_reference_layout():
    import isfusion_amd as m
    lb = m.LidarBranch()
    keys = set(lb.state_dict().keys())
    # names a released IS-Fusion checkpoint uses
    for k in ["pts_voxel_encoder.vfe_layers.0.linear.weight", "pts_voxel_encoder.vfe_layers.1.norm.running_var",
              "pts_middle_encoder.conv_input.0.weight", "pts_middle_encoder.conv_input.1.running_mean",
              "pts_middle_encoder.encoder_layers.encoder_layer1.0.conv1.weight",
              "pts_middle_encoder.encoder_layers.encoder_layer1.0.bn2.bias",
              "pts_middle_encoder.encoder_layers.encoder_layer1.2.0.weight",
              "pts_middle_encoder.encoder_layers.encoder_layer4.1.conv2.weight",
              "pts_middle_encoder.conv_out.0.weight", "pts_middle_encoder.conv_out.1.weight"]:
        assert k in keys, k
    sd = lb.state_dict()
    assert tuple(sd["pts_middle_encoder.conv_input.0.weight"].shape) == (3, 3, 3, 64, 32)
    assert tuple(sd["pts_middle_encoder.conv_out.0.weight"].shape) == (3, 1, 1, 256, 256)
    assert tuple(sd["pts_voxel_encoder.vfe_layers.1.linear.weight"].shape) == (64, 128)


def test_spconv2_weight_layout_is_converted_on_load():
    import isfusion_amd as m
    conv = m.SparseConv3d(16, 32, 3, stride=2, padding=1, bias=False)
    w1 = torch.randn(3, 3, 3, 16, 32)
    conv.load_state_dict({"weight": w1.permute(4, 0, 1, 2, 3).contiguous()})  # spconv-2 layout [out,k,k,k,in]
    assert torch.equal(conv.weight.data, w1)
    conv.load_state_dict({"weight": w1 * 2})  # native layout passes through
    assert torch.equal(conv.weight.data, w1 * 2)


def test_encoder_plan_flattening():
    import isfusion_amd as m
    enc = m.LidarBranch().pts_middle_encoder
    plan = enc.export_plan()
    L = plan["layers"]
    assert len(L) == 21
    assert [x["kind"] for x in L].count("spconv") == 4
    # conv_input has no residual; block convs: conv1 none, conv2 adds the block input
    assert L[0]["residual_from"] is None and L[1]["residual_from"] is None and L[2]["residual_from"] == 0
    assert L[4]["residual_from"] == 2 and L[7]["residual_from"] == 5
    assert all(x["relu"] for x in L)
    assert L[15]["padding"] == [0, 1, 1] and L[20]["ksize"] == [3, 1, 1] and L[20]["stride"] == [2, 1, 1]
    npy = enc.plan_to_numpy(plan)
    assert npy["layers"][0]["weight"].shape == (3, 3, 3, 64, 32)


def test_bn_fold_equals_eval_batchnorm():
    from isfusion_amd.norm import fold_bn
    bn = torch.nn.BatchNorm1d(8, eps=1e-3)
    with torch.no_grad():
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2)
        bn.weight.normal_()
        bn.bias.normal_()
    bn.eval()
    x = torch.randn(50, 8)
    s, b = fold_bn(bn)
    assert torch.allclose(x * s + b, bn(x), atol=1e-6)


def test_f16x3_numerics():
    """The arithmetic of isf_spconv16.hip restated with torch on the CPU: hi/lo f16 split of both operands
    (weights scaled by a power of two), three products, fp32 accumulation -> same error class as an fp32
    matmul, 1000x better than a single f16/bf16 product."""
    torch.manual_seed(0)
    N, K, C = 256, 16 * 256, 128
    x = torch.relu(torch.randn(N, K))
    w = torch.randn(K, C) * (1.0 / (9 * 256)) ** 0.5
    ref = x.double() @ w.double()
    s = 2.0 ** (13 - int(torch.frexp(w.abs().max())[1]))
    ws = w * s
    xh = x.half().float(); xl = (x - xh).half().float()
    wh = ws.half().float(); wl = (ws - wh).half().float()
    y = (xl @ wh + xh @ wl + xh @ wh) / s
    err3 = (y.double() - ref).abs().max().item()
    err32 = ((x @ w).double() - ref).abs().max().item()
    err1 = ((xh @ wh / s).double() - ref).abs().max().item()
    assert err3 < 4 * err32 + 1e-6 and err3 < 5e-6 and err1 > 100 * err3


def test_synthetic_generator_is_seeded_and_shaped():
    from isfusion_amd import synthetic
    a = synthetic.lidar_sweeps(1234, 20000)
    b = synthetic.lidar_sweeps(1234, 20000)
    assert a.shape == (20000, 5) and a.dtype == np.float32 and np.array_equal(a, b)
    r = synthetic.PC_RANGE
    assert (a[:, 0] > r[0]).all() and (a[:, 0] < r[3]).all() and (a[:, 2] > r[2]).all() and (a[:, 2] < r[5]).all()
    assert not np.array_equal(a, synthetic.lidar_sweeps(1235, 20000))
    u = synthetic.uniform_cloud(1, 1000)
    assert u.shape == (1000, 5)


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
from isfusion_amd.norm import NaiveSyncBatchNorm1d
rank = dist.get_rank()
torch.manual_seed(0)
full = torch.randn(64, 6) * 3 + 1
x = full[rank * 32:(rank + 1) * 32].clone().requires_grad_()
bn = NaiveSyncBatchNorm1d(6, eps=1e-3, momentum=0.01).train()
y = bn(x)
y.square().sum().backward()
ref = torch.nn.BatchNorm1d(6, eps=1e-3, momentum=0.01).train()
xf = full.clone().requires_grad_()
yr = ref(xf)
yr.square().sum().backward()
assert torch.allclose(y, yr[rank * 32:(rank + 1) * 32], atol=1e-5), "forward"
assert torch.allclose(x.grad, xf.grad[rank * 32:(rank + 1) * 32], atol=1e-4), "backward"
assert torch.allclose(bn.running_mean, ref.running_mean, atol=1e-6)
# bench.py sharding helper: every rank gets its own frames, all frames covered exactly once
import bench
sh = [bench.frames_for_rank(r, 2, 4) for r in range(2)]
assert sh[0] != sh[1] and len(set(sh[0]) | set(sh[1])) == 8
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert t.item() == 2.0
dist.destroy_process_group()
print("OK", rank)
"""


def test_multiprocess_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2", PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "OK" in o, o


def test_bench_gpus_2_starts_two_ranks_by_itself():
    """`python bench.py --gpus 2` with no launcher around it must BECOME two ranks (the reference's
    tools/run-nus.sh:11-13 starts its ranks itself), not print an n_gpus-1 line: the gloo rehearsal runs the same
    self-launch / rendezvous / sharding / max-over-ranks clock / one-line path around a stub step."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3",
                        "--warmup", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                       # ONE line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["data"] == "rehearsal"
    pids = line["config"]["rank_pids"]
    assert len(set(pids)) == 2 and os.getpid() not in pids   # two distinct worker processes ran the step
    fr = line["config"]["rank_frames"]
    assert sorted(fr[0] + fr[1]) == list(range(2 * len(fr[0]))) and not set(fr[0]) & set(fr[1])


def test_self_launch_refuses_fewer_devices_and_is_a_noop_under_a_launcher(monkeypatch):
    from isfusion_amd import launch
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert launch.self_launch(1) is None
    if not torch.cuda.is_available():
        with pytest.raises(SystemExit) as e:             # no silent single-GPU run labelled N
            launch.self_launch(2, "nccl")
        assert "only 0 GPU" in str(e.value)
    monkeypatch.setenv("WORLD_SIZE", "2")
    assert launch.self_launch(2) is None                  # already a rank: nothing to do
    with pytest.raises(SystemExit):
        launch.self_launch(4)                             # launcher / flag mismatch is loud
    cmd = launch.launch_command("bench.py", ["--gpus", "8"], 8, 29500)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "127.0.0.1" in cmd


def test_detector_level_state_dict_uses_reference_names():
    """ISFusionPtsPath registers its sub-modules under the reference detector's attribute names (isfusion.py:20-51):
    a released checkpoint's pts_* / fusion_encoder.* keys load unchanged (no wrapper prefixes, no duplicates)."""
    from isfusion_amd.detector import ISFusionPtsPath
    net = ISFusionPtsPath()
    keys = list(net.state_dict().keys())
    prefixes = {k.split(".")[0] for k in keys}
    assert prefixes == {"pts_voxel_encoder", "pts_middle_encoder", "fusion_encoder", "pts_backbone", "pts_neck",
                        "pts_bbox_head"}
    assert len(keys) == len(set(keys))
    for k in ("pts_voxel_encoder.vfe_layers.0.linear.weight", "pts_middle_encoder.conv_input.0.weight",
              "fusion_encoder.grid2region_att.0.block_list.0.encoder_list.0.win_attn.self_attn.in_proj_weight",
              "fusion_encoder.instance_att.layers.1.cross_attn.sampling_offsets.bias",
              "pts_backbone.ds_layer.0.weight", "pts_neck.deblocks.1.0.weight",
              "pts_bbox_head.decoder.0.multihead_attn.in_proj_weight",
              "pts_bbox_head.prediction_heads.0.heatmap.1.bias"):
        assert k in keys, k
    # the transposed conv of the neck keeps torch's [Cin, Cout, k, k] layout
    assert tuple(net.state_dict()["pts_neck.deblocks.1.0.weight"].shape) == (256, 256, 2, 2)


def test_fusion_modules_fail_loudly_without_gpu():
    """no CPU fallback anywhere in the HSF / IGF / head path: CPU tensors raise instead of silently computing"""
    import pytest
    import torch
    from isfusion_amd import _lib
    from isfusion_amd import fusion_ops as ops
    from isfusion_amd.dense_conv import PackedConvBN, SplitMap
    from isfusion_amd.transfusion_head import TransFusionHeadV2
    with pytest.raises(_lib.IsfError):
        ops.PackedLinear(torch.zeros(16, 32))
    with pytest.raises(_lib.IsfError):
        SplitMap.from_nchw(torch.zeros(1, 32, 4, 4))
    with pytest.raises(_lib.IsfError):
        PackedConvBN(torch.nn.Conv2d(32, 32, 3, padding=1, bias=False), None)
    with pytest.raises(_lib.IsfError):
        ops.instance_topk(torch.zeros(1, 10, 8, 8), 4)
    head = TransFusionHeadV2(test_cfg=dict(dataset="nuScenes", grid_size=[64, 64, 40], out_size_factor=8)).eval()
    with pytest.raises(_lib.IsfError):
        head.forward_single(torch.zeros(1, 512, 8, 8))
    head.train()
    with pytest.raises(AssertionError):
        head.forward_single(torch.zeros(1, 512, 8, 8))


def test_seeded_state_dict_keeps_signal_alive():
    """normalisation scales are drawn around 1 for every norm layer (found by module type), so a parity check on the
    last feature map of a conv stack exercises the whole data path"""
    import torch
    from isfusion_amd.fusion_modules import SECONDV2, seeded_state_dict
    bb = SECONDV2(in_channels=128, out_channels=[128, 256], layer_nums=[5, 5], layer_strides=[1, 2])
    sd = seeded_state_dict(bb, 200)
    for name, m in bb.named_modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            w = sd[name + ".weight"]
            assert float(w.min()) >= 0.75 and float(w.max()) <= 1.25, name
    assert seeded_state_dict(bb, 200)["blocks.0.0.weight"].equal(sd["blocks.0.0.weight"])   # deterministic


def test_input_loader_refuses_cpu_and_replays_the_reference_rng():
    """no CPU fallback; draw_train_aug consumes numpy's RNG in the reference's order (same draws as the golden)"""
    import numpy as np
    import pytest
    from input_common import train_aug
    from isfusion_amd import _lib
    from isfusion_amd.input_pipeline import MultiSweepPointLoader, draw_train_aug
    with pytest.raises(_lib.IsfError):
        MultiSweepPointLoader(device="cpu")([dict(pts_filename=np.zeros((4, 5), np.float32), timestamp=0.0)])
    np.random.seed(77)
    mine, ref = draw_train_aug(), train_aug(77)
    assert mine["scale"] == ref["scale"] and np.array_equal(mine["translation"], ref["translation"])
    assert np.array_equal(mine["rot_mat_T"], ref["rot_mat_T"])
    assert (mine["flip_horizontal"], mine["flip_vertical"]) == (ref["flip_horizontal"], ref["flip_vertical"])
    ld = MultiSweepPointLoader(sweeps_num=3, test_mode=True)
    assert list(ld._choose(2)) == [0, 1] and list(ld._choose(7)) == [0, 1, 2]


def test_parameter_changes_invalidate_packed_weight_caches():
    """packed weights / tables are cached per module, keyed on (version, address) of its parameters and buffers: a
    checkpoint loaded after the first forward -- through load_state_dict, through mmcv's load_checkpoint (which
    recurses over _load_from_state_dict and fires no post hooks) or through an in-place copy_ -- must drop them, or the
    new weights would be ignored silently; freeze() opts out of the scan"""
    import torch
    from isfusion_amd import fusion_ops as ops
    from fusion_common import CONFIGS, encoder_kwargs
    from isfusion_amd.fusion_encoder import ISFusionEncoder
    enc = ISFusionEncoder(**encoder_kwargs(CONFIGS["small"]))
    sub = enc.grid2region_att[0]
    dev = torch.device("cpu")
    ops._cache(sub, dev)["linear0"] = "stale"               # what a first forward leaves behind
    assert ops._cache(sub, dev).get("linear0") == "stale"   # the cache survives ordinary calls
    enc.load_state_dict(enc.state_dict())                   # ancestor load (in-place copies bump the versions)
    assert "linear0" not in ops._cache(sub, dev)
    ops._cache(sub, dev)["linear0"] = "stale"
    for name, p in sub.named_parameters():                  # what mmcv's loader does: no hooks, in-place copy
        with torch.no_grad():
            p.copy_(p.detach().clone())
        break
    assert "linear0" not in ops._cache(sub, dev)
    ops.freeze(sub.eval())                                  # an inference-deployment switch: eval mode (drops the caches)
    ops._cache(sub, dev)["linear0"] = "kept"                # packed after the freeze
    with torch.no_grad():
        next(sub.parameters()).mul_(1.0)
    assert ops._cache(sub, dev).get("linear0") == "kept"    # frozen: no scan
    ops.freeze(sub, False)
    assert "linear0" not in ops._cache(sub, dev)


def test_registry_surface_and_build_from_the_unmodified_reference_config():
    """SURVEY 8b level 1: the reference's type names resolve to this build's classes; the point-cloud path builds from
    configs/isfusion/isfusion_0075voxel.py as it stands (authoring container only: the file is not copied)"""
    import os
    import pytest
    from isfusion_amd import registry
    from isfusion_amd.detector import ISFusionPtsPath

    class FakeRegistry:
        def __init__(self):
            self.mods = {}

        def register_module(self, name=None, force=False, module=None):
            assert force and name not in self.mods
            self.mods[name] = module

    regs = {n: FakeRegistry() for n in ("MODELS", "BACKBONES", "NORM_LAYERS")}
    done = registry.register_into(regs)
    assert "MODELS.ISFusionEncoder" in done and "BACKBONES.SECONDV2" in done and "NORM_LAYERS.naiveSyncBN1d" in done
    assert not any(d.startswith("CONV_LAYERS") for d in done)          # registry not offered -> skipped
    assert regs["MODELS"].mods["SparseEncoder"] is registry.lookup("SparseEncoder")
    vfe = registry.build(dict(type="HardSimpleVFE", num_features=5))
    assert type(vfe).__name__ == "HardSimpleVFE" and vfe.num_features == 5
    with pytest.raises(KeyError):
        registry.lookup("CenterHead")
    cfg = "/root/reference/configs/isfusion/isfusion_0075voxel.py"
    if not os.path.exists(cfg):
        pytest.skip("reference tree not present (GPU box)")
    net = registry.build_pts_path(cfg)
    ref = ISFusionPtsPath()
    a = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert a == {k: tuple(v.shape) for k, v in ref.state_dict().items()} and len(a) > 400
    assert net.pts_bbox_head.test_cfg["nms_type"] is None and net.pts_bbox_head.bbox_coder["code_size"] == 10
    assert net.fusion_encoder.num_points_in_pillar == 12 and net.pillar_size == [0.6, 0.6, 8]


def test_neck_gemm_formulation_equals_the_conv_modules():
    """SECONDFPN "hip" path: BN-folded 1x1 conv and 2x2 stride-2 transposed conv as token GEMMs +
    sub-cell interleave -- with a torch GEMM standing in for the HIP linear kernel, against the stock modules"""
    import torch
    from isfusion_amd.fusion_modules import SECONDFPN, seeded_state_dict
    neck = SECONDFPN().eval()
    neck.load_state_dict(seeded_state_dict(neck, 250))
    g = torch.Generator().manual_seed(0)
    x = [torch.randn((2, 128, 12, 12), generator=g), torch.randn((2, 256, 6, 6), generator=g)]

    def linear_relu(t, w, b):
        tok = t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])
        return torch.relu(tok @ w.t() + b)

    neck.dense_conv = "stock"          # the torch modules (the default "hip" path needs the GPU library)
    with torch.no_grad():
        want = neck(x)[0]
        got = neck.forward_tokens(x, linear_relu)[0]
    assert got.shape == want.shape == (2, 512, 12, 12)
    assert (got - want).abs().max().item() < 1e-4


def test_camera_matrix_fold_batched_equals_the_per_camera_chain():
    """fusion_ops.p2g_camera_params folds lidar2img . inverse(lidar_aug) and the image augmentation of every (sample,
    camera) in one batched float64 pass; it must reproduce the per-camera chain of img_point_sampling
    (fusion_encoder.py:1030-1047) written out one camera at a time"""
    import torch
    from isfusion_amd import fusion_ops as ops, synthetic
    inp = synthetic.fusion_inputs(5, 3)
    l2i, ia, la = [torch.from_numpy(inp[k]).double() for k in ("lidar2img", "img_aug_matrix", "lidar_aug_matrix")]
    B, ncam = l2i.shape[:2]
    want = torch.empty((B, ncam, 20), dtype=torch.float64)
    for b in range(B):
        rinv = torch.inverse(la[b, :3, :3])
        for k in range(ncam):
            m = l2i[b, k, :3, :3] @ rinv
            want[b, k, :9] = m.reshape(-1)
            want[b, k, 9:12] = l2i[b, k, :3, 3] - m @ la[b, :3, 3]
            want[b, k, 12:18] = ia[b, k, :2, :3].reshape(-1)
            want[b, k, 18:20] = ia[b, k, :2, 3]
    got = ops.p2g_camera_params(inp["lidar2img"], inp["img_aug_matrix"], inp["lidar_aug_matrix"])
    assert got.shape == (B * ncam, 20) and got.dtype == torch.float32
    assert (got.double() - want.reshape(B * ncam, 20)).abs().max().item() < 1e-6 * max(1.0, want.abs().max().item())


def test_prediction_heads_as_grouped_gemms_equal_the_conv1d_stacks(monkeypatch):
    """TransFusionHeadV2._pack_prediction_heads: the six Conv1d(k=1) + BN + ReLU + Conv1d(k=1) stacks of an FFN as two
    groups (4 + 2 outputs) of [first layers side by side | block-diagonal last layers] GEMMs -- with a torch GEMM
    standing in for the HIP linear kernel, against the module stacks"""
    import torch
    from isfusion_amd import fusion_ops as ops, transfusion_head as th
    from isfusion_amd.fusion_modules import seeded_state_dict

    class PL:
        def __init__(self, w, b=None):
            self.w, self.bias = w.float(), b

    def lin(x, pl, act=0, **kw):
        y = x @ pl.w.t() + (pl.bias if pl.bias is not None else 0)
        return torch.relu(y) if act == ops.ACT_RELU else y

    monkeypatch.setattr(ops, "PackedLinear", PL)
    head = th.TransFusionHeadV2().eval()
    head.load_state_dict(seeded_state_dict(head, 300))
    ffn = head.prediction_heads[0]
    groups = th.TransFusionHeadV2._pack_prediction_heads(ffn)
    assert [len(g["cols"]) for g in groups] == [4, 2]
    assert all(g["l2"].w.shape[0] % 16 == 0 and g["l1"].w.shape[0] in (64, 128, 256) for g in groups)
    B, P, E = 2, 50, 128
    q = torch.randn(B * P, E, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        want = ffn(q.view(B, P, E).transpose(1, 2).contiguous())
        got = {}
        for g in groups:
            o = lin(lin(q, g["l1"], act=ops.ACT_RELU), g["l2"]).view(B, P, -1)
            for n, a, b in g["cols"]:
                got[n] = o[:, :, a:b].transpose(1, 2)
    assert set(got) == set(want)
    for k in want:
        assert (got[k] - want[k]).abs().max().item() < 1e-5, k


def test_training_helpers_chunked_weight_grad_and_packed_conv_io():
    """fusion_train._weight_grad (dW = g^T x as a chunked batched GEMM) equals the plain product for row counts around
    the chunk size; fusion_train.pack_stock_convs makes a stock conv see contiguous inputs and contiguous output
    gradients (the token-major ops around it hand over permuted views)"""
    import torch
    from isfusion_amd import fusion_train as tr
    g = torch.Generator().manual_seed(0)
    for M in (100, 8191, 8192, 20000, 64800):
        gy, x = torch.randn((M, 48), generator=g), torch.randn((M, 32), generator=g)
        want = gy.double().t() @ x.double()
        assert (tr._weight_grad(gy, x).double() - want).abs().max().item() < 2e-3 * max(1.0, want.abs().max().item())
    conv = tr.pack_stock_convs(torch.nn.Conv2d(8, 8, 3, padding=1))
    assert tr.pack_stock_convs(conv) is conv and conv._isf_packed_io          # idempotent
    seen = {}

    def see_input(mod, args):                      # runs after pack_stock_convs' own pre-hook (registration order)
        seen["in_contig"] = args[0].is_contiguous()

    conv.register_forward_pre_hook(see_input)
    x = torch.randn((2, 6, 5, 8), generator=g).permute(0, 3, 1, 2).requires_grad_()     # channels-last storage, NCHW view
    assert not x.is_contiguous()
    conv(x).permute(0, 2, 3, 1).reshape(-1, 8).sum(0).sum().backward()
    assert seen == {"in_contig": True} and x.grad is not None and torch.isfinite(x.grad).all()
    # the gradient that reaches the producer of a guarded tensor is contiguous although the consumer hands back a view
    t = torch.randn((2, 8, 5, 6), generator=g).requires_grad_()
    mid = t * 1.0

    def see_grad(gr):
        seen["grad_contig"] = gr.is_contiguous()

    mid.register_hook(see_grad)
    w = torch.randn((2, 5, 6, 8), generator=g).permute(0, 3, 1, 2)                        # non-contiguous [2, 8, 5, 6]
    (tr._PackedGrad.apply(mid) * 1.0).backward(w)
    assert seen["grad_contig"] is True
    seen.clear()
    mid2 = t * 1.0
    mid2.register_hook(see_grad)
    (mid2 * 1.0).backward(w)
    assert seen["grad_contig"] is False                                                   # what the guard is for


def test_pow2_rescale_keeps_non_finite_entries_local():
    """_lib.pow2_rescale (gradient range shift in front of the f16x3 dX GEMMs): exact power-of-two scale from the largest
    finite entry; one inf no longer zeroes the scale (g * 0 / 0 made the whole dX tensor NaN)"""
    import torch
    from isfusion_amd import _lib
    g = torch.tensor([[1e-7, -3e-6], [2e-9, 5e-8]])
    gs, s = _lib.pow2_rescale(g)
    assert float(torch.log2(s)) == round(float(torch.log2(s))) and 512 <= gs.abs().max().item() <= 1024
    assert torch.equal(gs / s, g)
    g2 = g.clone()
    g2[0, 0] = float("inf")
    gs2, s2 = _lib.pow2_rescale(g2)
    assert torch.isfinite(s2) and s2.item() > 0
    back = gs2 / s2
    assert torch.isinf(back[0, 0]) and torch.equal(back[1], g[1]) and back[0, 1] == g[0, 1]
    gs3, s3 = _lib.pow2_rescale(torch.zeros(3, 3))
    assert torch.isfinite(s3) and torch.equal(gs3, torch.zeros(3, 3))


def test_freeze_ends_when_weights_can_change():
    """fusion_ops.freeze skips the per-call parameter scans of the packed-weight caches; a load_state_dict anywhere below
    the frozen module, or a forward in training mode, must end the skip (stale packed weights survived both)"""
    import torch
    from isfusion_amd import fusion_ops as ops
    net = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Sequential(torch.nn.Linear(4, 4))).eval()
    ops.freeze(net)
    assert all(ops.frozen(m) for m in net.modules())
    c0 = ops._cache(net[0], torch.device("cpu"))
    with torch.no_grad():
        net[0].weight.add_(1.0)
    assert ops._cache(net[0], torch.device("cpu")) is c0            # frozen: the scan is skipped (documented contract)
    net[1].load_state_dict(net[1].state_dict())                     # weights (re)loaded somewhere below the root
    assert not any(ops.frozen(m) for m in net.modules())
    assert ops._cache(net[0], torch.device("cpu")) is not c0        # re-validated by key: the stale cache is dropped
    ops.freeze(net)
    net.train()
    assert not ops.frozen(net[0]) and not ops.frozen(net[0].eval())  # seen in training mode: stays unfrozen after eval()
    import pickle
    pickle.dumps(net)                                               # hooks are module-level functions


def test_freeze_after_a_weight_change_never_reuses_the_old_packed_copies():
    """load_state_dict -> freeze() -> forward with NO forward in between (and train -> step -> eval ->
    freeze) used to leave a cache packed from the old weights in place, and a frozen cache is used without a look at
    the parameters.  Every change of the flag now drops the derived state: caches, C plan, VFE fold, HIP graphs."""
    import torch
    from isfusion_amd import fusion_ops as ops
    dev = torch.device("cpu")
    net = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Linear(4, 4)).eval()
    ops.freeze(net)
    c0 = ops._cache(net[0], dev)
    c0["packed"] = "from the old weights"
    net.load_state_dict({k: v + 1 for k, v in net.state_dict().items()})   # unfreezes (pre-hook) ...
    ops.freeze(net)                                                         # ... and is frozen again at once
    c1 = ops._cache(net[0], dev)
    assert c1 is not c0 and "packed" not in c1
    # the other stale route: weights written in place while unfrozen, then freeze() without a forward in between
    ops.freeze(net, False)
    c2 = ops._cache(net[1], dev)
    c2["packed"] = "old"
    with torch.no_grad():
        net[1].weight.mul_(2.0)
    ops.freeze(net)
    assert "packed" not in ops._cache(net[1], dev)
    # derived state kept outside _cache(): plan / VFE fold / captured graphs are dropped with the flag change
    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(2, 2)
            self._plan, self._vfe_cache, self._frozen = "plan", "vfe", False
            self.__dict__["_graphs"] = {("k",): "captured"}
    h = Holder().eval()
    ops.freeze(h)
    assert h._plan is None and h._vfe_cache is None and h._graphs == {} and h._frozen is True
    h._plan, h._graphs[("k",)] = "plan2", "g2"
    h.lin.load_state_dict(h.lin.state_dict())
    assert h._plan is None and h._graphs == {} and h._frozen is False



def test_layer_level_kernel_choice_mirrors_the_engine(monkeypatch):
    """spconv.sparse_conv_forward_best is what SparseConvFunction (the sparse_conv_ext.indice_conv replacement) calls:
    narrow shapes go to the LDS-DMA gather kernel, wider ones to the split kernel with the rulebook's tile-order table,
    the timing diagnostics to the plain gather kernel -- the same choices isf_sparse_encoder_forward makes per layer."""
    from isfusion_amd import spconv as sp
    calls = []
    monkeypatch.setattr(sp, "sparse_conv_forward_dma", lambda *a, **k: calls.append(("dma", a[3], a[4], a[-1])) or "dma")
    monkeypatch.setattr(sp, "sparse_conv_forward_f16x3",
                        lambda *a, **k: calls.append(("split", a[3], a[4], a[10] if len(a) > 10 else 0,
                                                      k.get("table", a[11] if len(a) > 11 else None))) or "split")
    monkeypatch.setattr(sp, "tile_order", lambda rb, ci, co, mode=0: ("order", ci, co, mode))
    rb = object()
    assert sp.sparse_conv_forward_best(None, None, 27, 64, 64, rb) == "dma"
    assert sp.sparse_conv_forward_best(None, None, 27, 32, 64, rb, mode=257) == "dma"
    assert sp.sparse_conv_forward_best(None, None, 27, 64, 128, rb) == "split"
    assert sp.sparse_conv_forward_best(None, None, 27, 256, 256, rb, mode=1) == "split"
    assert sp.sparse_conv_forward_best(None, None, 27, 64, 64, rb, mode=16) == "split"     # diagnostic: gather kernel
    assert sp.sparse_conv_forward_best(None, None, 27, 256, 256, rb, mode=2) == "split"
    assert calls == [("dma", 64, 64, 0), ("dma", 32, 64, 257),
                     ("split", 64, 128, 0, ("order", 64, 128, 0)), ("split", 256, 256, 1, ("order", 256, 256, 1)),
                     ("split", 64, 64, 16, None), ("split", 256, 256, 2, None)]


def test_h2d_async_cpu_path_and_ring_bookkeeping():
    """fusion_ops.h2d_async: on a CPU target it is a plain copy (nothing to keep ahead of); the pinned ring is keyed by
    (device, shape, dtype) and only ever touched for CUDA targets"""
    from isfusion_amd import fusion_ops as ops
    before = dict(ops._H2D_RING)
    t = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    out = ops.h2d_async(t, "cpu")
    assert torch.equal(out, t) and out.device.type == "cpu"
    assert ops._H2D_RING == before
