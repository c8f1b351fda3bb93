"""Data-parallel training steps of the point-cloud path under torchrun (BASELINE configs[3] shape: B frames per GPU, RCCL
all-reduce over xGMI on the GRADIENTS only -- the forward has no collective):

    python tools/train_step.py --gpus N [--batch 2] [--points 60000] [--steps 3] [--bf16] [--autocast]

starts N ranks by itself (isfusion_amd.launch.self_launch: a re-exec under torch.distributed.run on 127.0.0.1; fewer
than N visible GPUs is an error) -- the reference's tools/run-nus.sh:11-13; under a launcher it runs as the rank it is:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29540 \
        tools/train_step.py --gpus N [...]

One process per GPU; every rank draws its own synthetic frames.  The module is wrapped in DistributedDataParallel
(bucketed gradient all-reduce overlapped with the backward); the path's BatchNorm layers are the config's:
isfusion_amd.norm.NaiveSyncBatchNorm where the reference uses naiveSyncBN (cross-rank statistics, one all_reduce of
[2C] per layer, whenever the world size is > 1), plain BatchNorm (local-shard statistics) elsewhere.  The loss is a stand-in (feature energy + heat-map mean): the detection losses and
target assignment are the reference's training control plane.  Prints one JSON line per rank 0 with ms per step.
Also runs on a single GPU without torchrun (world size 1)."""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="ranks to start (0 = whatever the launcher started, else 1)")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--points", type=int, default=60000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--bf16", action="store_true", help="camera features in bfloat16 (the reference's autocast dtype)")
    ap.add_argument("--autocast", action="store_true",
                    help="run forward + loss under torch.autocast(bfloat16): stock convs / linears in bf16, the HIP "
                         "autograd Functions cast their inputs to fp32")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real thing); gloo with --shared-device = a 1-GPU REHEARSAL of the "
                         "multi-rank step (DDP's bucketed all-reduce and the sync-BN exchange go through gloo)")
    ap.add_argument("--shared-device", action="store_true",
                    help="every rank uses cuda:0 (1-GPU box): proves that several processes of libisf_hip.so train side "
                         "by side; with --backend gloo")
    ap.add_argument("--stock-dense", action="store_true",
                    help="the dense 3x3 conv + BatchNorm stacks on the stock modules (MIOpen) instead of dense_train.py (A/B)")
    a = ap.parse_args()
    from isfusion_amd import launch, synthetic
    if a.stock_dense:
        from isfusion_amd import dense_train
        dense_train.ENABLED = False
    if a.gpus > 0:
        launch.self_launch(a.gpus, a.backend)
    from isfusion_amd.detector import ISFusionPtsPath
    from isfusion_amd.fusion_modules import seeded_state_dict
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if a.shared_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29540")
    dist.init_process_group(a.backend, rank=rank, world_size=world)  # "nccl" IS RCCL on ROCm
    net = ISFusionPtsPath().train()
    net._lidar.randomize_weights_(0).randomize_bn_(1)
    for mod, seed in ((net.fusion_encoder, 100), (net.pts_backbone, 200), (net.pts_neck, 250)):
        mod.load_state_dict(seeded_state_dict(mod, seed))
    for p in net.pts_bbox_head.parameters():
        p.requires_grad_(False)                                      # head losses: control plane, not exercised here
    net = net.to(dev)

    class Wrap(torch.nn.Module):                                     # DDP hooks forward(); the path's entry is a method
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, pts, img, metas, kw):
            return self.m.forward_train_pts(pts, img, metas, **kw)

    # gradient_as_bucket_view: the gradients ARE views of the all-reduce buckets -- no per-parameter copy into a bucket
    # after the backward pass (~300 copy launches per step); every parameter that requires a gradient gets one in the
    # step, so no unused-parameter search either
    ddp = torch.nn.parallel.DistributedDataParallel(Wrap(net), device_ids=[local], gradient_as_bucket_view=True)
    opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=1e-4, momentum=0.9)
    pts = [torch.from_numpy(synthetic.lidar_sweeps(9000 + 100 * rank + i, a.points)).to(dev) for i in range(a.batch)]
    inp = synthetic.fusion_inputs(7 + rank, a.batch)
    img = tuple(torch.from_numpy(x).to(dev).to(torch.bfloat16 if a.bf16 else torch.float32) for x in inp["img_feats"])
    kw = dict(lidar2img=torch.from_numpy(inp["lidar2img"]), img_aug_matrix=torch.from_numpy(inp["img_aug_matrix"]),
              lidar_aug_matrix=torch.from_numpy(inp["lidar_aug_matrix"]))
    metas = [dict(input_shape=inp["input_shape"]) for _ in range(a.batch)]
    losses, t0 = [], None
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 2)]   # per-step GPU time stamps (no extra sync)
    for step in range(a.steps + 1):
        marks[step].record()
        if step == 1:
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=a.autocast):
            out, hm = ddp(pts, img, metas, kw)
            loss = (out[0].float() ** 2).mean() + hm.float().sigmoid().mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()                                              # bucketed RCCL all-reduce inside
        opt.step()
        losses.append(float(loss))
    marks[a.steps + 1].record()
    torch.cuda.synchronize()
    dist.barrier()
    dt = (time.perf_counter() - t0) / max(a.steps, 1)
    per_step = [round(marks[i].elapsed_time(marks[i + 1]), 1) for i in range(a.steps + 1)]   # [0] = the warm-up step
    if rank == 0:
        print(json.dumps({"world_size": world, "n_gpus": world, "parallelism": f"dp{world}", "rccl": launch.rccl_version(), "backend": a.backend, "shared_device": a.shared_device, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), "batch_per_gpu": a.batch, "points": a.points, "bf16_camera_features": a.bf16, "autocast_bf16": a.autocast,
                          "ms_per_train_step": round(dt * 1e3, 2), "ms_each_step_gpu_clock": per_step, "losses": [round(v, 5) for v in losses]}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
