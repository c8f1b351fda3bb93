"""The point-cloud half of ``ISFusionDetector`` (mmdet3d/models/detectors/isfusion.py:83-121, 148-176): everything
between the raw sweeps (+ camera feature maps from the stock image backbone / neck) and the multi-scale BEV features
the detection neck / head consume.

    ISFusionPtsPath.extract_pts_feat(pts, img_feats, img_metas, lidar2img=..., img_aug_matrix=..., lidar_aug_matrix=...)
        -> [f1 [B,128,180,180], f2 [B,256,90,90]]   (+ the instance heat-map when ``return_heatmap``)

Same method names (``dynamic_voxelize`` is folded into the one-call LiDAR branch, ``voxelize(points, 'pillar')``,
``isfusion``, ``extract_pts_feat``), same sub-module attribute names (``pts_voxel_encoder``, ``pts_middle_encoder``,
``pts_pillar_layer``, ``fusion_encoder``, ``pts_backbone``) so a released checkpoint's ``pts_*`` / ``fusion_encoder.*``
keys load unchanged.  The image backbone / necks and the bbox head stay stock (out of scope).
"""
import torch

from . import _lib
from torch import nn

from .fusion_encoder import ISFusionEncoder
from .fusion_modules import SECONDFPN, SECONDV2
from .lidar_branch import ISFUSION_0075, LidarBranch
from .transfusion_head import TransFusionHeadV2
from .voxelize import Voxelization

ISFUSION_0075_FUSION = dict(
    out_size_factor=8,
    fusion_encoder=dict(num_points_in_pillar=12, embed_dims=256, bev_size=180, num_views=6,
                        region_shape=[(6, 6, 1), (6, 6, 1)], grid_size=[[180, 180, 1], [90, 90, 1]],
                        region_drop_info=[{0: {"max_tokens": 36, "drop_range": (0, 100000)}},
                                          {0: {"max_tokens": 36, "drop_range": (0, 100000)}}],
                        instance_num=200),
    pts_backbone=dict(in_channels=128, out_channels=[128, 256], layer_nums=[5, 5], layer_strides=[1, 2],
                      norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01), conv_cfg=dict(type="Conv2d", bias=False)),
    pts_neck=dict(in_channels=[128, 256], out_channels=[256, 256], upsample_strides=[1, 2],
                  norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01), upsample_cfg=dict(type="deconv", bias=False),
                  use_conv_for_no_stride=True),
    pts_bbox_head=dict(num_proposals=200, auxiliary=True, in_channels=512, hidden_channel=128, num_classes=10,
                       num_decoder_layers=1, num_heads=8, nms_kernel_size=3, ffn_channel=256,
                       common_heads=dict(center=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2)),
                       test_cfg=dict(dataset="nuScenes", grid_size=[1440, 1440, 40], out_size_factor=8)),
)


class ISFusionPtsPath(nn.Module):

    def __init__(self, voxel_size=None, pc_range=None, pts_voxel_encoder=None, pts_middle_encoder=None,
                 fusion_encoder=None, pts_backbone=None, out_size_factor=None, pts_neck=None, pts_bbox_head=None):
        super().__init__()
        cfg = ISFUSION_0075
        self.voxel_size = list(voxel_size or cfg["voxel_size"])
        self.pc_range = list(pc_range or cfg["point_cloud_range"])
        lidar = LidarBranch(self.voxel_size, self.pc_range, pts_voxel_encoder, pts_middle_encoder)
        # the one-call engine is kept OUT of the module tree; its two sub-modules are registered here under the
        # reference's attribute names (shared objects), so state-dict keys are pts_voxel_encoder.* / pts_middle_encoder.*
        object.__setattr__(self, "_lidar", lidar)
        self.pts_voxel_encoder = lidar.pts_voxel_encoder
        self.pts_middle_encoder = lidar.pts_middle_encoder
        fe = dict(fusion_encoder or ISFUSION_0075_FUSION["fusion_encoder"])
        fe.pop("type", None)
        self.fusion_encoder = ISFusionEncoder(**fe)
        bb = dict(pts_backbone or ISFUSION_0075_FUSION["pts_backbone"])
        bb.pop("type", None)
        self.pts_backbone = SECONDV2(**bb)
        nk = dict(pts_neck or ISFUSION_0075_FUSION["pts_neck"])
        nk.pop("type", None)
        self.pts_neck = SECONDFPN(**nk)
        hd = dict(pts_bbox_head or ISFUSION_0075_FUSION["pts_bbox_head"])
        for k in ("type", "loss_cls", "loss_bbox", "loss_heatmap", "loss_iou", "dropout", "bn_momentum", "activation"):
            hd.pop(k, None)   # losses / training-only knobs: control plane
        self.pts_bbox_head = TransFusionHeadV2(**hd)
        osf = out_size_factor or ISFUSION_0075_FUSION["out_size_factor"]
        # isfusion.py:45-51
        self.pillar_size = [self.voxel_size[0] * osf, self.voxel_size[1] * osf, self.pc_range[5] - self.pc_range[2]]
        self.pts_pillar_layer = Voxelization(max_num_points=self.fusion_encoder.num_points_in_pillar,
                                             voxel_size=self.pillar_size, max_voxels=(30000, 60000),
                                             point_cloud_range=self.pc_range)

    def freeze(self, flag=True):
        """Inference deployment (weights static): every module below skips its per-call "did a parameter change?" scan
        of the packed-weight caches -- about 0.4 ms of host time per forward, which at small batch is GPU idle time."""
        from . import fusion_ops as ops
        self._lidar.freeze(flag)
        ops.freeze(self, flag)
        return self

    def train(self, mode=True):
        """Entering training mode ends the freeze (optimizer steps are about to change the weights): the packed caches
        and the captured HIP graphs below this module are dropped here, not at the next forward that happens to look."""
        if mode and self.__dict__.get("_isf_frozen", False):
            self.freeze(False)
        return super().train(mode)

    @torch.no_grad()
    def voxelize(self, points, voxel_type="pillar"):
        """isfusion.py:148-176 (pillar branch): per-sample hard voxelization, batch index prepended."""
        assert voxel_type == "pillar", "the fine grid is voxelized dynamically inside the LiDAR branch"
        voxels, coors, num_points = [], [], []
        for i, res in enumerate(points):
            v, c, n = self.pts_pillar_layer(res)
            voxels.append(v)
            num_points.append(n)
            coors.append(torch.nn.functional.pad(c, (1, 0), mode="constant", value=i))
        return torch.cat(voxels, 0), torch.cat(num_points, 0), torch.cat(coors, 0)

    @torch.no_grad()
    def voxelize_async(self, points):
        """The pillar voxelization of every sample QUEUED on the current stream, no host wait (device-resident counts:
        isf_hard_voxelize_device) -> a callable that returns voxelize()'s triple; call it after other host work."""
        if 1 < len(points) <= 16 and self.__dict__.get("batched_pillars", True):
            # round 6: one pass for the whole batch (isf_hard_voxelize_batched_device): half the launches of two samples,
            # the concatenation and the sample column written by the kernel
            return self.pts_pillar_layer.forward_batch_async(points).result
        pend = [self.pts_pillar_layer.forward_async(res) for res in points]

        def finish():
            voxels, coors, num_points = [], [], []
            for i, p in enumerate(pend):
                v, c, n = p.result()
                voxels.append(v)
                num_points.append(n)
                coors.append(torch.nn.functional.pad(c, (1, 0), mode="constant", value=i))
            return torch.cat(voxels, 0), torch.cat(num_points, 0), torch.cat(coors, 0)
        return finish

    def isfusion(self, pts, pts_feats, img_feats, img_metas, batch_size, pillars=None, **kwargs):
        """isfusion.py:83-101.  pillars: a (pillars, num_points, coors) triple voxelized ahead of time
        (extract_pts_feat does it on a side stream while the LiDAR branch runs)."""
        pillars, pillars_num_points, pillar_coors = pillars or self.voxelize(pts, voxel_type="pillar")
        pts_metas = dict(pillars=pillars, pillars_num_points=pillars_num_points, pillar_coors=pillar_coors, pts=pts,
                         pillar_size=self.pillar_size)
        kwargs.update(dict(pts_metas=pts_metas, img_metas=img_metas, pts_backbone=self.pts_backbone))
        return self.fusion_encoder(img_feats, pts_feats, batch_size, **kwargs)

    @torch.no_grad()
    def extract_pts_feat(self, pts, img_feats, img_metas, return_heatmap=False, **kwargs):
        """isfusion.py:103-121 without the neck: dynamic voxelize + DynamicVFE + SparseEncoder (one C call), pillar
        voxelization, ISFusionEncoder with the SECONDV2 stages."""
        assert not self.training, "inference path (eval mode)"
        self._lidar.train(False)
        if "p2g_cam" not in kwargs and all(k in kwargs for k in ("lidar2img", "img_aug_matrix", "lidar_aug_matrix")):
            # host-side fold of the camera matrices BEFORE anything is queued: it overlaps nothing otherwise
            from . import fusion_ops as ops
            kwargs["p2g_cam"] = ops.h2d_async(ops.p2g_camera_params(kwargs["lidar2img"], kwargs["img_aug_matrix"],
                                                                    kwargs["lidar_aug_matrix"]), pts[0].device)
        # The pillar voxelization has a host round trip per sample (the voxel count sizes its outputs).  Issued after the
        # LiDAR branch on the same stream, each of them waits for the whole branch and then leaves the GPU idle until
        # the host has launched the next piece (tools/timeline_gaps.py: ~0.25 ms per forward).  On a side stream --
        # the library's workspaces are per (device, stream) -- they wait for the pillar kernels only, while the
        # branch's convolutions keep the GPU busy.
        main = torch.cuda.current_stream()
        side = self.__dict__.setdefault("_side_streams", {}).setdefault(pts[0].device, None)
        if side is None:
            side = self._side_streams[pts[0].device] = torch.cuda.Stream(device=pts[0].device)
        side.wait_stream(main)                       # the points are ready on the main stream
        # Round 6: the pillar kernels are queued FIRST, on the side stream, with their counts left on the device
        # (voxelize_async); the LiDAR branch -- whose own host waits keep this thread busy for most of its duration -- is
        # queued next, and only then are the pillar counts looked at: they arrived long ago, nothing waits, and the
        # 270-450 us the GPU used to idle per forward between the branch and Point-to-Grid are gone
        # (profiles/r05_v2_timeline_gaps_cfg3.txt: copyBuffer -> copyBuffer).
        with torch.cuda.stream(side):
            finish = self.voxelize_async(pts)
        # the BEV map leaves the branch as split-format token matrices when conv_fusion reads exactly that (no fp32 map, no
        # NCHW -> split passes: -2 launches, -0.1 ms per forward at B = 2)
        cd = self._lidar.pts_middle_encoder.out_channels_and_shape()[0]
        split = self._split_handover(img_feats[1].shape[1], cd)
        x = self._lidar(pts, bev_split=split)
        img_bev = None
        with torch.cuda.stream(side):
            pil = finish()                           # slices + concatenation, on the stream that produced them
            if "p2g_cam" in kwargs and kwargs.get("p2g_out") is None:
                # Point-to-Grid needs the pillars and the camera features only: on the side stream too, under the tail
                # of the LiDAR branch (its last launches are still running when the host gets here)
                img_bev = self.fusion_encoder.img_fv_to_bev(
                    [img_feats[1]], len(pts), pts_metas=dict(pillars=pil[0], pillar_coors=pil[2]), img_metas=img_metas,
                    p2g_split=split, **kwargs)       # (split rows: what conv_fusion reads, no NCHW -> split pass)
        main.wait_stream(side)
        for t in pil + ((getattr(img_bev, "data", img_bev),) if img_bev is not None else ()):
            t.record_stream(main)                    # allocated on the side stream, consumed on the main one
        feats, ins_heatmap = self.isfusion(pts, x, img_feats, img_metas, len(pts), pillars=pil, img_bev=img_bev, **kwargs)
        return (feats, ins_heatmap) if return_heatmap else feats

    def _forward_pts_graph(self, pts, img_feats, img_metas, **kwargs):
        """extract_pts_feat's eager head (LiDAR branch || pillar voxelization -> Point-to-Grid) writing into the input
        buffers of the captured tail, then one graph launch.

        Never on the legacy NULL stream.  With torch's default stream (HIP's legacy NULL stream) as the launch stream,
        pillar tensors produced on a side stream and read by the Point-to-Grid launch that precedes hipGraphLaunch ended
        4 of 5 runs in a GPU memory fault at the fourth unsynchronised forward -- at addresses gigabytes away from every
        allocation of the library, of torch and of the graph, not reproducible with serialised launches
        (AMD_SERIALIZE_KERNEL=3), with a copy of the tensors in between, or -- 0 of 6 runs -- with ANY non-NULL stream as
        the launch stream (tools/graph_fault.py, profiles/r04_graph_fault.txt).  So a caller on the
        NULL stream is moved onto a private launch stream for the duration of the forward (both directions ordered by
        events); callers on their own streams run where they are."""
        cur = torch.cuda.current_stream()
        if cur.cuda_stream != 0 or self.__dict__.get("_graph_on_null_stream", False):
            return self._forward_pts_graph_on_stream(pts, img_feats, img_metas, **kwargs)
        dev = pts[0].device
        ls = self.__dict__.setdefault("_launch_streams", {}).get(dev)
        if ls is None:
            ls = self._launch_streams[dev] = torch.cuda.Stream(device=dev)
        ls.wait_stream(cur)          # the inputs, and the previous forward's consumers of the graph's output buffers
        with torch.cuda.stream(ls):
            out = self._forward_pts_graph_on_stream(pts, img_feats, img_metas, **kwargs)
        cur.wait_stream(ls)
        return out

    def _forward_pts_graph_on_stream(self, pts, img_feats, img_metas, **kwargs):
        from . import fusion_ops as ops
        assert not self.training, "inference path (eval mode)"
        self._lidar.train(False)
        dev = pts[0].device
        B = len(pts)
        c_lidar = self._lidar.pts_middle_encoder.out_channels_and_shape()[0]
        g, img_bev, x, out = self._graph_for(B, dev, img_feats[1].shape[1], c_lidar)
        split = self._split_handover(img_feats[1].shape[1], c_lidar)
        cam = kwargs.get("p2g_cam")
        if cam is None:
            cam = ops.h2d_async(ops.p2g_camera_params(kwargs["lidar2img"], kwargs["img_aug_matrix"],
                                                      kwargs["lidar_aug_matrix"]), dev)
        main = torch.cuda.current_stream()
        if self.__dict__.get("_graph_pillar_side", False):
            # as in extract_pts_feat: the pillar voxelization's per-sample host round trips wait for the pillar kernels
            # only, on a side stream, while the LiDAR branch keeps the GPU busy on the launch stream
            side = self.__dict__.setdefault("_side_streams", {}).setdefault(dev, None)
            if side is None:
                side = self._side_streams[dev] = torch.cuda.Stream(device=dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                finish = self.voxelize_async(pts)    # queued first, counts on the device (see extract_pts_feat)
            self._lidar(pts, out=x, bev_split=split)
            with torch.cuda.stream(side):
                pil = finish()
            main.wait_stream(side)
            for t in pil:
                t.record_stream(main)
        else:
            finish = self.voxelize_async(pts)        # same stream, in front of the branch: its counts are in host memory
            self._lidar(pts, out=x, bev_split=split)  # long before the branch's own host waits are over
            pil = finish()
        ops.p2g_sample(pil[0], pil[2], img_feats[1], None, None, None, img_metas[0]["input_shape"], B,
                       self.fusion_encoder.bev_size, self.fusion_encoder.num_views, cam=cam, out=img_bev, split=split)
        g.replay()
        return out

    def forward_train_pts(self, pts, img_feats, img_metas, **kwargs):
        """training mode: extract_pts_feat + pts_neck WITH gradients (towards every parameter of the
        LiDAR branch, the fusion encoder, the backbone stages and the neck, and towards the camera feature maps).  The
        head's losses / target assignment are the reference's training control plane (out of scope): apply them to the
        returned neck output."""
        assert self.training, "call .train() first (eval mode runs the inference engine)"
        from .fusion_train import pack_stock_convs
        pack_stock_convs(self)
        self._lidar.train(True)
        x = self._lidar(pts)
        feats, ins_heatmap = self.isfusion(pts, x, img_feats, img_metas, len(pts), **kwargs)
        return self.pts_neck(feats), ins_heatmap

    # ------------------------------------------------------------------------------------------- HIP graph
    def enable_graph(self, flag=True, pillar_side_stream=False, on_null_stream=False):
        """Inference deployment: `forward_pts` replays everything behind Point-to-Grid -- conv_fusion, Grid-to-Region,
        instance fusion, SECONDV2 stages, neck, head: ~330 launches whose shapes depend on the batch size only -- as ONE
        HIP graph per batch size (captured on the first call), so the host no longer paces those launches (the GPU was
        idle 10-14 % of a step between them).  The LiDAR branch, the pillar voxelization and Point-to-Grid (their sizes
        depend on the frame) stay eager and write straight into the graph's input buffers.  Weights must be final
        (freeze()); freeze() / a load_state_dict below this module / train() drop the captured graphs together with the
        packed-weight caches their kernels point into (fusion_ops.drop_caches)."""
        self.__dict__["_graph_on"] = bool(flag)
        # pillar_side_stream: the pillar voxelization beside the LiDAR branch on a side stream.  OFF by default in graph
        # mode since round 5: the round-3 / round-4 memory fault needed exactly that hand-over (side-stream
        # tensors read by the launch that precedes hipGraphLaunch) and its cause is narrowed (legacy NULL stream), not
        # proven; the eager path keeps the side stream (never faulted, tests/test_gpu_e2e.py soak).  on_null_stream=True
        # lets the forward run on the legacy NULL stream (the faulting set-up; tools/graph_fault.py)
        self.__dict__["_graph_pillar_side"] = bool(pillar_side_stream)
        self.__dict__["_graph_on_null_stream"] = bool(on_null_stream)
        self.__dict__["_graphs"] = {}
        return self

    def _tail(self, img_bev, x, bs):
        enc = self.fusion_encoder
        feats, _ = enc.forward_tail(img_bev, x, bs, pts_backbone=self.pts_backbone, feats_split=True)
        return self.pts_bbox_head.forward_split(self.pts_neck.forward_split(feats))

    def _graph_for(self, bs, dev, c_img, c_lidar):
        from . import fusion_ops as ops
        if not ops.frozen(self):            # weights may have changed: captured kernels hold the old packed copies
            self._graphs.clear()
            raise _lib.IsfError("ISFusionPtsPath.enable_graph: call freeze() after the weights are final (eval mode)")
        key = (bs, str(dev))
        if key in self._graphs:
            return self._graphs[key]
        S = self.fusion_encoder.bev_size
        main = torch.cuda.current_stream()
        cap = torch.cuda.Stream(device=dev)          # private capture stream: its library workspace is never used eagerly
        img_bev = torch.zeros((bs, c_img, S, S), dtype=torch.float32, device=dev)
        x = torch.zeros((bs, c_lidar, S, S), dtype=torch.float32, device=dev)
        # the graph's inputs in the form conv_fusion reads (as in extract_pts_feat): the same buffers seen as split-format
        # token matrices, filled by isf_p2g_forward_split / isf_encoder_options.bev_format = 1
        tail_in = (img_bev, x)
        if self._split_handover(c_img, c_lidar):
            from .dense_conv import SplitMap
            step = bs * S * S * 256 * 4
            raw = x.view(torch.uint8).view(-1)
            tail_in = (SplitMap(img_bev.view(torch.uint8).view(-1), bs, c_img, S, S),
                       [SplitMap(raw[i * step:(i + 1) * step], bs, 256, S, S) for i in range(c_lidar // 256)])
        cap.wait_stream(main)
        with torch.cuda.stream(cap):
            for _ in range(3):                       # warm-up on the capture stream: workspace, packed caches, tables
                self._tail(tail_in[0], tail_in[1], bs)
        cap.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            out = self._tail(tail_in[0], tail_in[1], bs)
        main.wait_stream(cap)
        self._graphs[key] = (g, tail_in[0], x, out)
        return self._graphs[key]

    def _split_handover(self, c_img, c_lidar):
        """do Point-to-Grid and the LiDAR branch hand their maps to conv_fusion as split-format token matrices?"""
        return getattr(self.fusion_encoder, "dense_conv", "") == "hip" and c_img == 256 and c_lidar % 256 == 0

    @torch.no_grad()
    def forward_pts(self, pts, img_feats, img_metas, **kwargs):
        """extract_pts_feat -> pts_neck -> pts_bbox_head (mvx_two_stage.py simple_test_pts without box decoding):
        the raw head outputs [[dict(center, height, dim, rot, vel, heatmap, query_heatmap_score, dense_heatmap)]].
        With enable_graph(): the outputs live in the graph's static buffers (valid until the next call)."""
        if self.__dict__.get("_graph_on", False) and self.pts_neck.dense_conv == "hip" and \
                self.pts_bbox_head.dense_conv == "hip" and self.fusion_encoder.dense_conv == "hip":
            return (self._forward_pts_graph(pts, img_feats, img_metas, **kwargs),)
        engine = self.pts_neck.dense_conv == "hip" and self.pts_bbox_head.dense_conv == "hip" and \
            getattr(self.pts_backbone, "dense_conv", "") == "hip"
        feats = self.extract_pts_feat(pts, img_feats, img_metas, feats_split=engine, **kwargs)
        if self.pts_neck.dense_conv == "hip" and self.pts_bbox_head.dense_conv == "hip":
            # engine-level hand-over: the neck's levels stay split-format token matrices (no [B, 512, H, W] tensor, no
            # permute copy, no NCHW -> split pass); the public neck / head forwards keep the reference's tensors
            return (self.pts_bbox_head.forward_split(self.pts_neck.forward_split(feats)),)
        return self.pts_bbox_head(self.pts_neck(feats), img_feats, img_metas)

    @torch.no_grad()
    def simple_test_pts(self, x, x_img, img_metas, rescale=False):
        """isfusion.py:274-283: head forward + get_bboxes + bbox3d2result (core/bbox/transforms.py:50-76) on the neck
        output -> [dict(boxes_3d, scores_3d, labels_3d)] on the CPU, one per sample."""
        outs = self.pts_bbox_head(x, x_img, img_metas)
        out = []
        for boxes, scores, labels in self.pts_bbox_head.get_bboxes(outs, img_metas, rescale=rescale):
            out.append(dict(boxes_3d=boxes.to("cpu"), scores_3d=scores.cpu(), labels_3d=labels.cpu()))
        return out

    @torch.no_grad()
    def simple_test(self, points, img_metas, img_feats, rescale=False, **kwargs):
        """isfusion.py:285-305 for the point-cloud branch; `img_feats` are the camera neck outputs (the camera
        backbone stays stock).  -> [dict(pts_bbox=...)] per sample."""
        x = self.pts_neck(self.extract_pts_feat(points, img_feats, img_metas, **kwargs))
        return [dict(pts_bbox=r) for r in self.simple_test_pts(x, img_feats, img_metas, rescale=rescale)]
