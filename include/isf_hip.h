/*
 * isf_hip.h -- C ABI of libisf_hip.so: the MI355X (gfx950) implementation of IS-Fusion's
 * LiDAR voxelization -> sparse-conv -> BEV hot path (BASELINE.json north_star).
 *
 * Conventions
 *   - extern "C", plain pointers + sizes, no torch/ATen types.  All data pointers are DEVICE pointers
 *     unless the parameter name ends in `_host`.  All tensors are dense, row-major, contiguous.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls are asynchronous on
 *     that stream, except where an output count is returned through a `*_host` pointer: those calls
 *     synchronise the stream once before returning (the reference ops do the same, see each entry).
 *   - Return value: ISF_OK (0) or a negative ISF_ERR_* code; never throws.  isf_last_error() returns a
 *     thread-local message for the last failure.
 *   - Buffers are caller-owned.  Scratch memory (bump arena, side stream, event pool) is one object per
 *     (device, stream) owned by the library, grown on demand with hipMalloc and released by
 *     isf_release_workspace(): calls on different streams are independent; one call at a time per stream.
 *   - No process-global options and no environment switches: precision / diagnostics travel with the call.
 *   - Voxel coordinates are int32 (z, y, x) or (b, z, y, x) exactly like the reference.
 *
 * Each entry cites the reference interface it replaces (paths relative to the reference root).
 */
#ifndef ISF_HIP_H
#define ISF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISF_OK 0
#define ISF_ERR_ARG (-1)         /* bad argument (null pointer, negative size, unsupported shape) */
#define ISF_ERR_HIP (-2)         /* a HIP runtime call failed */
#define ISF_ERR_NOMEM (-3)       /* workspace allocation failed */
#define ISF_ERR_UNSUPPORTED (-4) /* valid request outside what this build implements */
#define ISF_ERR_CAPACITY (-5)    /* caller-provided output capacity too small */

#define ISF_REDUCE_SUM 0
#define ISF_REDUCE_MEAN 1
#define ISF_REDUCE_MAX 2

#define ISF_CONV_SUBM 0   /* SubMConv3d  */
#define ISF_CONV_SPARSE 1 /* SparseConv3d */

typedef void* isf_stream_t; /* hipStream_t */

/* library / runtime ----------------------------------------------------------------------------- */
int isf_version(void);                 /* (major<<16)|(minor<<8)|patch */
const char* isf_last_error(void);      /* message of the last failure on this thread ("" if none) */
int isf_device_count(int* count_host); /* number of visible HIP devices (0 on a CPU-only host) */
int isf_release_workspace(void);       /* free every (device, stream) workspace */
int isf_workspace_bytes(size_t* bytes_host); /* current arena size on the current device */
/* DIAGNOSTIC: every device allocation behind the workspaces of the current device: stream_base_bytes [max_triples][3] =
 * (stream handle, base address, bytes) -- per workspace its blocks, then its two persistent byte maps; *num_triples = how
 * many exist.  tools/graph_fault.py maps a GPU memory-fault address onto them. */
int isf_debug_workspace_blocks(unsigned long long* stream_base_bytes, int max_triples, int* num_triples);

/* A1  dynamic voxelization ------------------------------------------------------------------------
 * replaces voxel_layer.dynamic_voxelize(points, coors, voxel_size, coors_range, NDim=3)
 *   mmdet3d/ops/voxel/src/voxelization.h:83-95, voxelization_cuda.cu:24-61,485-528; caller
 *   mmdet3d/ops/voxel/voxelize.py:52-55.
 * points [P, C>=3] fp32 -> coors [P,3] int32 (z,y,x); rows with any axis outside the grid become
 * (-1,-1,-1) (CPU-path semantics, voxelization_cpu.cpp:8-43).  fp32 arithmetic: floor((p-min)/vs).
 * Asynchronous; no device-wide sync (the reference calls cudaDeviceSynchronize, :524). */
int isf_dynamic_voxelize(const float* points, int num_points, int num_features,
                         const float voxel_size_host[3], const float coors_range_host[6],
                         int32_t* coors, isf_stream_t stream);

/* Batched form used by ISFusionDetector.dynamic_voxelize (detectors/isfusion.py:123-146): samples
 * are concatenated, point_offsets_host[b]..[b+1] delimits sample b; writes coors4 [P,4] (b,z,y,x). */
int isf_dynamic_voxelize_batched(const float* points, const int64_t* point_offsets_host,
                                 int batch_size, int num_features, const float voxel_size_host[3],
                                 const float coors_range_host[6], int32_t* coors4,
                                 isf_stream_t stream);

/* A2  hard (deterministic) voxelization -----------------------------------------------------------
 * replaces voxel_layer.hard_voxelize(points, voxels, coors, num_points_per_voxel, voxel_size,
 *   coors_range, max_points, max_voxels, NDim=3, deterministic=True) -> int voxel_num
 *   voxelization.h:58-81, voxelization_cpu.cpp:45-144, voxelization_cuda.cu:231-373;
 *   caller voxelize.py:56-70 (outputs pre-allocated at max size and ZERO-FILLED by the caller).
 * Voxels in first-appearance order of the points, first max_points points of each in point order,
 * voxels beyond max_voxels dropped.  Writes the prefix [0, voxel_num) of voxels [max_voxels,
 * max_points, C], coors [max_voxels,3], num_points_per_voxel [max_voxels].  Synchronises once to
 * return *voxel_num_host (the reference returns the count to Python as well). */
int isf_hard_voxelize(const float* points, int num_points, int num_features,
                      const float voxel_size_host[3], const float coors_range_host[6],
                      int max_points, int max_voxels, float* voxels, int32_t* coors,
                      int32_t* num_points_per_voxel, int* voxel_num_host, isf_stream_t stream);
/* The same voxelization WITHOUT the host read-back (round 6: device-resident count): the outputs are sized for max_voxels
 * and need NOT be zeroed; rows [0, *voxel_num_device) are written completely (padding slots of a voxel as zeros), the
 * rows behind them are left untouched; *voxel_num_device (device int32) = min(voxels found, max_voxels).  Nothing waits
 * on the host: the caller reads the count when it needs it (the host mirror copies it to pinned memory behind the
 * kernels and slices the outputs after the LiDAR branch has been queued -- detector.ISFusionPtsPath.extract_pts_feat).
 * Replaces the same reference code as isf_hard_voxelize (mmdet3d/ops/voxel/src/voxelization_cuda.cu:231-373), whose
 * `voxel_num` is a host-side .item() (voxelization_cuda.cu:366-371). */
int isf_hard_voxelize_device(const float* points, int num_points, int num_features,
                             const float voxel_size_host[3], const float coors_range_host[6],
                             int max_points, int max_voxels, float* voxels, int32_t* coors,
                             int32_t* num_points_per_voxel, int32_t* voxel_num_device, isf_stream_t stream);

/* The samples of a batch in ONE pass (round 6; replaces the per-sample loop of ISFusionDetector.voxelize,
 * mmdet3d/models/detectors/isfusion.py:148-176: `for res in points: voxel_layer(res)` + cat + F.pad with the sample index):
 * points = the samples one behind the other, sample b = rows [point_offsets_host[b], point_offsets_host[b + 1]) (HOST array of
 * batch_size + 1 entries, batch_size <= 16).  Per sample exactly isf_hard_voxelize_device's result (same voxels, same order,
 * max_voxels each); the samples' voxels are written one behind the other: voxels [rows, max_points, C], coors4 [rows, 4] =
 * (sample, z, y, x), num_points_per_voxel [rows], buffers sized batch_size * max_voxels rows; voxel_num_device
 * [batch_size + 1] = voxels kept per sample, then their sum (= rows written).  Asynchronous, no host read-back. */
int isf_hard_voxelize_batched_device(const float* points, const int64_t* point_offsets_host, int batch_size,
                                     int num_features, const float voxel_size_host[3], const float coors_range_host[6],
                                     int max_points, int max_voxels, float* voxels, int32_t* coors4,
                                     int32_t* num_points_per_voxel, int32_t* voxel_num_device, isf_stream_t stream);

/* A3  DynamicScatter ------------------------------------------------------------------------------
 * replaces voxel_layer.dynamic_point_to_voxel_forward(feats, coors, reduce_type) ->
 *   [reduced_feats, out_coors, coors_map int32, reduce_count int32]
 *   voxelization.h:108-121, scatter_points_cuda.cu:183-239.
 * feats [P,C], coors [P,3] int32 (any row with a negative entry is dropped).  Outputs have capacity
 * P rows; the first *num_voxels_host rows are valid: out_coors sorted lexicographically (the order of
 * at::unique_dim(sorted=true)), coors_map[i] = voxel of point i (-1 if dropped), reduce_count.
 * Synchronises once to return the count. */
int isf_dynamic_point_to_voxel_forward(const float* feats, const int32_t* coors, int num_points,
                                       int num_feats, int reduce_type, float* reduced_feats,
                                       int32_t* out_coors, int32_t* coors_map,
                                       int32_t* reduce_count, int* num_voxels_host,
                                       isf_stream_t stream);

/* replaces voxel_layer.dynamic_point_to_voxel_backward(grad_feats, grad_reduced_feats, feats,
 *   reduced_feats, coors_idx, reduce_count, reduce_type)   voxelization.h:123-140,
 *   scatter_points_cuda.cu:241-308.  grad_feats [P,C] is fully overwritten.  Asynchronous. */
int isf_dynamic_point_to_voxel_backward(float* grad_feats, const float* grad_reduced_feats,
                                        const float* feats, const float* reduced_feats,
                                        const int32_t* coors_map, const int32_t* reduce_count,
                                        int num_points, int num_voxels, int num_feats,
                                        int reduce_type, isf_stream_t stream);

/* A4  DynamicVFE.forward (fused) -------------------------------------------------------------------
 * replaces DynamicVFE.forward(features, coors) for the IS-Fusion configuration
 *   (with_cluster_center, with_voxel_center, mode='max', two DynamicVFELayers, eval-mode BN)
 *   mmdet3d/models/voxel_encoders/voxel_encoder.py:453-547, utils.py:129-144.
 * points [P,Cin], coors4 [P,4] (b,z,y,x).  w1 [c1, Cin+6], w2 [c2, 2*c1] in torch Linear layout;
 * scaleN/shiftN = eval BatchNorm1d folded to y = x*scale + shift.  Outputs (capacity P rows):
 * voxel_feats [N,c2], voxel_coors [N,4] sorted by (b,z,y,x) (= per-sample unique_dim order,
 * scatter_points.py:75-96), optional pt2vox [P] (may be NULL).  Synchronises once for the count. */
int isf_dynamic_vfe_forward(const float* points, const int32_t* coors4, int num_points,
                            int in_channels, int batch_size, const float voxel_size_host[3],
                            const float coors_range_host[6], const float* w1, const float* scale1,
                            const float* shift1, int c1, const float* w2, const float* scale2,
                            const float* shift2, int c2, float* voxel_feats, int32_t* voxel_coors,
                            int32_t* pt2vox, int* num_voxels_host, isf_stream_t stream);

/* cfg-1 VFE: HardSimpleVFE.forward(features[M,T,C], num_points[M]) -> [M, num_features]
 *   voxel_encoder.py:28-45.  Asynchronous. */
int isf_hard_simple_vfe(const float* voxels, const int32_t* num_points, int num_voxels,
                        int max_points, int num_point_features, int num_features, float* out,
                        isf_stream_t stream);

/* A5  sparse-conv rulebook --------------------------------------------------------------------------
 * replaces sparse_conv_ext.get_indice_pairs_3d(indices, batch_size, out_shape, spatial_shape, ksize,
 *   stride, padding, dilation, out_padding, subm, transpose) -> [outids, indice_pairs, indice_num]
 *   mmdet3d/ops/bevfusion-ops/spconv/src/all.cc:21-51, include/spconv/spconv_ops.h:27-141,
 *   geometry.h:24-297 (same role: spconv.pytorch / mmcv.ops get_indice_pairs).
 * Native format = output-stationary neighbour table: nbr [K, nbr_stride] int32 with
 * nbr[k*nbr_stride + o] = input row feeding output row o through tap k (-1 if none), tap index
 * k = (kz*Ky + ky)*Kx + kx, in = out*stride - padding + k (cross-correlation, dilation 1).
 * SubM: outputs = inputs in the SAME row order (out_indices may be NULL).  Sparse (strided):
 * out_indices [cap,4] are written sorted by (b,z,y,x) (spconv's GPU path sorts too,
 * spconv_ops.h:130; the CPU path numbers them first-come -- irrelevant after dense()).
 * nbr_stride >= round_up(num_out, 128); the caller provides nbr with K*nbr_stride entries where
 * nbr_stride = isf_nbr_stride(capacity).  Synchronises once for *num_out_host (sparse only). */
int isf_nbr_stride(int num_rows);
int isf_conv_out_shape(const int in_shape_host[3], const int ksize_host[3],
                       const int stride_host[3], const int padding_host[3], int out_shape_host[3]);
int isf_build_rulebook(const int32_t* indices, int num_in, int batch_size,
                       const int spatial_shape_host[3], const int ksize_host[3],
                       const int stride_host[3], const int padding_host[3], int conv_type,
                       int32_t* out_indices, int out_capacity, int32_t* nbr, int nbr_stride,
                       int* num_out_host, isf_stream_t stream);

/* spconv-1 interchange format: indice_pairs [K,2,num_in] (-1 padded) + indice_num [K]
 * (what indice_conv_fp32 consumes, spconv_ops.h:260-271).  Pairs of one tap are ordered by output row. */
int isf_rulebook_to_indice_pairs(const int32_t* nbr, int nbr_stride, int num_out, int num_taps,
                                 int num_in, int32_t* indice_pairs, int32_t* indice_num,
                                 isf_stream_t stream);
int isf_indice_pairs_to_rulebook(const int32_t* indice_pairs, const int32_t* indice_num,
                                 int num_taps, int num_in, int num_out, int32_t* nbr,
                                 int nbr_stride, isf_stream_t stream);

/* A6/A7  sparse convolution forward, fused epilogue ---------------------------------------------------
 * replaces sparse_conv_ext.indice_conv_fp32(features, filters, indice_pairs, indice_num,
 *   num_act_out, inverse, subm)  (spconv_ops.h:260-361; functional.py:22-97) PLUS the
 *   BatchNorm1d(eval) / residual add / ReLU that follow it in SparseSequential / SparseBasicBlock
 *   (ops/sparse_block.py:117-134, sparse_encoder.py:75-104):
 *     y[o,:] = act( (sum_k x[nbr[k][o],:] @ W[k]) * scale + shift + residual[o,:] )
 * features [num_in,Cin]; filters [K,Cin,Cout] = the reference layout [kD,kH,kW,Cin,Cout] (conv.py:100);
 * scale/shift [Cout] (NULL = identity); residual [num_out,Cout] or NULL; relu 0/1; out [num_out,Cout].
 * Each output row is written exactly once (no scatter-add, no atomics).  Asynchronous. */
int isf_sparse_conv_forward(const float* features, int num_in, int c_in, const float* filters,
                            int num_taps, int c_out, const int32_t* nbr, int nbr_stride, int num_out,
                            const float* scale, const float* shift, const float* residual, int relu,
                            float* out, isf_stream_t stream);

/* Same with filters pre-packed by isf_pack_filters (MFMA fragment order); packed size in floats =
 * isf_packed_filter_elems(K, Cin, Cout).  Use this on the hot path: weights are static. */
size_t isf_packed_filter_elems(int num_taps, int c_in, int c_out);
int isf_pack_filters(const float* filters, int num_taps, int c_in, int c_out, float* packed,
                     isf_stream_t stream);
int isf_sparse_conv_forward_packed(const float* features, int num_in, int c_in, const float* packed,
                                   int num_taps, int c_out, const int32_t* nbr, int nbr_stride,
                                   int num_out, const float* scale, const float* shift,
                                   const float* residual, int relu, float* out, isf_stream_t stream);

/* Split-precision ("f16x3") sparse convolution -------------------------------------------------------------
 * Same contract as isf_sparse_conv_forward_packed, evaluated on the f16 matrix cores with fp32-equivalent
 * accuracy: operands are carried as hi + lo f16 halves (22 significant bits), products as
 * a_lo*b_hi + a_hi*b_lo + a_hi*b_hi with fp32 accumulation.  Activations are exchanged in the SPLIT format:
 * row-major [N, C/32] chunks of 128 bytes, a chunk = 32 channels as 4 x (8 f16 hi) followed by 4 x (8 f16 lo)
 * (same 4 bytes/element as fp32; the 16-byte pieces the four k-group lanes of an MFMA row fetch together are
 * contiguous); isf_f32_to_split / isf_split_to_f32 convert ([N, C] row-major fp32, N*C a multiple of 32).
 * |activation| must be < 65504.  Cin, Cout in {32,64,128,256}.
 * `mode` of isf_sparse_conv_forward_f16x3 (per call; there is no process-wide switch): 0 = split precision (default);
 * 1 = single-pass f16 (opt-in): the same kernels fetch and multiply only the hi halves -- f16 operands, fp32
 * accumulate, the accuracy of the reference's indice_conv_half under fp16 autocast (BASELINE configs[4]); results are
 * still exchanged in the split format.  TIMING DIAGNOSTICS (tools/conv_knockout.sh; never in production): 2 = no
 * activation gathers, 4 = no weight streaming, 6 = neither, 8 = no main loop -- the RESULTS ARE GARBAGE, only kernel
 * times are meaningful; 16 = gather every row (no neighbour sharing): results valid and
 * bit-identical to mode 0, the reference the sharing is tested against; +32 (combinable) = uniform row tiles instead
 * of the full / half-tile mix that evens out the row groups per SIMD on launches of a single round of workgroups
 * (results bit-identical: a row's products and their order do not depend on the tile it falls into).
 * Round 5, mode 0 only, results valid and bit-identical, deep (128 / 256-column) shapes: +4096 / +8192 = one column block
 * for the 256-column layers (4 x 32-row / 8 x 16-row waves), +65536 = staggered issue phases, +131072 = round 4's issue
 * phase (index reads inside the step, separate weight-DMA pieces), +262144 = gathered rows two steps ahead -- experiments
 * measured slower than the default, kept as tested opt-ins.
 * Round 6, mode 0 / +32, c_out = 256 (other shapes ignore it): +524288 = CHUNK SPLIT -- a tile is computed by two workgroups,
 * each over half of the 32-channel chunks; the second to arrive adds the other's accumulator tile and runs the epilogue.
 * Deterministic, NOT the bits of mode 0 (the sum over chunks becomes (lower half) + (upper half)); measured slower; opt-in.
 * Workgroup shape (mode 0 / +32; chosen per launch from c_out and num_out, never changes a result): 4 waves x 32 rows and two
 * 128-column blocks for c_out = 256, 8 waves x 32 rows for c_out = 128 from 2048 rows up; launches the tile plan would cut
 * into half tiles only (c_out = 256: num_out <= 96 x CUs; c_out = 128: 2048 <= num_out <= 256 x CUs) run with 16 rows per
 * wave instead.  Mode 16 keeps the 32-row shapes: the bit-equality reference of that rule too. */
size_t isf_packed_filter16_bytes(int num_taps, int c_in, int c_out);
int isf_pack_filters_f16x3(const float* filters, int num_taps, int c_in, int c_out, void* packed16,
                           isf_stream_t stream);
/* ... from the per-tap TRANSPOSE: filters_t [num_taps, c_out, c_in] -- e.g. the forward filter when the packed one (c_in x c_out
 * = forward c_out x c_in) is the data gradient's, dX = dY W_k^T: no transposed copy of the weights per layer and step */
int isf_pack_filters_f16x3_transposed(const float* filters_t, int num_taps, int c_in, int c_out, void* packed16,
                                      isf_stream_t stream);
int isf_f32_to_split(const float* x, size_t num_elems, void* xs, isf_stream_t stream);
int isf_split_to_f32(const void* xs, size_t num_elems, float* x, isf_stream_t stream);
/* f16 STORAGE (mode 257 = 256 | 1 of isf_sparse_conv_forward_f16x3; isf_encoder_options.precision = 2): features,
 * residual and output rows are plain f16, [N, C] row-major, 2 bytes per element -- the data type of the reference's
 * indice_conv_half / indice_conv_backward_half end to end (mmdet3d/ops/bevfusion-ops/spconv/src/all.cc:35-37; BASELINE
 * configs[4] "fp16, HBM-bound") -- with f16 operands and fp32 accumulation; every layer moves half the activation
 * bytes of the split format.  isf_f32_to_half / isf_half_to_f32 convert (num_elems a multiple of 8). */
int isf_f32_to_half(const float* x, size_t num_elems, void* xh, isf_stream_t stream);
int isf_half_to_f32(const void* xh, size_t num_elems, float* x, isf_stream_t stream);
int isf_sparse_conv_forward_f16x3(const void* features_split, int num_in, int c_in, const void* packed16,
                                  int num_taps, int c_out, const int32_t* nbr, int nbr_stride, int num_out,
                                  const float* scale, const float* shift, const void* residual_split, int relu,
                                  void* out_split, int mode, isf_stream_t stream);
/* TILE ORDER of a launch that is resident in one round of workgroups.  Such a launch ends when its busiest CU does; the
 * tiles a CU gets in launch order (slots j, j + 32, j + 64 of its XCD) add up unevenly because the work of a tile -- the
 * (16-row group, tap) pairs that have a neighbour -- varies ~3x across a LiDAR sweep.  isf_sparse_conv_tile_order counts
 * that work per tile of the launch isf_sparse_conv_forward_f16x3 would make for (c_in, c_out, mode, num_out) and hands
 * the tiles out longest-first to the least-loaded CU with a free slot; `order` (and the scratch `work`) hold at most
 * 8 * 255 ints, *num_entries = how many were written (0: the launch is not a single round -- pass order = NULL).  A
 * table belongs to ONE kernel's launch plan: mode + 2048 builds it for isf_sparse_conv_forward_dma (whose workgroups per
 * CU, and with them the tiles, differ from isf_sparse_conv_forward_f16x3's on the same shape).
 * One table serves every layer of that channel shape on the rulebook (isf_sparse_encoder_forward builds it per level
 * behind the neighbour table).  isf_sparse_conv_forward_f16x3_ordered = the same convolution with workgroup slot j
 * working on tile order[j]: results bit-identical to isf_sparse_conv_forward_f16x3 (the same tiles compute the same
 * rows).  Replaces nothing in the reference (its gather -> GEMM -> scatter has no tiles); measured in DESIGN.md
 * section 5.1.  isf_sparse_conv_trace: DIAGNOSTIC -- the production launch of a 128 -> 128 or 256 -> 256 layer that also
 * writes 8 int64 per workgroup (trace [trace_capacity_blocks * 8], zero it first; a launch of more workgroups than the
 * buffer holds is refused: 100 MHz time stamps at entry / after the
 * prologue / after the multiply loop / at exit, steps, HW_ID, XCC_ID, first row | half tile << 32); tools/conv_trace.py. */
int isf_sparse_conv_tile_order(const int32_t* nbr, int nbr_stride, int num_taps, int num_out, int c_in, int c_out,
                               int mode, int32_t* work, int32_t* order, int* num_entries, isf_stream_t stream);
int isf_sparse_conv_forward_f16x3_ordered(const void* features_split, int num_in, int c_in, const void* packed16,
                                          int num_taps, int c_out, const int32_t* nbr, int nbr_stride, int num_out,
                                          const float* scale, const float* shift, const void* residual_split, int relu,
                                          void* out_split, int mode, const int32_t* order, isf_stream_t stream);
/* TILE TABLE of a launch that is resident in one round of workgroups (levels 3 / 4: 2.5 tiles per compute unit).  Uniform
 * tiles carry 3x different matrix work (scene density), so the launch ends with its densest CU.  isf_sparse_conv_tile_table
 * deals the 16-row groups of every XCD's row range to that XCD's compute units as contiguous runs of about EQUAL WORK
 * (taps with a neighbour per group) and cuts a CU's run into full tiles plus one remainder tile, slot = where the
 * dispatcher places it: table [parts][workgroups per CU * CUs per XCD][2] = (first group, groups), a buffer of 3072 ints
 * (8 parts x 2 x 192 slots per XCD; a launch that would need more is refused with ISF_ERR_UNSUPPORTED, never written past the
 * buffer); scratch holds 2 * ceil(num_out / 16) ints; *num_ints = ints written (0: the launch is not one round -- use the
 * plain entry).
 * isf_sparse_conv_forward_f16x3_tiled runs the convolution over it: BIT-IDENTICAL results (a row's products and their
 * order do not depend on the tile it falls into); the table belongs to one (c_in, c_out, mode): the workgroup shape depends
 * on all three.  MEASURED SLOWER than uniform tiles + isf_sparse_conv_tile_order on the MI355X (256 -> 256: 1.32 vs 1.265 ms
 * per step; profiles/r04_tile_tables.txt, profiles/EXPERIMENTS.md section 5.4: a tile's time follows its STEP count, not its matrix
 * work), so it is an OPT-IN: isf_sparse_encoder_forward builds the tables with diagnostic +32768.
 * isf_sparse_conv_tile_table_host: the same arithmetic on the host (tests). */
int isf_sparse_conv_tile_table(const int32_t* nbr, int nbr_stride, int num_taps, int num_out, int c_in, int c_out,
                               int mode, int32_t* scratch, int32_t* table, int* num_ints, isf_stream_t stream);
int isf_sparse_conv_forward_f16x3_tiled(const void* features_split, int num_in, int c_in, const void* packed16,
                                        int num_taps, int c_out, const int32_t* nbr, int nbr_stride, int num_out,
                                        const float* scale, const float* shift, const void* residual_split, int relu,
                                        void* out_split, int mode, const int32_t* table, isf_stream_t stream);
int isf_sparse_conv_tile_table_host(const int32_t* work, int num_groups, int part_groups, int parts, int cus, int wgs_per_cu,
                                    int groups_per_tile, int32_t* tiles, int* fits);
int isf_sparse_conv_trace(const void* features_split, int num_in, int c_in, const void* packed16, int num_taps,
                          int c_out, const int32_t* nbr, int nbr_stride, int num_out, const float* scale,
                          const float* shift, const void* residual_split, int relu, void* out_split,
                          const int32_t* order, long long* trace, int trace_capacity_blocks, int* grid_blocks,
                          isf_stream_t stream);
/* DIAGNOSTIC: isf_sparse_conv_trace plus per-WAVE phase stamps of every step of the multiply loop (shader clock,
 * s_memtime: 1 tick = 1 shader cycle): top of the step / after its s_waitcnt vmcnt(0) / after the barrier / after issuing
 * the next step's loads; the multiply section is what is left until the next top.  This image's rocprofv3 has no
 * thread-trace decoder (--att: "rocprof-trace-decoder library path not found"), so this is the instruction-level account of
 * the dominant kernel.  trace: [*grid_blocks][8] int64 as isf_sparse_conv_trace, then per wave (*waves_per_block per
 * workgroup) *dwords_per_wave uint32: {clock at loop entry lo, hi, HW_ID, steps, tap mask of row group 0, of row group 1,
 * tap mask of the workgroup, clock at loop exit lo} + 4 per step; trace_bytes is checked against the launch (zero the
 * buffer first).  tools/conv_phase_trace.py -> profiles/r05_att_256.txt. */
int isf_sparse_conv_phase_trace(const void* features_split, int num_in, int c_in, const void* packed16, int num_taps,
                                int c_out, const int32_t* nbr, int nbr_stride, int num_out, const float* scale,
                                const float* shift, const void* residual_split, int relu, void* out_split,
                                const int32_t* order, long long* trace, size_t trace_bytes, int* grid_blocks,
                                int* waves_per_block, int* dwords_per_wave, isf_stream_t stream);

/* DIAGNOSTIC: the per-workgroup trace of the NARROW layers' kernel (isf_sparse_conv_forward_dma / _dma_lines; c_in,
 * c_out in {32, 64}): one production launch with 16 int64 per workgroup -- constant-clock (100 MHz) stamps at entry / after
 * the prologue / after the multiply loop / at exit, steps, HW_ID, XCC_ID, first row | half tile << 32, then wave 0's
 * shader-clock cycles summed over the steps: at the per-step vmcnt(0), at the barrier, in the section that reads the
 * transit / weight buffers and issues the next step's loads, in the multiply section; [12], [13]: of the third, the fragment
 * reads' LDS round trip and the index arithmetic + weight run (the rest: the row gathers).  table / mask: the dense neighbour
 * table (mask NULL, taps_per_line ignored) or the line-compressed one.  tools/conv_trace.py --level 0 | 1. */
int isf_sparse_conv_dma_trace(const void* features_split, int num_in, int c_in, const void* packed16, int num_taps,
                              int taps_per_line, int c_out, const int32_t* table, const uint32_t* mask, int nbr_stride,
                              int num_out, const float* scale, const float* shift, const void* residual_split, int relu,
                              void* out_split, long long* trace, int trace_capacity_blocks, int* grid_blocks,
                              isf_stream_t stream);
/* The same convolution for the NARROW layers (c_in, c_out in {32, 64}) with the gathered rows brought in by LDS-DMA
 * (isf_spconv_dma.hip; mode 0 | 1 | 257, +32; order: NULL or isf_sparse_conv_tile_order's table).  A gather instruction of
 * isf_sparse_conv_forward_f16x3 loads straight into the MFMA operand layout -- four different rows = four cache lines per
 * lane quad: 64 address-unit cycles -- and on the narrow layers (few MFMAs per gathered row) the address unit sets the
 * step time.  Here a quad fetches the 64 contiguous bytes of ONE row (16 cycles) into a wave-private LDS transit buffer
 * and the MFMA layout is produced by the LDS read; no neighbour table in LDS, 5-6 workgroups per CU.  Results are
 * BIT-IDENTICAL to isf_sparse_conv_forward_f16x3 (same products in the same order per accumulator);
 * isf_sparse_encoder_forward / isf_lidar_branch_forward run their narrow layers on it (diagnostic +128: gather kernel).
 * Replaces the reference's gather stage (bevfusion-ops/spconv/include/spconv/reordering.cu.h:21-97) for those layers. */
int isf_sparse_conv_forward_dma(const void* features_split, int num_in, int c_in, const void* packed16, int num_taps,
                                int c_out, const int32_t* nbr, int nbr_stride, int num_out, const float* scale,
                                const float* shift, const void* residual_split, int relu, void* out_split, int mode,
                                const int32_t* order, isf_stream_t stream);
/* LINE-COMPRESSED neighbour table (narrow layers).  The taps of one (kz, ky) line of a 3-wide kernel probe x-adjacent cells
 * and rows are sorted by (b, z, y, x), so the neighbours a row has through a line are CONSECUTIVE rows: one int32 per line
 * (the row of the first present neighbour, -1 if none) + one bit per tap say everything the dense table says --
 *   lines [num_taps / taps_per_line][nbr_stride] int32, mask [nbr_stride] uint32 (bit k: tap k present),
 *   nbr[k][o] = mask[o] bit k ? lines[k / tpl][o] + popcount(mask[o] bits of that line below k) : -1
 * -- in 40 bytes per row (3 x 3 x 3) instead of 108: the dense table of a narrow layer is as many bytes per row as a
 * 32-channel activation row, read twice per convolution (tap masks, then indices).  Only tables in rank order qualify
 * (what isf_build_rulebook produces for sorted inputs; isf_rulebook_to_lines sets *not_consecutive_flag (device int, may be
 * NULL) to 1 for any other table).  isf_rulebook_to_lines / isf_lines_to_rulebook convert; isf_sparse_conv_forward_dma_lines
 * = isf_sparse_conv_forward_dma reading the compressed table (a lane keeps its rows' masks in registers and loads one
 * index per LINE): results BIT-IDENTICAL.  isf_sparse_encoder_forward / isf_lidar_branch_forward build the tables of their
 * narrow levels directly in this form (diagnostic +16384: dense tables).  taps_per_line = kernel width (1 or 3).
 * Replaces nothing in the reference (its rulebook is pair lists, indice.cu.h:22-203); measured in DESIGN.md section 5.3. */
int isf_rulebook_to_lines(const int32_t* nbr, int nbr_stride, int num_taps, int taps_per_line, int num_out,
                          int32_t* lines, uint32_t* mask, int* not_consecutive_flag, isf_stream_t stream);
int isf_lines_to_rulebook(const int32_t* lines, const uint32_t* mask, int nbr_stride, int num_taps, int taps_per_line,
                          int32_t* nbr, isf_stream_t stream);
int isf_sparse_conv_forward_dma_lines(const void* features_split, int num_in, int c_in, const void* packed16, int num_taps,
                                      int taps_per_line, int c_out, const int32_t* lines, const uint32_t* mask,
                                      int nbr_stride, int num_out, const float* scale, const float* shift,
                                      const void* residual_split, int relu, void* out_split, int mode,
                                      isf_stream_t stream);
/* The same convolution for the 256-COLUMN layers (c_out = 256, c_in in {128, 256}: levels 3 / 4 of the encoder) as ONE
 * WORKGROUP PER COMPUTE UNIT (isf_spconv_cu.hip).  A level-3 launch is 160 rows x 256 columns per CU: cut into 128 x 128
 * tiles it ends with its busiest CU (1.4x the mean work: the density of a LiDAR sweep varies 3x) and every tile streams
 * its own weight stage.  isf_sparse_conv_cu_plan cuts the rows of a rulebook into units of whole 16-row groups with EQUAL
 * matrix work (taps with a neighbour per group; <= 256 rows; cus * r units) -- plan_buf holds
 * isf_sparse_conv_cu_plan_ints(num_out) int32, the plan struct points into it, no host sync (the grid is sized by
 * max_units) -- once per rulebook; isf_sparse_conv_forward_cu runs one workgroup per unit over ALL 256 output columns; the
 * production variant (variant 0 = kCuWavesProd) is FOUR waves, wave w owns columns [64 w, 64 w + 64) (256 accumulator
 * registers = the AGPR half of the file; variants 6 / 7 are the 8-wave x 32-column shape), its weight fragments go
 * global -> VGPR (no LDS, one stream per CU instead of one per tile), the gathered rows come in by LDS-DMA once per CU
 * through a ring of prefetch depth + 1 stages.  Results are
 * BIT-IDENTICAL to isf_sparse_conv_forward_f16x3 (mode 0).  MEASURED SLOWER than the tile kernel on the MI355X (256 -> 256:
 * 1.39 .. 1.50 ms per step against 1.26; profiles/r04_cu_kernel_ab.txt, profiles/EXPERIMENTS.md section 5.2), so it is an OPT-IN:
 * isf_sparse_encoder_forward / isf_lidar_branch_forward run their 256-column layers on it with diagnostic +512.  isf_sparse_conv_cu_plan_host / _max_units: the plan
 * arithmetic on the host (tests, tools; no device work): work [num_groups] -> units [max_units][2] = (first group, groups).
 * Replaces the reference's per-tap gather -> GEMM -> scatter-add (spconv_ops.h:260-361) for those layers. */
typedef struct isf_conv_cu_plan {
  const int32_t* group_masks;   /* device [ceil(num_out / 16)] */
  const int32_t* units;         /* device [*num_units][2] */
  const int32_t* num_units;     /* device scalar */
  int max_units;                /* host bound of *num_units */
  int num_out;                  /* rows the plan was built for */
  int variant;                  /* 0 = production.  DIAGNOSTICS: 1 / 2 / 3 = no gathers / no weight loads / neither (TIMING
                                   ONLY, results garbage); 4 / 5 = 4-wave workgroups at prefetch depth 1 / 2 steps, 6 / 7 =
                                   8-wave workgroups at depth 1 / 2 (results valid); round 6, hand-scheduled assembly
                                   multiply phase: 8 = 8 waves, one workgroup per CU; 9 / 10 = two 4-wave workgroups per
                                   CU over units of <= 8 groups at depth 1 / 2 (valid); 11-15 their knock-outs (timing
                                   only).  isf_sparse_conv_cu_plan READS this field: the variant decides the unit shape */
  int cap;                      /* groups per unit the plan was cut for (16 or 8); set by isf_sparse_conv_cu_plan */
} isf_conv_cu_plan;
int isf_sparse_conv_cu_plan_ints(int num_out, size_t* num_ints);
int isf_sparse_conv_cu_plan(const int32_t* nbr, int nbr_stride, int num_taps, int num_out, int32_t* plan_buf,
                            isf_conv_cu_plan* plan, isf_stream_t stream);
int isf_sparse_conv_forward_cu(const void* features_split, int num_in, int c_in, const void* packed16, int num_taps,
                               int c_out, const int32_t* nbr, int nbr_stride, int num_out, const float* scale,
                               const float* shift, const void* residual_split, int relu, void* out_split,
                               const isf_conv_cu_plan* plan, isf_stream_t stream);
int isf_sparse_conv_cu_plan_host(const int32_t* work, int num_groups, int cus, int32_t* units, int max_units,
                                 int* num_units);
int isf_sparse_conv_cu_max_units(int num_groups, int cus);
/* The same convolution with the tile's input rows staged in LDS ("LDS staging of active-voxel tiles"; the reference's
 * gather stage: bevfusion-ops/spconv/include/spconv/reordering.cu.h:21-97 -> staging buffer -> GEMM, spconv_ops.h:300-345).
 * isf_rulebook_stage_tables derives the staging tables of a neighbour table once per rulebook: for every unit of
 * isf_stage_unit_rows() (64) consecutive output rows, ulist [nbr_stride / 64][isf_stage_unit_cap()] = the distinct
 * input rows the unit's taps name (ascending), ucount [nbr_stride / 64] = how many, and slots [num_taps][nbr_stride]
 * (uint16) = the position of nbr[k][o] in its unit's list, 0xFFFF where nbr is -1.  isf_sparse_conv_forward_staged
 * copies the listed rows global -> LDS once per tile and 32-channel chunk (row-coalesced LDS-DMA) and reads the A
 * operands of all taps from LDS; `stage_rows` = LDS rows per 128-row tile (clamped to what 160 KiB hold; list entries
 * beyond a unit's share are gathered from memory, so every value is correct).  Results are bit-identical to
 * isf_sparse_conv_forward_f16x3 (same products, same order).  mode: 0 | 1 (single-pass f16), +32 uniform tiles. */
int isf_stage_unit_rows(void);
int isf_stage_unit_cap(void);
int isf_rulebook_stage_tables(const int32_t* nbr, int nbr_stride, int num_taps, uint16_t* slots, int32_t* ulist,
                              int32_t* ucount, isf_stream_t stream);
int isf_sparse_conv_forward_staged(const void* features_split, int num_in, int c_in, const void* packed16,
                                   int num_taps, int c_out, const uint16_t* slots, int nbr_stride,
                                   const int32_t* ulist, const int32_t* ucount, int num_out, const float* scale,
                                   const float* shift, const void* residual_split, int relu, void* out_split,
                                   int stage_rows, int mode, isf_stream_t stream);
/* A7  SparseConvTensor.dense() + view(N, C*D, H, W) ---------------------------------------------------
 * replaces structure.py:49-59 + sparse_encoder.py:133-136: out[b, c*D+z, y, x] = feats[i,c], zeros
 * elsewhere; out [B, C*D, H, W] is written completely (no separate memset).  Asynchronous. */
int isf_sparse_to_dense_bev(const float* features, const int32_t* indices, int num_rows, int channels,
                            int batch_size, int D, int H, int W, float* out, isf_stream_t stream);

/* A7  whole SparseEncoder.forward in one call -----------------------------------------------------------
 * replaces SparseEncoder.forward(voxel_features, coors, batch_size)
 *   mmdet3d/models/middle_encoders/sparse_encoder.py:107-138 (eval-mode BN folded).
 * A plan is an ordered list of conv layers; SubM layers at one resolution share one rulebook. */
typedef struct isf_conv_layer {
  int conv_type;       /* ISF_CONV_SUBM | ISF_CONV_SPARSE */
  int ksize[3], stride[3], padding[3];
  int c_in, c_out;
  const float* packed; /* device: isf_pack_filters output (fp32 MFMA path) */
  const void* packed16; /* device: isf_pack_filters_f16x3 output, or NULL (then the fp32 path runs) */
  const float* scale;  /* device [c_out] */
  const float* shift;  /* device [c_out] */
  int relu;            /* apply ReLU at the end */
  int residual_from;   /* -2 none; -1 the encoder input; i>=0 output of layer i (added before ReLU) */
} isf_conv_layer;

typedef struct isf_encoder_stats { /* filled on the host after the call (for roofline accounting) */
  int num_layers;
  int num_in[32], num_out[32];
  long long pairs[32]; /* sum over taps of valid (in,out) pairs of layer i */
  float ms[32];        /* hipEvent time of layer i's conv kernel when timing was requested, else 0 */
  int precision;       /* 0 = fp32 MFMA kernels ran, 1 = f16x3 split-precision MFMA kernels ran */
} isf_encoder_stats;

/* per-call options of the two engine entry points (NULL = all defaults; no process-wide state):
 * precision  0 = f16x3 split MFMA when every layer carries packed16, else fp32 MFMA (default); 1 = force the fp32 MFMA
 *            kernels; 2 = f16 storage + single-pass f16 arithmetic (opt-in, the reference's fp16 mode: activations are
 *            f16 rows between the layers, mode 257 of isf_sparse_conv_forward_f16x3);
 * diagnostic timing diagnostics of the conv kernels (0 = off; 2 / 4 / 6 / 8 / 16, +32: see isf_sparse_conv_forward_f16x3;
 *            +64 = tiles in launch order, no isf_sparse_conv_tile_order tables; +128 = narrow layers on the gather
 *            kernel instead of isf_sparse_conv_forward_dma; +256 (isf_lidar_branch_forward) = the voxel encoder writes
 *            fp32 rows and a conversion pass makes the split rows, instead of writing them directly; +512 = the
 *            256-column layers on isf_sparse_conv_forward_cu (one workgroup per CU; opt-in: measured slower than
 *            the tile kernel) -- results bit-identical either way; +1024 * v = isf_conv_cu_plan.variant v of those layers (timing diagnostics, v < 16);
 *            +16384 = dense neighbour tables for the narrow layers instead of the line-compressed ones -- bit-identical;
 *            +131072 = the encoder's per-level row counts reach the host through hipMemcpyAsync + synchronise instead of
 *            the pinned-memory mailbox (post_int / wait_int) -- bit-identical;
 *            +65536 (isf_lidar_branch_forward) = one dynamic-voxelize launch per frame + a separate byte-map marking pass
 *            instead of the fused voxelize + mark launch -- bit-identical;
 *            +32768 = equal-work tile tables for the deep levels instead of uniform tiles + tile order (opt-in:
 *            measured slower) -- bit-identical;
 *            layers run on the gather kernel whenever a diagnostic other than 32 is set.
 *            Round 5 (all bit-identical to the default, all measured SLOWER, kept as tested opt-ins):
 *            +262144 / +524288 = the 256-column layers as ONE column block, 4 x 32-row / 8 x 16-row waves (conv mode 4096 /
 *            8192); +1048576 = staggered issue phases inside the deep layers' workgroups (conv mode 65536); +4194304 = the
 *            gathered rows two steps ahead (conv mode 262144); +2097152 = round 4's issue phase (row-index reads inside the
 *            step, four separate weight-DMA pieces; conv mode 131072) -- the A/B partner of the default, which reads the
 *            indices one step ahead and stages a wave's weight share as one run.
 *            Round 6: +536870912 = the 256-column layers with the chunk split (conv mode 524288): valid, deterministic, not
 *            the default's bits, measured slower (opt-in). */
typedef struct isf_encoder_options {
  int precision;
  int diagnostic;
  int stage_rows;  /* LDS-staged input rows per 128-row conv tile (isf_sparse_conv_forward_staged); 0 = the library's
                      per-layer default, -1 = staging off (every layer on the gather kernel) */
  int stage_mask;  /* with stage_rows > 0: bit i = layer i runs staged (tuning); 0 = every layer */
  int bev_format;  /* spatial_features: 0 = fp32 [B, C*D, H, W] (the reference's layout, sparse_encoder.py:137-139);
                      1 = the same map as split-format token matrices, one per 256-channel group: group g at byte offset
                      g * B*H*W * 1024, [B*H*W, 256] in the split activation format (token = (b*H + y)*W + x, channel
                      c*D + z) -- what the fusion encoder's convolutions read (isf_sparse_conv_forward_f16x3 over
                      isf_dense_grid_rulebook): no isf_nchw_to_split pass.  Same byte count; precision 0 only */
} isf_encoder_options;

int isf_sparse_encoder_forward(const float* voxel_features, const int32_t* coors, int num_voxels,
                               int batch_size, const int sparse_shape_host[3],
                               const isf_conv_layer* layers_host, int num_layers,
                               float* spatial_features, /* [B, C_last*D_last, H_last, W_last] */
                               int out_shape_host[4],   /* C*D, H, W and N_last (may be NULL) */
                               isf_encoder_stats* stats_host /* may be NULL */, int time_layers,
                               const isf_encoder_options* options /* may be NULL */, isf_stream_t stream);

/* LiDAR branch in one call: dynamic voxelize + DynamicVFE + SparseEncoder (isfusion.py:103-111) */
typedef struct isf_vfe_params {
  int in_channels, c1, c2;
  const float *w1, *scale1, *shift1, *w2, *scale2, *shift2; /* device */
  float voxel_size[3], coors_range[6];
} isf_vfe_params;

int isf_lidar_branch_forward(const float* points, const int64_t* point_offsets_host, int batch_size,
                             const isf_vfe_params* vfe_host, const int sparse_shape_host[3],
                             const isf_conv_layer* layers_host, int num_layers,
                             float* spatial_features, int out_shape_host[4],
                             isf_encoder_stats* stats_host, int time_layers,
                             const isf_encoder_options* options /* may be NULL */, isf_stream_t stream);

/* ===================================================================================================
 * HSF / IGF rows (A8, A10-A14).  Dense 3x3 convolutions around them stay stock
 * PyTorch-ROCm ops (north_star); everything below is what the reference runs through custom CUDA ops
 * or long chains of small torch kernels.
 * =================================================================================================== */

/* nn.Linear with fused epilogue -----------------------------------------------------------------------
 * replaces F.linear + bias + {GELU | ReLU} + residual add + LayerNorm chains of
 *   mmdet3d/models/sst/sst_basic_block_v2.py:104-126 (EncoderLayer), backbones/sst_v2.py:83 (linear0),
 *   middle_encoders/fusion_encoder.py:560-600 (MSDeformAttn projections), :653-674 (decoder layer FFN),
 *   :221-470 (MultiheadAttention in/out projections), :489-496 (Instane2SceneAtt).
 * isf_pack_linear splits weight [out, in] (torch layout) once into f16 hi/lo MFMA fragments.
 * y[r, :] = LN( act( x[r, :] . W^T + bias + row_table[row_table_index[r], :] ) + residual[r, :] )
 *   activation: 0 none, 1 ReLU, 2 GELU(erf);  LN only when ln_gamma != NULL (out_features <= 256).
 * in_features in {32,64,128,256}; out_features % 16 == 0; ldx % 8 == 0. */
size_t isf_packed_linear_bytes(int out_features, int in_features);
int isf_pack_linear(const float* weight, int out_features, int in_features, void* packed, isf_stream_t stream);
/* ... from the transpose: weight_t [in_features, out_features] row-major (the forward weight, for dX = dY W) */
int isf_pack_linear_transposed(const float* weight_t, int out_features, int in_features, void* packed, isf_stream_t stream);
int isf_linear_forward(const float* x, int num_rows, int in_features, int ldx, const void* packed_weight,
                       int out_features, const float* bias, const float* row_table, const int32_t* row_table_index,
                       int activation, const float* residual, const float* ln_gamma, const float* ln_beta,
                       float ln_eps, float* y, int ldy, int x_hw, int residual_hw, int y_hw, isf_stream_t stream);
/* x_hw / residual_hw / y_hw: 0 = row-major [num_rows, C]; hw > 0 = the tensor is channels-first [B, C, hw] with row
 * r = b*hw + pos (a BEV map [B, C, H, W], hw = H*W, hw % 4 == 0): the NCHW <-> token transposes of
 * fusion_encoder.py:1163, sst_v2.py:97-133 and :480-496 happen inside the GEMM's loads / stores. */

/* A10/A11  the attention half of an SST encoder layer in ONE kernel on the matrix cores -------------------------------
 * replaces, for the fusion encoder's dense grids (every cell a token, row = (b*S + y)*S + x), the whole of
 *   WindowAttention.forward + the residual / norm1 of EncoderLayer.forward (models/sst/sst_basic_block_v2.py:41-75,
 *   :104-116) incl. flat2window / window2flat / the position embedding add (ops/sst/sst_ops.py:63-143, 219-268):
 *     y = LayerNorm( x + out_proj( softmax(q k^T / sqrt(hd)) v ) ),  q, k = in_proj(x + pos), v = in_proj(x)
 * x, y [B*S*S, d] fp32; in_proj_bias [3d]; pos_table [window^2, 3d] = pos_embed @ in_proj_weight[:2d]^T (zeros in the
 * v columns); packed = isf_pack_window_block(in_proj_weight [3d, d], out_proj_weight [d, d]); shift as in
 * isf_window_attention_forward.  q, k, v, scores and the attention output never reach memory.  8 heads, 6x6 windows,
 * d = 128 (the 180 x 180 level; the d = 256 level runs linear -> isf_window_attention_forward -> linear, which
 * measured faster there). */
size_t isf_packed_window_block_bytes(int embed_dims);
int isf_pack_window_block(const float* in_proj_weight, const float* out_proj_weight, int embed_dims, int num_heads,
                          void* packed, isf_stream_t stream);
int isf_window_block_forward(const float* x, int batch_size, int grid_size, int embed_dims, int num_heads, int window,
                             int shift, const void* packed, const float* in_proj_bias, const float* pos_table,
                             const float* out_proj_bias, const float* ln_gamma, const float* ln_beta, float ln_eps,
                             float* y, isf_stream_t stream);

/* A10/A11  window attention core on a dense token grid (the unfused form; training uses it) -------------------
 * replaces get_window_coors / flat2window / nn.MultiheadAttention / window2flat of
 *   mmdet3d/ops/sst/sst_ops.py:219-268, 63-143 and models/sst/sst_basic_block_v2.py:41-75
 * for the fusion encoder's dense grids (fusion_encoder.py:1151-1189: every cell is a token, token row =
 * (b*S + y)*S + x).  qkv [B*S*S, 3*d] = (q | k | v) projections (q, k of x + pos, v of x);
 * shift 0: windows aligned at 0, shift 1: windows offset by -window/2 (partial edge windows attend among the
 * cells they hold).  out [B*S*S, d] = softmax(q k^T / sqrt(hd)) v per head, heads concatenated. */
int isf_window_attention_forward(const float* qkv, int batch_size, int grid_size, int embed_dims, int num_heads,
                                 int window, int shift, float* out, isf_stream_t stream);

/* A13/A14  multi-head attention core --------------------------------------------------------------------
 * replaces the softmax(QK^T)V part of multi_head_attention_forward (fusion_encoder.py:371-470; the head's
 * cross attention, transfusion_head_v2.py:104-108): q [B*Lq, ldq], k / v [B*Lk, ldkv] (already projected; head h =
 * columns h*hd..), out [B*Lq, ldo].  head_dim 16 (the IS-Fusion configuration): matrix-core kernel, keys resident in
 * LDS (up to 512 per workgroup; more keys are split and merged); head_dim 32: fp32 vector kernels.  Asynchronous. */
int isf_attention_forward(const float* q, int ldq, const float* k, const float* v, int ldkv, int batch_size,
                          int num_queries, int num_keys, int embed_dims, int num_heads, float* out, int ldo,
                          isf_stream_t stream);
/* ... in TRAINING mode with nn.MultiheadAttention's dropout on the attention probabilities (fusion_encoder.py:458
 * F.dropout(attn_output_weights, p); the IGF modules use p = 0.1, :476, :614): a probability is zeroed with probability
 * dropout_p and the kept ones scaled by 1 / (1 - dropout_p); the row normalisation keeps every term.  The keep / drop
 * decision of (sample * heads + head, query, key) is a counter-based hash of `seed` (MurmurHash3's finaliser over the
 * packed indices; isf_attention_backward_dropout recomputes it from the same seed), so no mask is stored.  dropout_p = 0
 * is isf_attention_forward.  head_dim 16, num_keys <= 512. */
int isf_attention_forward_dropout(const float* q, int ldq, const float* k, const float* v, int ldkv, int batch_size,
                                  int num_queries, int num_keys, int embed_dims, int num_heads, float dropout_p,
                                  unsigned long long seed, float* out, int ldo, isf_stream_t stream);

/* A14  per-channel map attention -----------------------------------------------------------------------
 * replaces fusion_encoder.py:497-502: for each of num_maps = B*C maps (size x size, row-major)
 *   out = query_scene + softmax(query_scene . query_ins^T, dim=-1) . query_ins */
int isf_channel_attention_forward(const float* query_scene, const float* query_ins, int num_maps, int size,
                                  float* out, isf_stream_t stream);

/* A8  Point-to-Grid sampling ------------------------------------------------------------------------------
 * replaces ISFusionEncoder.img_fv_to_bev + img_point_sampling (fusion_encoder.py:965-1070).
 * pillars [M, slots, pillar_ld] (x, y, z first; zero-padded slots are sampled like the reference does),
 * pillar_coors [M, 4] (b, z, y, x); img_nhwc [B*num_cam, feat_h, feat_w, C]; cam_params [B*num_cam, 20]:
 *   M(3x3) = lidar2img[:3,:3] . inv(lidar_aug[:3,:3]), v(3) = lidar2img[:3,3] - M . lidar_aug[:3,3],
 *   A(2x3) = img_aug[:2,:3], a(2) = img_aug[:2,3]   (host-side 4x4 algebra, see fusion_ops.p2g_camera_params)
 * out [B, C, bev, bev] is written completely (zeros where no pillar). */
int isf_p2g_forward(const float* pillars, int pillar_ld, int slots, const int32_t* pillar_coors, int num_pillars,
                    const float* img_nhwc, int batch_size, int num_cam, int feat_h, int feat_w, int channels,
                    const float* cam_params, int input_h, int input_w, int bev_size, float* out,
                    isf_stream_t stream);
/* ... with the result as ONE split-format token matrix [B*bev*bev, 256] (token = (b*bev + y)*bev + x; channels == 256; the
 * same bytes as the fp32 map, written completely): what conv_fusion reads (dense grid convolution on
 * isf_sparse_conv_forward_f16x3) -- no isf_nchw_to_split pass, and a pillar's 256 values leave the wave as one KiB. */
int isf_p2g_forward_split(const float* pillars, int pillar_ld, int slots, const int32_t* pillar_coors, int num_pillars,
                          const float* img_nhwc, int batch_size, int num_cam, int feat_h, int feat_w, int channels,
                          const float* cam_params, int input_h, int input_w, int bev_size, void* out_split,
                          isf_stream_t stream);

/* A12  instance mining: sigmoid + 3x3 NMS + top-k -------------------------------------------------------
 * replaces fusion_encoder.py:1100-1131.  heatmap [B, K, H, W] logits; classes with bit set in
 * pool1_class_mask use a 1x1 pool (every cell is its own maximum).  top_index [B, k] = flat index % (H*W),
 * top_index_raw [B, k] = flat index over (K, H, W), both in descending score order (ties: ascending index);
 * masked_heatmap [B, K*H*W] optional (NULL to skip). */
int isf_instance_topk(const float* heatmap, int batch_size, int num_classes, int height, int width, int k,
                      unsigned pool1_class_mask, int32_t* top_index, int32_t* top_index_raw, float* masked_heatmap,
                      isf_stream_t stream);

/* A13  multi-scale deformable attention with the op signature of mmcv -----------------------------------------
 * replaces ext_module.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
 *   sampling_locations, attention_weights, im2col_step)
 *   (mmdet3d/models/middle_encoders/multi_scale_deformable_attn_function.py:118-124; kernel
 *   ms_deform_im2col_cuda.cuh:237-299); im2col_step only blocks the reference's batch loop and has no counterpart.
 * value [B, num_keys, heads, hd]; spatial_shapes [L, 2] int64 (h, w); level_start_index [L] int64;
 * sampling_loc [B, Q, heads, L, P, 2] (x, y) in [0, 1]; attn_weight [B, Q, heads, L, P] (post-softmax);
 * out [B, Q, heads*hd].  Any head_dim / level / point count. */
int isf_ms_deform_attn_forward(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                               const float* sampling_loc, const float* attn_weight, int batch_size, int num_keys,
                               int num_heads, int head_dim, int num_queries, int num_levels, int num_points,
                               float* out, isf_stream_t stream);

/* ... and its backward, with the op's own contract:
 * replaces ext_module.ms_deform_attn_backward(value, value_spatial_shapes, value_level_start_index,
 *   sampling_locations, attention_weights, grad_output, grad_value, grad_sampling_loc, grad_attn_weight, im2col_step)
 *   (multi_scale_deformable_attn_function.py:150-160; kernels ms_deform_im2col_cuda.cuh:301-920).
 * grad_output [B, Q, heads*hd]; grad_value [B, num_keys, heads, hd], grad_sampling_loc [B, Q, heads, L, P, 2] and
 * grad_attn_weight [B, Q, heads, L, P] are ZEROED BY THE CALLER (as the reference does, :142-144) and accumulated
 * into.  isf_msda_backward is the fused single-level form the mirror's InsContextAtt trains through. */
int isf_ms_deform_attn_backward(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                const float* sampling_loc, const float* attn_weight, const float* grad_output,
                                int batch_size, int num_keys, int num_heads, int head_dim, int num_queries,
                                int num_levels, int num_points, float* grad_value, float* grad_sampling_loc,
                                float* grad_attn_weight, isf_stream_t stream);

/* A10  in-group (in-window) indices -------------------------------------------------------------------------
 * replaces TorchEx ingroup_indices.forward(group_inds, out_inds)
 *   (mmdet3d/ops/TorchEx/torchex/src/ingroup_inds/ingroup_inds.cpp:24-54, ingroup_inds_kernel.cu:17-31; called by
 *   get_inner_win_inds_cuda, ops/sst/sst_ops.py:197-211): out_inds[i] = a distinct number in [0, count(group of i))
 *   for every element.  The reference hands the numbers out with atomicAdd (any order); here out_inds[i] is the
 *   number of EARLIER elements of the same group -- one of the reference's possible results, and deterministic.
 * group_inds / out_inds [num] int64 (torch.long, as the reference op takes them). */
int isf_ingroup_indices(const int64_t* group_inds, int num, int64_t* out_inds, isf_stream_t stream);

/* A13  fused single-level form used by the InsContextAtt path -------------------------------------------------
 * replaces MultiScaleDeformableAttnFunction.forward (mmdet3d/ops/.../ms_deform_attn, called at
 * fusion_encoder.py:597) plus the softmax / location arithmetic of :585-596.
 * value [B, H*W, heads*hd]; sampling_offsets [B*Q, heads*P*2]; attention_logits [B*Q, heads*P] (pre-softmax);
 * reference_points [B*Q, 2] (x, y) in [0, 1]; out [B*Q, heads*hd]. */
int isf_msda_forward(const float* value, const float* sampling_offsets, const float* attention_logits,
                     const float* reference_points, int batch_size, int num_queries, int num_heads, int head_dim,
                     int num_points, int height, int width, float* out, isf_stream_t stream);

/* 8f #1  box decoding (detection-head post-processing) ---------------------------------------------------
 * replaces TransFusionHeadV2.get_bboxes with nms_type=None (dense_heads/transfusion_head_v2.py:1278-1312,1344-1418)
 * including TransFusionBBoxCoder.decode(filter=True) (core/bbox/coders/transfusion_bbox_coder.py:39-124).
 * heatmap / query_score [B, classes, ld] (the first num_proposals columns of each row are used; heatmap = logits),
 * query_labels [B, P] int64, center [B,2,ld] (BEV cells), height [B,1,ld], dim [B,3,ld] (log metres), rot [B,2,ld]
 * (sin, cos), vel [B,2,ld] or NULL.  coder = 12 HOST floats: out_size_factor*voxel_size (x, y), pc_range (x, y),
 * post_center_range (6), score_threshold, apply_threshold (0/1; the reference tests `if self.score_threshold:`, so 0.0
 * is not applied).  Outputs, compacted per sample in proposal order: boxes [B, P, 9 | 7 without vel]
 * (x, y, z_bottom, dx, dy, dz, yaw, vx, vy), scores [B, P], labels [B, P] int32, counts [B] = boxes kept. */
int isf_decode_boxes(const float* heatmap, const float* query_score, const int64_t* query_labels, const float* center,
                     const float* height, const float* dim, const float* rot, const float* vel, int batch_size,
                     int num_classes, int num_proposals, int ld, const float* coder, float* boxes, float* scores,
                     int32_t* labels, int32_t* counts, isf_stream_t stream);

/* 8f #1  proposal initialisation / prediction outputs of the detection head ------------------------------------
 * isf_head_query_init replaces the tensor ops of TransFusionHeadV2.forward_single between the top-k and the first
 *   decoder layer (dense_heads/transfusion_head_v2.py:806-842): query_labels = top // HW, query_pos = bev_pos[top % HW],
 *   query = feature column + class_encoding(one_hot(label)), the first layer's self_posembed(query_pos) as a row of a
 *   per-cell table, query + position, and query_heatmap_score = heatmap.gather(top % HW) (:888-890).
 *   top_index / top_raw [B, P] int32 as isf_instance_topk writes them; feat_tok [B*HW, E] token-major;
 *   tok_of_cell [HW] int64 (BEV cell -> token row inside a sample) or NULL = identity; class_table [classes, E] =
 *   class_encoding.weight[:, c] + bias; qpe_table [HW, E] or NULL (then qpe / x are not written); bev_pos [HW, 2];
 *   masked [B, classes * HW] = the suppressed heat-map.  Outputs: query / qpe / x [B*P, E], query_pos [B, P, 2],
 *   top_index64 / query_labels [B, P] int64, query_score [B, classes, P].
 * isf_head_scatter_predictions replaces the per-output transposes of FFN.forward's results (:505-590), `center +=
 *   query_pos` (:883) and `query_pos = center.detach().clone()` (:885): output h = columns [col0[h], col0[h] +
 *   channels[h]) of the token-major block src[h] [B*P, src_ld[h]] -> dst[h] [B, channels[h], P]; center_head (or -1)
 *   gets query_pos [B, P, 2] added and is copied to query_pos_next [B, P, 2] (or NULL).  The four arrays and src / dst
 *   are HOST arrays of num_heads (<= 8) entries.  Both asynchronous. */
int isf_head_query_init(const int32_t* top_index, const int32_t* top_raw, int batch_size, int num_proposals, int hw,
                        int embed, int num_classes, const float* feat_tok, const int64_t* tok_of_cell,
                        const float* class_table, const float* qpe_table, const float* bev_pos, const float* masked,
                        float* query, float* qpe, float* x, float* query_pos, int64_t* top_index64,
                        int64_t* query_labels, float* query_score, isf_stream_t stream);
int isf_head_scatter_predictions(int num_heads, const float* const* src, const int* src_ld, const int* col0,
                                 const int* channels, float* const* dst, int center_head, const float* query_pos,
                                 float* query_pos_next, int batch_size, int num_proposals, isf_stream_t stream);

/* A12/A13  the mined instances' features and positions ---------------------------------------------------------
 * replaces the tensor ops of ISFusionEncoder.instance_fusion after the top-k (middle_encoders/fusion_encoder.py:
 *   1133-1141: x_ins = x_scene.gather(top), query_pos = bev_pos.gather(top)) and InsContextAtt.forward's preamble
 *   (:800-812: positions / bev_size, query_pos_embed, features + embedding).
 * top [B, Q] int32: flat cells y'*S + x' of the TRANSPOSED map (isf_instance_topk on the transposed heat-map);
 * scene [B, E, S, S] in the un-transposed orientation (cell x'*S + y'); qpe_table [S*S, E] = query_pos_embed of every
 * create_2D_grid cell.  Outputs: top64 / cell64 [B, Q] int64 (the transposed / un-transposed cell), tokens, qpe,
 * tokens_pos = tokens + qpe [B*Q, E], query_pos [B, Q, 2] = (x' + .5, y' + .5), ref = query_pos / S.  Asynchronous. */
int isf_instance_gather(const int32_t* top, int batch_size, int num_instances, int bev_size, int embed,
                        const float* scene, const float* qpe_table, int64_t* top64, int64_t* cell64, float* tokens,
                        float* qpe, float* tokens_pos, float* query_pos, float* ref, isf_stream_t stream);

/* 8f #2  sparse convolution backward ------------------------------------------------------------------------
 * replaces sparse_conv_ext.indice_conv_backward_fp32(features, filters, out_bp, indice_pairs, indice_num, inverse,
 *   subm) -> [input_bp, filters_bp]   (spconv_ops.h:363-456; SparseConvFunction.backward, functional.py:38-52).
 * isf_transpose_rulebook: nbr_t [K, nbr_t_stride] with nbr_t[k][j] = o  <=>  nbr[k][o] = j (-1 elsewhere);
 *   nbr_t_stride = isf_nbr_stride(num_in).  Built once per rulebook, shared by the convs that share it.
 * isf_sparse_conv_backward_input: grad_in [num_in, Cin] = sum_k grad_out[nbr_t[k][j], :] @ W[k]^T, every row written
 *   once (the forward kernel over the transposed table; no scatter-add, no atomics).
 * isf_sparse_conv_backward_filter: grad_filters [K, Cin, Cout] = sum_o features[nbr[k][o], :]^T grad_out[o, :]
 *   (fp32 MFMA, deterministic two-pass reduction over row chunks).  filters in the reference layout [K, Cin, Cout].
 * All asynchronous. */
int isf_transpose_rulebook(const int32_t* nbr, int nbr_stride, int num_out, int num_taps, int num_in,
                           int32_t* nbr_t, int nbr_t_stride, isf_stream_t stream);
int isf_sparse_conv_backward_input(const float* grad_out, int num_out, int c_out, const float* filters, int num_taps,
                                   int c_in, const int32_t* nbr_t, int nbr_t_stride, int num_in, float* grad_in,
                                   isf_stream_t stream);
int isf_sparse_conv_backward_filter(const float* features, int num_in, int c_in, const float* grad_out, int num_out,
                                    int c_out, const int32_t* nbr, int nbr_stride, int num_taps,
                                    float* grad_filters, isf_stream_t stream);
/* Round 5: the FILTER gradient on the f16 matrix cores (isf_spconv_wgrad16.hip), the half instantiation of the reference's
 * indice_conv_backward (spconv_ops.h:363-456 with T = half, all.cc:35-51) carried at fp32 accuracy by the f16x3 split.
 * isf_rulebook_pair_lists: the neighbour table -> the spconv-1 interchange format the reference's backward walks,
 *   indice_pairs [K, 2, capacity] (input rows, output rows of a tap's pairs, ordered by output row, -1 padded for at
 *   least 32 entries past the count) + indice_num [K]; capacity = isf_pair_list_capacity(num_in, num_out) (a multiple of
 *   32).  Same content as isf_rulebook_to_indice_pairs (whose second dimension is num_in), built by a two-pass ordered
 *   compaction over 2048-row blocks instead of one workgroup per tap.  Once per rulebook.
 * isf_grad_to_split: grad [n] fp32 -> split rows of grad * s, s = the power of two that brings max|grad| into
 *   [2^9, 2^10) (non-finite entries do not set it); scale_out (device float[2]) = {s, 1 / s}.  Gradients of 1e-6 .. 1e-9
 *   would otherwise lie in f16's subnormal range.  No host sync.
 * isf_grad_rescale: the same scale with fp32 rows out (out = grad * s; the fused linear's dX GEMM splits its input itself):
 *   two launches instead of the nine torch ops of the round-2 form (_lib.pow2_rescale), ~50 calls per training step.
 * isf_split_to_f32_scaled: split rows -> fp32 * *mul (device scalar, NULL = 1): the way back for dX.
 * isf_sparse_conv_backward_filter_f16x3: grad_filters [K, Cin, Cout] = (*grad_inv_scale) * sum over the pairs of tap k of
 *   x[in]^T dY[out], x and dY in the split format (x: what the forward pass stored; dY: isf_grad_to_split), products as
 *   x_lo*g_hi + x_hi*g_lo + x_hi*g_hi on v_mfma_f32_16x16x32_f16 with fp32 accumulation; Cin, Cout in {32, 64, 128, 256};
 *   pair chunks -> partial blocks -> ordered second pass: deterministic.  mode 0 = that (fp32-class); mode 1 = SINGLE-PASS
 *   f16: only the hi halves are read and multiplied -- fp16 operands, fp32 accumulation: the arithmetic of the reference's
 *   indice_conv_backward<at::Half>, which is what its sparse convolutions run under autocast (functional.py:24
 *   custom_fwd(cast_inputs=torch.half)); the forward / dX counterpart is mode 1 of isf_sparse_conv_forward_f16x3 / _dma.
 *   mode + 2: every tap's pair list is full (the rulebook of a dense grid, dense_train.py) -- larger reduction chunks (same
 *   sums, chunk boundaries move; still deterministic).  All asynchronous. */
int isf_pair_list_capacity(int num_in, int num_out);
int isf_rulebook_pair_lists(const int32_t* nbr, int nbr_stride, int num_out, int num_taps, int capacity,
                            int32_t* indice_pairs, int32_t* indice_num, isf_stream_t stream);
int isf_grad_to_split(const float* grad, size_t num_elems, void* grad_split, float* scale_out, isf_stream_t stream);
int isf_grad_rescale(const float* grad, size_t num_elems, float* out, float* scale_out, isf_stream_t stream);
int isf_split_to_f32_scaled(const void* xs, size_t num_elems, const float* mul, float* x, isf_stream_t stream);
int isf_sparse_conv_backward_filter_f16x3(const void* features_split, int num_in, int c_in, const void* grad_out_split,
                                          int num_out, int c_out, const int32_t* indice_pairs, const int32_t* indice_num,
                                          int capacity, int num_taps, const float* grad_inv_scale, float* grad_filters,
                                          int mode, isf_stream_t stream);

/* Round 5: BatchNorm1d with BATCH statistics (+ residual, + ReLU) on [N, C] fp32 rows, forward and backward
 * (isf_bn_train.hip) -- the norm / activation of the reference's sparse blocks and DynamicVFE layers in TRAINING mode:
 * nn.BatchNorm1d / naiveSyncBN1d.forward (ops/norm.py:136-211) + ReLU (+ the identity add of SparseBasicBlock,
 * ops/sparse_block.py:117-134), and what autograd derives for them.  channels = 4 * a divisor of 256.
 *   isf_bn1d_stats: stats [2C] = (sum x, sum x^2) per channel (partial sums per block -> ordered second level:
 *     deterministic).  The caller all-reduces stats (and the row count) across ranks for sync-BN.
 *   isf_bn1d_apply: y = relu?(x * gamma * invstd + beta - mean * gamma * invstd + residual?), mean / var from stats /
 *     count; running_mean / running_var (may be NULL) += momentum * (batch - running), the variance unbiased (nn.BatchNorm1d)
 *     or biased (naiveSyncBN1d) by unbiased_running_var; mean_invstd [2C] saved for the backward pass.
 *   isf_bn1d_backward_sums: sums [2C] = (sum g, sum g * xhat), g = grad_y masked by y_relu > 0 (y_relu NULL: no ReLU).
 *   isf_bn1d_backward_apply: grad_x = gamma * invstd * (g - sum_g / count - xhat * sum_gx / count); grad_residual (may be
 *     NULL) = g; grad_gamma = sum_gx, grad_beta = sum_g (may be NULL).  All asynchronous. */
int isf_bn1d_stats(const float* x, int num_rows, int channels, float* stats, isf_stream_t stream);
int isf_bn1d_apply(const float* x, int num_rows, int channels, const float* stats, float count, const float* gamma,
                   const float* beta, float eps, float momentum, int unbiased_running_var, float* running_mean,
                   float* running_var, const float* residual, int relu, float* y, float* mean_invstd, isf_stream_t stream);
/* The same two with the sums taken ABOUT A PIVOT ROW (pivot [C], device; NULL = 0): stats = (sum (x - p), sum (x - p)^2),
 * mean = p + sum / count, var = sumsq / count - (sum / count)^2 -- any pivot gives the same statistics in exact arithmetic,
 * a pivot near the mean (the host mirror passes the batch's first row) keeps fp32 from cancelling when |mean| >> std, as
 * torch's native batch_norm does with its two-pass variance (the single-process nn.BatchNorm1d case; the multi-rank
 * naiveSyncBN keeps pivot NULL = the reference's E[x^2] - E[x]^2 of ops/norm.py:186-190). */
int isf_bn1d_stats_pivot(const float* x, int num_rows, int channels, const float* pivot, float* stats, isf_stream_t stream);
int isf_bn1d_apply_pivot(const float* x, int num_rows, int channels, const float* stats, const float* pivot, float count,
                         const float* gamma, const float* beta, float eps, float momentum, int unbiased_running_var,
                         float* running_mean, float* running_var, const float* residual, int relu, float* y,
                         float* mean_invstd, isf_stream_t stream);
/* ... which also adds 1 to the module's num_batches_tracked (int64 device scalar, NULL = none) inside the same launch */
int isf_bn1d_apply_pivot_counted(const float* x, int num_rows, int channels, const float* stats, const float* pivot, float count,
                                 const float* gamma, const float* beta, float eps, float momentum, int unbiased_running_var,
                                 float* running_mean, float* running_var, long long* num_batches_tracked,
                                 const float* residual, int relu, float* y, float* mean_invstd, isf_stream_t stream);
int isf_bn1d_backward_sums(const float* grad_y, const float* x, const float* y_relu, int num_rows, int channels,
                           const float* mean_invstd, float* sums, isf_stream_t stream);
int isf_bn1d_backward_apply(const float* grad_y, const float* x, const float* y_relu, int num_rows, int channels,
                            const float* mean_invstd, const float* gamma, const float* sums, float count, float* grad_x,
                            float* grad_residual, float* grad_gamma, float* grad_beta, isf_stream_t stream);

/* 8f #3  input pre-pass: multi-sweep assembly + augmentation + range filter ---------------------------------
 * replaces, per batch, the dataloader-side numpy / torch code of LoadPointsFromMultiSweeps.__call__
 * (datasets/pipelines/loading.py:860-903), the point side of GlobalRotScaleTransV2 and RandomFlip3DV2
 * (datasets/pipelines/transforms_3d.py:1887-1890, :1171-1183) and PointsRangeFilter (:2012-2025).
 * raw = the sweep files of the batch as loaded (flat float32 [P, 5]: x, y, z, intensity, ring), uploaded untouched.
 * One descriptor per file, grouped by ascending sample, key frame first within a sample (the order of the
 * reference's concatenation).  Output: the kept points of the batch, compacted in input order, float32 [<= P, 5]
 * (capacity P rows), and sample_offsets[batch+1] (rows; device, and host when sample_offsets_host != NULL).
 * The call synchronizes `stream` once. */
typedef struct {
  int64_t first_point;     /* row of the file's first point in raw */
  int32_t num_points;
  int32_t sample;          /* batch index */
  int32_t is_sweep;        /* 0: key frame (time column := 0)   1: previous sweep (pose applied, time := time_lag) */
  int32_t remove_close;    /* sweeps only: drop points with |x| < r and |y| < r before the pose (loading.py:824-844) */
  float close_radius;
  float time_lag;          /* float32(key timestamp - sweep timestamp / 1e6) */
  double rotation[9];      /* sensor2lidar_rotation, row-major; p @ R^T in float64 as numpy does (loading.py:883) */
  double translation[3];   /* sensor2lidar_translation, float64 (loading.py:885) */
} isf_sweep_t;

typedef struct {
  int32_t enabled;
  float rot_mat_T[9];      /* points[:, :3] @ rot_mat_T, float32 (core/points/base_points.py:178) */
  float translation[3];    /* then += translation (:206) */
  float scale;             /* then *= scale (:270) */
  int32_t flip_horizontal; /* then y := -y (core/points/lidar_points.py:31-32) */
  int32_t flip_vertical;   /* and x := -x (:33-34) */
} isf_point_aug_t;

int isf_assemble_points(const float* raw, const isf_sweep_t* sweeps, int num_sweeps, int batch_size,
                        const isf_point_aug_t* aug /* [batch] or NULL */,
                        const float* point_range /* 6 host floats or NULL */, float* points_out,
                        int32_t* sample_offsets, int32_t* sample_offsets_host, isf_stream_t stream);

/* 8f #2  Point-to-Grid backward ---------------------------------------------------------------------------------
 * replaces autograd through img_fv_to_bev's F.grid_sample (fusion_encoder.py:1049-1056) + the canvas scatter (:989-1003):
 * grad_out [B, C, bev, bev] -> grad_img_nhwc [B*num_cam, H, W, C] (written completely; fp32 atomics inside).  The
 * other arguments are isf_p2g_forward's.  No gradient flows to points / calibration (not trainable). */
int isf_p2g_backward(const float* pillars, int pillar_ld, int slots, const int32_t* pillar_coors, int num_pillars,
                     int batch_size, int num_cam, int feat_h, int feat_w, int channels, const float* cam_params,
                     int input_h, int input_w, int bev_size, const float* grad_out, float* grad_img_nhwc,
                     isf_stream_t stream);

/* 8f #2  multi-scale deformable attention backward (one level) --------------------------------------------------
 * replaces MultiScaleDeformableAttnFunction.backward -> ms_deform_attn_backward (ops/src/cuda/ms_deform_attn_cuda.cu,
 * ms_deform_im2col_cuda.cuh:301-920) plus the backward of the softmax / location arithmetic isf_msda_forward folds in.
 * Same tensors as the forward plus grad_out [B*Q, heads*hd]; outputs grad_value [B, H*W, heads*hd] (zeroed here,
 * fp32 atomics), grad_offsets [B*Q, heads*P*2], grad_logits [B*Q, heads*P].  No gradient for reference_points
 * (the IS-Fusion path feeds constants).  Asynchronous. */
int isf_msda_backward(const float* value, const float* sampling_offsets, const float* attention_logits,
                      const float* reference_points, const float* grad_out, int batch_size, int num_queries,
                      int num_heads, int head_dim, int num_points, int height, int width, float* grad_value,
                      float* grad_offsets, float* grad_logits, isf_stream_t stream);

/* 8f #2  attention backward ------------------------------------------------------------------------------------
 * replaces autograd through nn.MultiheadAttention's softmax(q k^T / sqrt(hd)) v core (sst_basic_block_v2.py:41-75,
 * fusion_encoder.py:371-470) for the layouts of isf_attention_forward / isf_window_attention_forward.  The Lq x Lk
 * probabilities are recomputed (row statistics logsumexp and dO.O), never stored; every gradient row is written
 * once, no atomics.  `out` = the forward result.  head dim 16 (window: 16 or 32).  Asynchronous. */
int isf_attention_backward(const float* q, int ldq, const float* k, const float* v, int ldkv, const float* out,
                           const float* grad_out, int ldo, int batch_size, int num_queries, int num_keys,
                           int embed_dims, int num_heads, float* grad_q, int ldgq, float* grad_k, float* grad_v,
                           int ldgkv, isf_stream_t stream);
/* ... of isf_attention_forward_dropout (same dropout_p and seed; `out` = that call's result) */
int isf_attention_backward_dropout(const float* q, int ldq, const float* k, const float* v, int ldkv, const float* out,
                                   const float* grad_out, int ldo, int batch_size, int num_queries, int num_keys,
                                   int embed_dims, int num_heads, float dropout_p, unsigned long long seed, float* grad_q,
                                   int ldgq, float* grad_k, float* grad_v, int ldgkv, isf_stream_t stream);
int isf_window_attention_backward(const float* qkv, const float* grad_out, int batch_size, int grid_size,
                                  int embed_dims, int num_heads, int window, int shift, float* grad_qkv,
                                  isf_stream_t stream);

/* A9 / A15  dense 3x3 BEV convolutions on the sparse-conv kernel ------------------------
 * replaces mmcv ConvModule / nn.Conv2d + BatchNorm2d + ReLU (fusion_encoder.py:862-960, backbones/second.py:126-165,
 * MIOpen Winograd + 2 elementwise kernels per layer).  A dense B x H x W grid is a sparse tensor with every cell
 * active: isf_dense_grid_rulebook writes its neighbour table arithmetically (tap k = ky*kernel_w + kx; with
 * transpose_taps the taps are enumerated (kx, ky), which is the convolution of the SPATIALLY TRANSPOSED map expressed
 * on the un-transposed tokens -- the bev_feats.permute(0,1,3,2) of fusion_encoder.py:1093 costs nothing);
 * isf_sparse_conv_forward_f16x3 then runs Conv + BN + ReLU (+ partial sums of > 256 input channels through its
 * residual input) in one launch.  nbr == NULL only queries out_hw_host.
 * isf_nchw_to_split / isf_split_to_nchw convert between [B, C, hw] fp32 maps and split-format token matrices
 * (token = b*hw + pos); x_channel_offset / channel_offset select a channel slice of the source map / destination
 * rows (<= 256 channels per call). */
int isf_dense_grid_rulebook(int batch_size, int height, int width, int kernel_h, int kernel_w, int stride, int padding,
                            int transpose_taps, int32_t* nbr, int nbr_stride, int out_hw_host[2],
                            isf_stream_t stream);
int isf_nchw_to_split(const float* x, int batch_size, int x_channels, int x_channel_offset, int channels, int hw,
                      void* out_split, int out_channels, int channel_offset, isf_stream_t stream);
int isf_split_to_nchw(const void* x_split, int batch_size, int channels, int hw, float* out, isf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ISF_HIP_H */
