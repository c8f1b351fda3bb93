// isf_input.hip -- SURVEY 8f #3: the point-cloud input pipeline as a GPU pre-pass.  Replaces, for a whole batch in three
// launches, what the reference's dataloader workers do per sample in numpy / torch on the CPU:
//   LoadPointsFromMultiSweeps.__call__   datasets/pipelines/loading.py:860-903  (time column, remove_close,
//                                         p @ R^T + t per previous sweep, concatenation)
//   GlobalRotScaleTransV2 (points)       datasets/pipelines/transforms_3d.py:1887-1890 (rotate, translate, scale)
//   RandomFlip3DV2 (points)              transforms_3d.py:1171-1183 / core/points/lidar_points.py:29-34
//   PointsRangeFilter                    transforms_3d.py:2012-2025 / core/points/base_points.py:224-229
// The raw sweep files (flat float32 [P, 5]) are uploaded untouched; the kept points come out compacted per sample in
// file order (key frame first), which is the order the reference's boolean-mask indexing produces -- hard
// voxelization and the float sums of the VFE depend on it.
//
// HBM-bound byte work (20 B in twice, <= 20 B out once per point), nothing here is a GEMM.  A workgroup owns 256
// consecutive points of ONE sweep: the 5120-byte span is read coalesced into LDS, every lane transforms its point,
// ballot + popcount give the in-block rank, kept points are packed in LDS and written back coalesced.
// pass 1 counts, an exclusive scan over the per-block counts places the blocks, pass 2 recomputes and writes
// (recomputing is cheaper than a round trip of the transformed points through HBM).
#include <algorithm>

#include "isf_common.h"

namespace isf {

constexpr int kInBlock = 256;   // points per workgroup
constexpr int kDim = 5;         // x, y, z, intensity, time

struct SweepDev {               // device copy of isf_sweep_t + the block prefix
  double rot[9];
  double trans[3];
  long long first_point;
  int num_points;
  int sample;
  int is_sweep;
  int remove_close;
  float close_radius;
  float time_lag;
  int block_begin;              // first workgroup of this sweep
  int pad_;
};

struct AugDev {
  int enabled;
  float rot_t[9];
  float trans[3];
  float scale;
  int flip_h, flip_v;
};

// -> keep flag; p[] holds the transformed point.  Arithmetic types follow the reference: float64 for the sensor pose
// (numpy promotes the float32 points against the float64 matrices and rounds on every store), float32 for the
// augmentation (torch ops on the float32 cloud).
__device__ __forceinline__ bool transform_point(float p[kDim], const SweepDev& sw, const AugDev* __restrict__ aug,
                                                const float* __restrict__ range, bool use_range) {
  bool keep = true;
  if (sw.is_sweep) {
    if (sw.remove_close) keep = !(fabsf(p[0]) < sw.close_radius && fabsf(p[1]) < sw.close_radius);
    const double x = p[0], y = p[1], z = p[2];
    const float rx = (float)(x * sw.rot[0] + y * sw.rot[1] + z * sw.rot[2]);
    const float ry = (float)(x * sw.rot[3] + y * sw.rot[4] + z * sw.rot[5]);
    const float rz = (float)(x * sw.rot[6] + y * sw.rot[7] + z * sw.rot[8]);
    p[0] = (float)((double)rx + sw.trans[0]);
    p[1] = (float)((double)ry + sw.trans[1]);
    p[2] = (float)((double)rz + sw.trans[2]);
    p[4] = sw.time_lag;
  } else {
    p[4] = 0.f;
  }
  if (aug) {
    const AugDev& a = aug[sw.sample];
    if (a.enabled) {
      const float x = p[0], y = p[1], z = p[2];
      // rotate (a float32 [P,3] x [3,3] matmul whose summation order the BLAS picks), then translate and scale as
      // separate, individually rounded float32 ops (no contraction across them)
      p[0] = __fmul_rn(__fadd_rn(x * a.rot_t[0] + y * a.rot_t[3] + z * a.rot_t[6], a.trans[0]), a.scale);
      p[1] = __fmul_rn(__fadd_rn(x * a.rot_t[1] + y * a.rot_t[4] + z * a.rot_t[7], a.trans[1]), a.scale);
      p[2] = __fmul_rn(__fadd_rn(x * a.rot_t[2] + y * a.rot_t[5] + z * a.rot_t[8], a.trans[2]), a.scale);
      if (a.flip_h) p[1] = -p[1];
      if (a.flip_v) p[0] = -p[0];
    }
  }
  if (use_range)
    keep = keep && p[0] > range[0] && p[1] > range[1] && p[2] > range[2] && p[0] < range[3] && p[1] < range[4] &&
           p[2] < range[5];
  return keep;
}

struct RangeArg { float r[6]; int use; };

template <bool WRITE>
__global__ __launch_bounds__(kInBlock) void assemble_kernel(const float* __restrict__ raw,
                                                            const SweepDev* __restrict__ sweeps, int num_sweeps,
                                                            const AugDev* __restrict__ aug, RangeArg rng,
                                                            uint32_t* __restrict__ block_counts,
                                                            const uint32_t* __restrict__ block_offsets,
                                                            float* __restrict__ out) {
  __shared__ float tile[kInBlock * kDim];
  __shared__ int wave_count[kInBlock / 64];
  const int blk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the sweep this workgroup belongs to: last sweep whose block_begin <= blk (sweeps without points own no block)
  int lo = 0, hi = num_sweeps - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (sweeps[mid].block_begin <= blk) lo = mid; else hi = mid - 1;
  }
  const SweepDev& sw = sweeps[lo];
  const int start = (blk - sw.block_begin) * kInBlock;
  const int n = min(kInBlock, sw.num_points - start);
  const float* src = raw + (sw.first_point + start) * (long long)kDim;
  for (int i = tid; i < n * kDim; i += kInBlock) tile[i] = src[i];
  __syncthreads();
  float p[kDim];
  bool keep = false;
  if (tid < n) {
#pragma unroll
    for (int c = 0; c < kDim; ++c) p[c] = tile[tid * kDim + c];
    keep = transform_point(p, sw, aug, rng.r, rng.use != 0);
  }
  const unsigned long long mask = __ballot(keep);
  if (lane == 0) wave_count[wave] = __popcll(mask);
  __syncthreads();
  int before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kInBlock / 64; ++w) {
    if (w < wave) before += wave_count[w];
    total += wave_count[w];
  }
  if (!WRITE) {
    if (tid == 0) block_counts[blk] = (uint32_t)total;
    return;
  }
  // every lane has read its point: the tile can be reused for the packed output
  const int rank = before + __popcll(mask & ((1ull << lane) - 1ull));
  if (keep) {
#pragma unroll
    for (int c = 0; c < kDim; ++c) tile[rank * kDim + c] = p[c];
  }
  __syncthreads();
  float* dst = out + (size_t)block_offsets[blk] * kDim;
  for (int i = tid; i < total * kDim; i += kInBlock) dst[i] = tile[i];
}

__global__ void sample_offsets_kernel(const uint32_t* __restrict__ block_offsets, int num_blocks,
                                      const int* __restrict__ sample_first_block, int batch_size,
                                      int32_t* __restrict__ sample_offsets) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > batch_size) return;
  const int blk = sample_first_block[b];
  sample_offsets[b] = (int32_t)block_offsets[min(blk, num_blocks)];   // block_offsets[num_blocks] = total
}

}  // namespace isf

extern "C" {

int isf_assemble_points(const float* raw, const isf_sweep_t* sweeps, int num_sweeps, int batch_size,
                        const isf_point_aug_t* aug, const float* point_range, float* points_out,
                        int32_t* sample_offsets, int32_t* sample_offsets_host, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(num_sweeps >= 0 && batch_size > 0, ISF_ERR_ARG, "assemble_points: bad sizes");
  ISF_REQUIRE(sample_offsets, ISF_ERR_ARG, "assemble_points: null sample_offsets");
  hipStream_t st = as_stream(stream);
  std::vector<SweepDev> dev(num_sweeps);
  std::vector<int> first_block(batch_size + 1, 0);
  long long blocks = 0;
  int prev_sample = 0;
  for (int i = 0; i < num_sweeps; ++i) {
    const isf_sweep_t& s = sweeps[i];
    ISF_REQUIRE(s.num_points >= 0 && s.first_point >= 0, ISF_ERR_ARG, "assemble_points: sweep %d has a bad extent", i);
    ISF_REQUIRE(s.sample >= prev_sample && s.sample < batch_size, ISF_ERR_ARG,
                "assemble_points: sweeps must be grouped by ascending sample (sweep %d: sample %d)", i, s.sample);
    for (int b = prev_sample + 1; b <= s.sample; ++b) first_block[b] = (int)blocks;
    prev_sample = s.sample;
    SweepDev& d = dev[i];
    for (int k = 0; k < 9; ++k) d.rot[k] = s.rotation[k];
    for (int k = 0; k < 3; ++k) d.trans[k] = s.translation[k];
    d.first_point = s.first_point;
    d.num_points = s.num_points;
    d.sample = s.sample;
    d.is_sweep = s.is_sweep;
    d.remove_close = s.remove_close;
    d.close_radius = s.close_radius;
    d.time_lag = s.time_lag;
    d.block_begin = (int)blocks;
    d.pad_ = 0;
    blocks += ceil_div(s.num_points, kInBlock);
    ISF_REQUIRE(blocks < (1ll << 30), ISF_ERR_ARG, "assemble_points: too many points");
  }
  for (int b = prev_sample + 1; b <= batch_size; ++b) first_block[b] = (int)blocks;
  const int num_blocks = (int)blocks;
  if (num_blocks == 0) {
    ISF_HIP_TRY(hipMemsetAsync(sample_offsets, 0, sizeof(int32_t) * (batch_size + 1), st));
    if (sample_offsets_host) std::fill(sample_offsets_host, sample_offsets_host + batch_size + 1, 0);
    return ISF_OK;
  }
  ISF_REQUIRE(raw && points_out, ISF_ERR_ARG, "assemble_points: null pointer");
  // sweeps that own no block must not win the binary search: give them the block_begin of the next sweep (already so:
  // block_begin is a prefix and the search takes the LAST sweep with block_begin <= blk, i.e. the one that owns it,
  // provided trailing empty sweeps are cut)
  int live = num_sweeps;
  while (live > 0 && dev[live - 1].num_points == 0) --live;

  Arena& a = arena_for_stream(as_stream(stream));
  ISF_TRY(a.reset());
  SweepDev* d_sweeps;
  AugDev* d_aug = nullptr;
  int* d_first;
  uint32_t *d_counts, *d_offsets;
  ISF_TRY(a.alloc_n(&d_sweeps, (size_t)live));
  ISF_TRY(a.alloc_n(&d_first, (size_t)batch_size + 1));
  ISF_TRY(a.alloc_n(&d_counts, (size_t)num_blocks));
  ISF_TRY(a.alloc_n(&d_offsets, (size_t)num_blocks + 1));   // the scan writes n + 1 entries (out[n] = total)
  ISF_HIP_TRY(hipMemcpyAsync(d_sweeps, dev.data(), sizeof(SweepDev) * live, hipMemcpyHostToDevice, st));
  ISF_HIP_TRY(hipMemcpyAsync(d_first, first_block.data(), sizeof(int) * (batch_size + 1), hipMemcpyHostToDevice, st));
  std::vector<AugDev> haug;
  if (aug) {
    haug.resize(batch_size);
    for (int b = 0; b < batch_size; ++b) {
      AugDev& h = haug[b];
      h.enabled = aug[b].enabled;
      for (int k = 0; k < 9; ++k) h.rot_t[k] = aug[b].rot_mat_T[k];
      for (int k = 0; k < 3; ++k) h.trans[k] = aug[b].translation[k];
      h.scale = aug[b].scale;
      h.flip_h = aug[b].flip_horizontal;
      h.flip_v = aug[b].flip_vertical;
    }
    ISF_TRY(a.alloc_n(&d_aug, (size_t)batch_size));
    ISF_HIP_TRY(hipMemcpyAsync(d_aug, haug.data(), sizeof(AugDev) * batch_size, hipMemcpyHostToDevice, st));
  }
  RangeArg rng;
  rng.use = point_range != nullptr;
  for (int k = 0; k < 6; ++k) rng.r[k] = point_range ? point_range[k] : 0.f;

  hipLaunchKernelGGL((assemble_kernel<false>), dim3(num_blocks), dim3(kInBlock), 0, st, raw, d_sweeps, live, d_aug,
                     rng, d_counts, (const uint32_t*)nullptr, (float*)nullptr);
  ISF_LAUNCH_CHECK();
  ISF_TRY(scan_u32_exclusive(a, d_counts, d_offsets, (size_t)num_blocks, st));
  hipLaunchKernelGGL((assemble_kernel<true>), dim3(num_blocks), dim3(kInBlock), 0, st, raw, d_sweeps, live, d_aug, rng,
                     d_counts, d_offsets, points_out);
  ISF_LAUNCH_CHECK();
  hipLaunchKernelGGL(sample_offsets_kernel, dim3(ceil_div(batch_size + 1, 64)), dim3(64), 0, st, d_offsets, num_blocks,
                     d_first, batch_size, sample_offsets);
  ISF_LAUNCH_CHECK();
  // ONE stream sync per batch: the caller needs the per-sample counts to hand the points on (the reference's
  // pipeline is host code throughout), and the pageable descriptor vectors above must outlive their uploads
  if (sample_offsets_host)
    ISF_HIP_TRY(hipMemcpyAsync(sample_offsets_host, sample_offsets, sizeof(int32_t) * (batch_size + 1),
                               hipMemcpyDeviceToHost, st));
  ISF_HIP_TRY(hipStreamSynchronize(st));
  return ISF_OK;
}

}  // extern "C"
