// isf_boundary.hip -- native-op entry points of the reference that the fused HSF / IGF path does not need itself (on a
// dense BEV grid window membership is arithmetic, and InsContextAtt runs the fused single-level kernel in isf_fusion.hip)
// but that a maintainer binding the reference's op layer expects to find:
//   * TorchEx `ingroup_indices.forward(group_inds, out_inds)`  mmdet3d/ops/TorchEx/torchex/src/ingroup_inds/ingroup_inds.cpp:24-54,
//     ingroup_inds_kernel.cu:17-31 (get_inner_win_inds of SST, ops/sst/sst_ops.py:197-211)
//   * mmcv `_ext.ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)`
//     (models/middle_encoders/multi_scale_deformable_attn_function.py:118-124; kernel ms_deform_im2col_cuda.cuh:237-299)
#include <hipcub/hipcub.hpp>

#include "isf_common.h"

namespace isf {

__global__ void iota_kernel(int32_t* __restrict__ v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = i;
}

// start[pos] = pos where a new group begins in the sorted key sequence, else 0 (an inclusive max-scan then carries the
// start of every element's segment)
__global__ void segment_starts_kernel(const int64_t* __restrict__ keys, int n, int32_t* __restrict__ start) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) start[i] = (i == 0 || keys[i] != keys[i - 1]) ? i : 0;
}

__global__ void ingroup_rank_kernel(const int32_t* __restrict__ seg_start, const int32_t* __restrict__ order, int n,
                                    int64_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[order[i]] = (int64_t)(i - seg_start[i]);
}

// one thread per (b, q, head, channel): channels of a head are consecutive lanes, so the four bilinear taps of a
// sampling point are coalesced 4*D-byte runs; the (level, point) loop is the reference's (ms_deform_im2col_cuda.cuh:262-296)
__global__ __launch_bounds__(256) void ms_deform_attn_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ spatial_shapes, const int64_t* __restrict__ level_start,
    const float* __restrict__ loc, const float* __restrict__ weight, int B, int S, int M, int D, int Q, int L, int P,
    float* __restrict__ out) {
  const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)B * Q * M * D;
  if (tid >= total) return;
  const int d = (int)(tid % D);
  const int m = (int)((tid / D) % M);
  const long long bq = tid / ((long long)D * M);
  const int b = (int)(bq / Q);
  const float* wq = weight + (bq * M + m) * (long long)L * P;
  const float* lq = loc + (bq * M + m) * (long long)L * P * 2;
  const size_t vstride = (size_t)M * D;
  float acc = 0.f;
  for (int l = 0; l < L; ++l) {
    const int H = (int)spatial_shapes[2 * l], W = (int)spatial_shapes[2 * l + 1];
    const float* vb = value + ((size_t)b * S + (size_t)level_start[l]) * vstride + (size_t)m * D + d;
    for (int p = 0; p < P; ++p) {
      const float lx = lq[(l * P + p) * 2], ly = lq[(l * P + p) * 2 + 1];
      const float w_im = lx * (float)W - 0.5f, h_im = ly * (float)H - 0.5f;
      if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
        const float fh = floorf(h_im), fw = floorf(w_im);
        const int h0 = (int)fh, w0 = (int)fw;
        const float lh = h_im - fh, lw = w_im - fw;
        float s = 0.f;
        if (h0 >= 0 && w0 >= 0) s += (1.f - lh) * (1.f - lw) * vb[((size_t)h0 * W + w0) * vstride];
        if (h0 >= 0 && w0 + 1 <= W - 1) s += (1.f - lh) * lw * vb[((size_t)h0 * W + w0 + 1) * vstride];
        if (h0 + 1 <= H - 1 && w0 >= 0) s += lh * (1.f - lw) * vb[((size_t)(h0 + 1) * W + w0) * vstride];
        if (h0 + 1 <= H - 1 && w0 + 1 <= W - 1) s += lh * lw * vb[((size_t)(h0 + 1) * W + w0 + 1) * vstride];
        acc += wq[l * P + p] * s;
      }
    }
  }
  out[tid] = acc;
}

}  // namespace isf

extern "C" {

int isf_ingroup_indices(const int64_t* group_inds, int num, int64_t* out_inds, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(num >= 0, ISF_ERR_ARG, "ingroup_indices: bad size");
  if (num == 0) return ISF_OK;
  ISF_REQUIRE(group_inds && out_inds, ISF_ERR_ARG, "ingroup_indices: null pointer");
  hipStream_t st = as_stream(stream);
  Arena& a = arena_for_stream(as_stream(stream));
  ISF_TRY(a.reset());
  int64_t* keys = nullptr;
  int32_t *iota = nullptr, *order = nullptr, *start = nullptr, *seg = nullptr;
  ISF_TRY(a.alloc_n(&keys, (size_t)num));
  ISF_TRY(a.alloc_n(&iota, (size_t)num));
  ISF_TRY(a.alloc_n(&order, (size_t)num));
  ISF_TRY(a.alloc_n(&start, (size_t)num));
  ISF_TRY(a.alloc_n(&seg, (size_t)num));
  const dim3 grid(ceil_div(num, 256)), block(256);
  hipLaunchKernelGGL(iota_kernel, grid, block, 0, st, iota, num);
  size_t tb1 = 0, tb2 = 0;
  ISF_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb1, group_inds, keys, iota, order, num, 0, 64, st));
  ISF_HIP_TRY(hipcub::DeviceScan::InclusiveScan(nullptr, tb2, start, seg, hipcub::Max(), num, st));
  void* temp = nullptr;
  ISF_TRY(a.alloc(&temp, tb1 > tb2 ? tb1 : tb2));
  // stable LSD radix sort: equal group ids keep their input order, so the position inside a segment is the
  // first-come rank -- one of the numberings the reference's atomicAdd can produce, and always the same one
  ISF_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(temp, tb1, group_inds, keys, iota, order, num, 0, 64, st));
  hipLaunchKernelGGL(segment_starts_kernel, grid, block, 0, st, keys, num, start);
  ISF_HIP_TRY(hipcub::DeviceScan::InclusiveScan(temp, tb2, start, seg, hipcub::Max(), num, st));
  hipLaunchKernelGGL(ingroup_rank_kernel, grid, block, 0, st, seg, order, num, out_inds);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int isf_ms_deform_attn_forward(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                               const float* sampling_loc, const float* attn_weight, int batch_size, int num_keys,
                               int num_heads, int head_dim, int num_queries, int num_levels, int num_points,
                               float* out, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(batch_size >= 0 && num_keys >= 0 && num_heads > 0 && head_dim > 0 && num_queries >= 0 &&
                  num_levels > 0 && num_points > 0, ISF_ERR_ARG, "ms_deform_attn_forward: bad sizes");
  if (batch_size == 0 || num_queries == 0) return ISF_OK;
  ISF_REQUIRE(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && out, ISF_ERR_ARG,
              "ms_deform_attn_forward: null pointer");
  const long long total = (long long)batch_size * num_queries * num_heads * head_dim;
  hipLaunchKernelGGL(ms_deform_attn_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), value,
                     spatial_shapes, level_start_index, sampling_loc, attn_weight, batch_size, num_keys, num_heads,
                     head_dim, num_queries, num_levels, num_points, out);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

}  // extern "C"
