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

// backward of ms_deform_attn_kernel with the mmcv op's contract (multi_scale_deformable_attn_function.py:150-160 ->
// ms_deformable_col2im, ms_deform_im2col_cuda.cuh:301-920): grad_value / grad_sampling_loc / grad_attn_weight are
// ZEROED BY THE CALLER and accumulated into.  Same thread mapping as the forward: one thread per (b, q, head,
// channel); the bilinear taps scatter into grad_value with atomics (a value cell is hit by many queries), the location
// and weight gradients of a (query, head, level, point) are first summed over the head's channels inside the wave
// (channels are consecutive lanes: a power-of-two head_dim <= 64 is one shuffle tree) and added once.
template <bool TREE>
__global__ __launch_bounds__(256) void ms_deform_attn_backward_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ spatial_shapes, const int64_t* __restrict__ level_start,
    const float* __restrict__ loc, const float* __restrict__ weight, const float* __restrict__ grad_out, int B, int S,
    int M, int D, int Q, int L, int P, float* __restrict__ grad_value, float* __restrict__ grad_loc,
    float* __restrict__ grad_weight) {
  const long long tid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)B * Q * M * D;
  const bool live = tid < total;
  const long long t = live ? tid : total - 1;            // idle lanes shadow a live one: the shuffles stay uniform
  const int d = (int)(t % D);
  const int m = (int)((t / D) % M);
  const long long bq = t / ((long long)D * M);
  const int b = (int)(bq / Q);
  const long long qm = bq * M + m;
  const float* wq = weight + qm * (long long)L * P;
  const float* lq = loc + qm * (long long)L * P * 2;
  const size_t vstride = (size_t)M * D;
  const float go = live ? grad_out[t] : 0.f;
  for (int l = 0; l < L; ++l) {
    const int H = (int)spatial_shapes[2 * l], W = (int)spatial_shapes[2 * l + 1];
    const size_t vo = ((size_t)b * S + (size_t)level_start[l]) * vstride + (size_t)m * D + d;
    for (int p = 0; p < P; ++p) {
      const float lx = lq[(l * P + p) * 2], ly = lq[(l * P + p) * 2 + 1];
      const float aw = wq[l * P + p];
      const float w_im = lx * (float)W - 0.5f, h_im = ly * (float)H - 0.5f;
      float g_w = 0.f, g_x = 0.f, g_y = 0.f;
      if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
        const float fh = floorf(h_im), fw = floorf(w_im);
        const int h0 = (int)fh, w0 = (int)fw;
        const float lh = h_im - fh, lw = w_im - fw, hh = 1.f - lh, hw = 1.f - lw;
        const float tg = go * aw;                          // top_grad * attn_weight
        float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
        if (h0 >= 0 && w0 >= 0) {
          const size_t o = vo + ((size_t)h0 * W + w0) * vstride;
          v1 = value[o];
          if (live) atomicAdd(grad_value + o, hh * hw * tg);
        }
        if (h0 >= 0 && w0 + 1 <= W - 1) {
          const size_t o = vo + ((size_t)h0 * W + w0 + 1) * vstride;
          v2 = value[o];
          if (live) atomicAdd(grad_value + o, hh * lw * tg);
        }
        if (h0 + 1 <= H - 1 && w0 >= 0) {
          const size_t o = vo + ((size_t)(h0 + 1) * W + w0) * vstride;
          v3 = value[o];
          if (live) atomicAdd(grad_value + o, lh * hw * tg);
        }
        if (h0 + 1 <= H - 1 && w0 + 1 <= W - 1) {
          const size_t o = vo + ((size_t)(h0 + 1) * W + w0 + 1) * vstride;
          v4 = value[o];
          if (live) atomicAdd(grad_value + o, lh * lw * tg);
        }
        g_w = go * (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4);
        g_x = (float)W * tg * (-hh * v1 + hh * v2 - lh * v3 + lh * v4);
        g_y = (float)H * tg * (-hw * v1 - lw * v2 + hw * v3 + lw * v4);
      }
      if (TREE) {   // D lanes of one (query, head) are an aligned power-of-two lane group
        for (int s = D >> 1; s >= 1; s >>= 1) {
          g_w += __shfl_xor(g_w, s, 64);
          g_x += __shfl_xor(g_x, s, 64);
          g_y += __shfl_xor(g_y, s, 64);
        }
        if (live && d == 0) {
          atomicAdd(grad_weight + qm * (long long)L * P + l * P + p, g_w);
          atomicAdd(grad_loc + (qm * (long long)L * P + l * P + p) * 2, g_x);
          atomicAdd(grad_loc + (qm * (long long)L * P + l * P + p) * 2 + 1, g_y);
        }
      } else if (live) {
        atomicAdd(grad_weight + qm * (long long)L * P + l * P + p, g_w);
        atomicAdd(grad_loc + (qm * (long long)L * P + l * P + p) * 2, g_x);
        atomicAdd(grad_loc + (qm * (long long)L * P + l * P + p) * 2 + 1, g_y);
      }
    }
  }
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

int isf_ms_deform_attn_backward(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                const float* sampling_loc, const float* attn_weight, const float* grad_output,
                                int batch_size, int num_keys, int num_heads, int head_dim, int num_queries,
                                int num_levels, int num_points, float* grad_value, float* grad_sampling_loc,
                                float* grad_attn_weight, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(batch_size >= 0 && num_keys >= 0 && num_heads > 0 && head_dim > 0 && num_queries >= 0 &&
                  num_levels > 0 && num_points > 0, ISF_ERR_ARG, "ms_deform_attn_backward: bad sizes");
  if (batch_size == 0 || num_queries == 0) return ISF_OK;
  ISF_REQUIRE(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && grad_output &&
                  grad_value && grad_sampling_loc && grad_attn_weight, ISF_ERR_ARG,
              "ms_deform_attn_backward: null pointer");
  const long long total = (long long)batch_size * num_queries * num_heads * head_dim;
  const bool tree = head_dim <= 64 && (head_dim & (head_dim - 1)) == 0;
  if (tree)
    hipLaunchKernelGGL(ms_deform_attn_backward_kernel<true>, dim3(ceil_div(total, 256)), dim3(256), 0,
                       as_stream(stream), value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                       grad_output, batch_size, num_keys, num_heads, head_dim, num_queries, num_levels, num_points,
                       grad_value, grad_sampling_loc, grad_attn_weight);
  else
    hipLaunchKernelGGL(ms_deform_attn_backward_kernel<false>, dim3(ceil_div(total, 256)), dim3(256), 0,
                       as_stream(stream), value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                       grad_output, batch_size, num_keys, num_heads, head_dim, num_queries, num_levels, num_points,
                       grad_value, grad_sampling_loc, grad_attn_weight);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

}  // extern "C"
