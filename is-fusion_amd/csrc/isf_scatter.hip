// isf_scatter.hip -- A3 DynamicScatter forward/backward and the cfg-1 HardSimpleVFE mean.
//
// The reference sorts the coordinate rows (at::unique_dim) to number the voxels; here the voxel number
// of a coordinate is its rank in an occupancy bitmap spanning the coordinate extents, which yields the
// same lexicographically sorted numbering with one streaming scan instead of a radix sort.
#include <algorithm>

#include "isf_common.h"

namespace isf {

// grid-stride over the rows, one atomic triple per WORKGROUP of a bounded grid: three hot words take ~88 memory-side
// updates per microsecond, and one triple per wave of a P / 256-block launch was 28 k updates = 320 us per call at
// 600 k points (six calls per training step: profiles/r05_train_step_300k.txt)
__global__ __launch_bounds__(256) void sc_extent_kernel(const int32_t* __restrict__ coors, int P, int* __restrict__ ext) {
  int m0 = -1, m1 = -1, m2 = -1;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
    const int a = coors[(size_t)i * 3], b = coors[(size_t)i * 3 + 1], c = coors[(size_t)i * 3 + 2];
    if (a >= 0 && b >= 0 && c >= 0) { m0 = max(m0, a); m1 = max(m1, b); m2 = max(m2, c); }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    m0 = max(m0, __shfl_xor(m0, d, 64));
    m1 = max(m1, __shfl_xor(m1, d, 64));
    m2 = max(m2, __shfl_xor(m2, d, 64));
  }
  __shared__ int w[4][3];
  if ((threadIdx.x & 63) == 0) { w[threadIdx.x >> 6][0] = m0; w[threadIdx.x >> 6][1] = m1; w[threadIdx.x >> 6][2] = m2; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int m = max(max(w[0][threadIdx.x], w[1][threadIdx.x]), max(w[2][threadIdx.x], w[3][threadIdx.x]));
    if (m >= 0) atomicMax(&ext[threadIdx.x], m);
  }
}

__global__ void sc_mark_kernel(const int32_t* __restrict__ coors, int P, int H, int W,
                               unsigned long long* __restrict__ bits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int a = coors[(size_t)i * 3], b = coors[(size_t)i * 3 + 1], c = coors[(size_t)i * 3 + 2];
  if (a < 0 || b < 0 || c < 0) return;
  const unsigned long long cell = ((unsigned long long)a * H + b) * W + c;
  const unsigned long long bit = 1ull << (cell & 63);
  unsigned long long* p = bits + (cell >> 6);
  if (!(*p & bit)) atomicOr(p, bit);
}

__global__ void sc_coors_out_kernel(const int32_t* __restrict__ c4, const int* __restrict__ total,
                                    int32_t* __restrict__ out3) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= *total) return;
  const int4 c = reinterpret_cast<const int4*>(c4)[r];
  out3[(size_t)r * 3 + 0] = c.y;
  out3[(size_t)r * 3 + 1] = c.z;
  out3[(size_t)r * 3 + 2] = c.w;
}

__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
  // order-preserving integer views of IEEE floats: non-negative -> signed max, negative -> unsigned min
  if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__global__ void sc_fill_kernel(float* __restrict__ p, size_t n, float v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ void sc_map_count_kernel(const int32_t* __restrict__ coors, int P, int H, int W,
                                    const unsigned long long* __restrict__ bits,
                                    const uint32_t* __restrict__ prefix, int32_t* __restrict__ cmap,
                                    int32_t* __restrict__ count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int a = coors[(size_t)i * 3], b = coors[(size_t)i * 3 + 1], c = coors[(size_t)i * 3 + 2];
  int v = -1;
  if (a >= 0 && b >= 0 && c >= 0) {
    v = occ_lookup(bits, prefix, ((unsigned long long)a * H + b) * W + c);
    atomicAdd(&count[v], 1);
  }
  cmap[i] = v;
}

// one wave handles 64 (point, channel) items laid out channel-fastest, so lanes of one point hit one row
__global__ void sc_reduce_kernel(const float* __restrict__ feats, const int32_t* __restrict__ cmap,
                                 long long total, int C, int reduce, float* __restrict__ out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int i = (int)(t / C), k = (int)(t % C);
  const int v = cmap[i];
  if (v < 0) return;
  const float f = feats[t];
  float* o = out + (size_t)v * C + k;
  if (reduce == ISF_REDUCE_MAX) atomic_max_f32(o, f);
  else atomicAdd(o, f);
}

__global__ void sc_mean_div_kernel(float* __restrict__ out, const int32_t* __restrict__ count,
                                   long long total, int C) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  out[t] = __fdiv_rn(out[t], (float)count[t / C]);
}

int dynamic_scatter_forward_impl(Arena& a, const float* feats, const int32_t* coors, int P, int C,
                                 int reduce, float* reduced, int32_t* out_coors, int32_t* cmap,
                                 int32_t* count, int* M_host, hipStream_t st) {
  *M_host = 0;
  if (P <= 0) return ISF_OK;
  int* ext = nullptr;
  ISF_TRY(a.alloc_n(&ext, 64));
  ISF_HIP_TRY(hipMemsetAsync(ext, 0xff, 3 * sizeof(int), st));  // -1
  hipLaunchKernelGGL(sc_extent_kernel, dim3(std::min(512, ceil_div(P, 256))), dim3(256), 0, st, coors, P, ext);
  ISF_LAUNCH_CHECK();
  int h_ext[3];
  ISF_HIP_TRY(hipMemcpyAsync(h_ext, ext, sizeof(h_ext), hipMemcpyDeviceToHost, st));
  ISF_HIP_TRY(hipStreamSynchronize(st));
  if (h_ext[0] < 0) {  // every row invalid: empty output, map = -1
    ISF_HIP_TRY(hipMemsetAsync(cmap, 0xff, (size_t)P * sizeof(int32_t), st));
    return ISF_OK;
  }
  const long long D = h_ext[0] + 1ll, H = h_ext[1] + 1ll, W = h_ext[2] + 1ll;
  ISF_REQUIRE(D * H * W <= (1ll << 35), ISF_ERR_UNSUPPORTED,
              "dynamic_scatter: coordinate extent %lldx%lldx%lld too large for the bitmap index", D, H, W);
  OccIndex occ;
  ISF_TRY(occ_create(a, &occ, 1, (int)D, (int)H, (int)W, st));
  hipLaunchKernelGGL(sc_mark_kernel, dim3(ceil_div(P, 256)), dim3(256), 0, st, coors, P, (int)H, (int)W,
                     occ.bits);
  ISF_LAUNCH_CHECK();
  ISF_TRY(occ_scan(a, occ, st));
  int M = 0;
  ISF_TRY(read_int(occ.total, &M, st));
  *M_host = M;
  int32_t* c4 = nullptr;
  ISF_TRY(a.alloc_n(&c4, (size_t)M * 4));
  ISF_TRY(occ_compact_coords4(occ, c4, st));
  hipLaunchKernelGGL(sc_coors_out_kernel, dim3(ceil_div(M, 256)), dim3(256), 0, st, c4, occ.total,
                     out_coors);
  const size_t n_out = (size_t)M * C;
  if (reduce == ISF_REDUCE_MAX)
    hipLaunchKernelGGL(sc_fill_kernel, dim3(ceil_div((long long)n_out, 256)), dim3(256), 0, st, reduced,
                       n_out, -INFINITY);
  else
    ISF_HIP_TRY(hipMemsetAsync(reduced, 0, n_out * sizeof(float), st));
  ISF_HIP_TRY(hipMemsetAsync(count, 0, (size_t)M * sizeof(int32_t), st));
  hipLaunchKernelGGL(sc_map_count_kernel, dim3(ceil_div(P, 256)), dim3(256), 0, st, coors, P, (int)H,
                     (int)W, occ.bits, occ.prefix, cmap, count);
  const long long total = (long long)P * C;
  hipLaunchKernelGGL(sc_reduce_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, feats, cmap, total,
                     C, reduce, reduced);
  if (reduce == ISF_REDUCE_MEAN)
    hipLaunchKernelGGL(sc_mean_div_kernel, dim3(ceil_div((long long)n_out, 256)), dim3(256), 0, st,
                       reduced, count, (long long)n_out, C);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

// ---------------------------------------------------------------------------------------- backward
__global__ void sc_bwd_add_kernel(float* __restrict__ g, const float* __restrict__ gr,
                                  const int32_t* __restrict__ cmap, const int32_t* __restrict__ count,
                                  long long total, int C, int mean) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int v = cmap[t / C];
  float r = 0.f;
  if (v >= 0) {
    r = gr[(size_t)v * C + (t % C)];
    if (mean) r = __fdiv_rn(r, (float)count[v]);
  }
  g[t] = r;
}

__global__ void sc_bwd_argmax_kernel(const float* __restrict__ feats, const float* __restrict__ reduced,
                                     const int32_t* __restrict__ cmap, long long total, int C,
                                     int* __restrict__ from) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int i = (int)(t / C), k = (int)(t % C);
  const int v = cmap[i];
  if (v < 0) return;
  if (feats[t] == reduced[(size_t)v * C + k]) atomicMin(&from[(size_t)v * C + k], i);
}

__global__ void sc_bwd_max_scatter_kernel(float* __restrict__ g, const float* __restrict__ gr,
                                          const int* __restrict__ from, long long total_mc, int C, int P) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total_mc) return;
  const int i = from[t];
  if (i < P) g[(size_t)i * C + (t % C)] = gr[t];
}

__global__ void sc_fill_int_kernel(int* __restrict__ p, size_t n, int v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

int dynamic_scatter_backward_impl(Arena& a, float* g, const float* gr, const float* feats,
                                  const float* reduced, const int32_t* cmap, const int32_t* count, int P,
                                  int M, int C, int reduce, hipStream_t st) {
  if (P <= 0) return ISF_OK;
  const long long total = (long long)P * C;
  if (M <= 0) {
    ISF_HIP_TRY(hipMemsetAsync(g, 0, (size_t)total * sizeof(float), st));
    return ISF_OK;
  }
  if (reduce != ISF_REDUCE_MAX) {
    hipLaunchKernelGGL(sc_bwd_add_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, g, gr, cmap,
                       count, total, C, reduce == ISF_REDUCE_MEAN ? 1 : 0);
  } else {
    int* from = nullptr;
    const long long mc = (long long)M * C;
    ISF_TRY(a.alloc_n(&from, (size_t)mc));
    ISF_HIP_TRY(hipMemsetAsync(g, 0, (size_t)total * sizeof(float), st));
    hipLaunchKernelGGL(sc_fill_int_kernel, dim3(ceil_div(mc, 256)), dim3(256), 0, st, from, (size_t)mc, P);
    hipLaunchKernelGGL(sc_bwd_argmax_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, feats, reduced,
                       cmap, total, C, from);
    hipLaunchKernelGGL(sc_bwd_max_scatter_kernel, dim3(ceil_div(mc, 256)), dim3(256), 0, st, g, gr, from,
                       mc, C, P);
  }
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

// ---------------------------------------------------------------------------------------- HardSimpleVFE
__global__ void hard_simple_vfe_kernel(const float* __restrict__ voxels, const int32_t* __restrict__ npts,
                                       int M, int T, int C, int F, float* __restrict__ out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)M * F) return;
  const int v = (int)(t / F), k = (int)(t % F);
  float s = 0.f;
  for (int j = 0; j < T; ++j) s += voxels[((size_t)v * T + j) * C + k];  // same order as sum(dim=1)
  out[t] = __fdiv_rn(s, (float)npts[v]);
}

}  // namespace isf

extern "C" {

int isf_dynamic_point_to_voxel_forward(const float* feats, const int32_t* coors, int num_points,
                                       int num_feats, int reduce_type, float* reduced_feats,
                                       int32_t* out_coors, int32_t* coors_map, int32_t* reduce_count,
                                       int* num_voxels_host, isf_stream_t stream) {
  ISF_REQUIRE(num_points >= 0 && num_feats > 0 && num_voxels_host && reduce_type >= 0 && reduce_type <= 2,
              ISF_ERR_ARG, "dynamic_point_to_voxel_forward: bad arguments");
  ISF_REQUIRE(num_points == 0 || (feats && coors && reduced_feats && out_coors && coors_map && reduce_count),
              ISF_ERR_ARG, "dynamic_point_to_voxel_forward: null pointer");
  isf::Arena& a = isf::arena_for_stream(isf::as_stream(stream));
  ISF_TRY(a.reset());
  return isf::dynamic_scatter_forward_impl(a, feats, coors, num_points, num_feats, reduce_type,
                                           reduced_feats, out_coors, coors_map, reduce_count,
                                           num_voxels_host, isf::as_stream(stream));
}

int isf_dynamic_point_to_voxel_backward(float* grad_feats, const float* grad_reduced_feats,
                                        const float* feats, const float* reduced_feats,
                                        const int32_t* coors_map, const int32_t* reduce_count,
                                        int num_points, int num_voxels, int num_feats, int reduce_type,
                                        isf_stream_t stream) {
  ISF_REQUIRE(num_points >= 0 && num_voxels >= 0 && num_feats > 0 && reduce_type >= 0 && reduce_type <= 2,
              ISF_ERR_ARG, "dynamic_point_to_voxel_backward: bad arguments");
  ISF_REQUIRE(num_points == 0 || (grad_feats && coors_map), ISF_ERR_ARG,
              "dynamic_point_to_voxel_backward: null pointer");
  isf::Arena& a = isf::arena_for_stream(isf::as_stream(stream));
  ISF_TRY(a.reset());
  return isf::dynamic_scatter_backward_impl(a, grad_feats, grad_reduced_feats, feats, reduced_feats,
                                            coors_map, reduce_count, num_points, num_voxels, num_feats,
                                            reduce_type, isf::as_stream(stream));
}

int isf_hard_simple_vfe(const float* voxels, const int32_t* num_points, int num_voxels, int max_points,
                        int num_point_features, int num_features, float* out, isf_stream_t stream) {
  ISF_REQUIRE(num_voxels >= 0 && max_points > 0 && num_features > 0 && num_features <= num_point_features,
              ISF_ERR_ARG, "hard_simple_vfe: bad arguments");
  if (num_voxels == 0) return ISF_OK;
  ISF_REQUIRE(voxels && num_points && out, ISF_ERR_ARG, "hard_simple_vfe: null pointer");
  hipLaunchKernelGGL(isf::hard_simple_vfe_kernel,
                     dim3(isf::ceil_div((long long)num_voxels * num_features, 256)), dim3(256), 0,
                     isf::as_stream(stream), voxels, num_points, num_voxels, max_points,
                     num_point_features, num_features, out);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

}  // extern "C"
