// isf_vfe.hip -- A4 DynamicVFE.forward fused (voxel_encoder.py:453-547).
//
// Reference data flow: 3x (unique_dim sort + atomic scatter) + 2 dense int64 canvases of B*D*H*W
// entries (663 MB/sample) to map voxels back to points.  Here:
//   mark bitmap -> popcount scan (voxel id = rank, sorted (b,z,y,x) order, no sort, no canvas)
//   pass A  per point: voxel id, exact fixed-point xyz sums + count          (cluster centre)
//   pass B  per point: 11 features -> Linear+BN+ReLU (64) -> per-voxel max   (layer 1, h1 never stored)
//   pass C  per voxel: u = W2[:,64:] . vmax1                                  (voxel half of layer 2)
//   pass D  per point: recompute h1, W2[:,:64].h1 + u[voxel] -> BN+ReLU -> per-voxel max (layer 2)
// Point features [P,64] are never written to HBM (307 MB at P=1.2M); they are recomputed (704 FMA).
// Thread-per-point kernels keep the 64 accumulators in VGPRs and stream the weights through SGPRs
// (wave-uniform s_load), i.e. the FMA pipe sees one VGPR + one SGPR operand per op; the per-voxel max
// is issued channel-per-lane (one 256-B row per wave instruction) after an LDS transpose.
#include "isf_common.h"

namespace isf {

static constexpr int kVfeThreads = 128;
static constexpr int kC = 64;            // c1 == c2 == 64 (config); other widths -> ISF_ERR_UNSUPPORTED
static constexpr int kLdsStride = kC + 1;
static constexpr double kFix = 16777216.0;  // 2^24 fixed point for the exact coordinate sums

struct VfeGeom {
  float vx, vy, vz, ox, oy, oz;  // voxel size, centre offsets (vs/2 + range_min)
};

__global__ void vfe_transpose_kernel(const float* __restrict__ w, int rows, int cols, int col0,
                                     int ncols, float* __restrict__ wt) {
  // wt[k][o] = w[o][col0 + k]   (w is [rows, cols] torch Linear layout; wt is [ncols, rows])
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * ncols) return;
  const int k = t / rows, o = t % rows;
  wt[t] = w[(size_t)o * cols + col0 + k];
}

template <int CIN>
__global__ __launch_bounds__(256) void vfe_mean_kernel(const float* __restrict__ points,
                                                       const int32_t* __restrict__ coors4, int P, int D,
                                                       int H, int W,
                                                       const unsigned long long* __restrict__ bits,
                                                       const uint32_t* __restrict__ prefix,
                                                       int32_t* __restrict__ pt2vox,
                                                       long long* __restrict__ sums /*[N][3]*/,
                                                       int32_t* __restrict__ cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int4 c = reinterpret_cast<const int4*>(coors4)[i];
  int v = -1;
  if (c.y >= 0 && c.z >= 0 && c.w >= 0)
    v = occ_lookup(bits, prefix, (((unsigned long long)c.x * D + c.y) * H + c.z) * W + c.w);
  pt2vox[i] = v;
  if (v < 0) return;
  const float* p = points + (size_t)i * CIN;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const long long q = __double2ll_rn((double)p[k] * kFix);
    atomicAdd(reinterpret_cast<unsigned long long*>(&sums[(size_t)v * 3 + k]), (unsigned long long)q);
  }
  atomicAdd(&cnt[v], 1);
}

template <int CIN>
__device__ __forceinline__ void vfe_point_features(const float* __restrict__ p, int4 c, int v,
                                                   const long long* __restrict__ sums,
                                                   const int32_t* __restrict__ cnt, VfeGeom g,
                                                   float (&f)[CIN + 6]) {
#pragma unroll
  for (int k = 0; k < CIN; ++k) f[k] = p[k];
  const double n = (double)cnt[v];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float mean = (float)((double)sums[(size_t)v * 3 + k] / (kFix * n));
    f[CIN + k] = __fsub_rn(p[k], mean);                        // xyz - cluster centre (:500-503)
  }
  // xyz - voxel centre, centre = idx*vs + (vs/2 + min)  (:505-512); no FMA contraction
  f[CIN + 3] = __fsub_rn(p[0], __fadd_rn(__fmul_rn((float)c.w, g.vx), g.ox));
  f[CIN + 4] = __fsub_rn(p[1], __fadd_rn(__fmul_rn((float)c.z, g.vy), g.oy));
  f[CIN + 5] = __fsub_rn(p[2], __fadd_rn(__fmul_rn((float)c.y, g.vz), g.oz));
}

template <int F>
__device__ __forceinline__ void vfe_layer1(const float (&f)[F], const float* __restrict__ w1t,
                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                           float (&h)[kC]) {
#pragma unroll
  for (int o = 0; o < kC; ++o) h[o] = 0.f;
#pragma unroll
  for (int k = 0; k < F; ++k) {
#pragma unroll
    for (int o = 0; o < kC; ++o) h[o] = fmaf(f[k], w1t[k * kC + o], h[o]);
  }
#pragma unroll
  for (int o = 0; o < kC; ++o) h[o] = fmaxf(fmaf(h[o], scale[o], shift[o]), 0.f);
}

// per-voxel max, channel-per-lane: values are post-ReLU (>= 0) so the int view is order preserving and
// the zero-initialised destination equals the reference's -inf start for every non-empty voxel.
__device__ __forceinline__ void vfe_wave_max_rows(const float* __restrict__ tile /*[64][kLdsStride]*/,
                                                  const int* __restrict__ vox /*[64]*/, int lane,
                                                  int* __restrict__ dst /*[N][64] as int*/) {
  for (int p = 0; p < 64; ++p) {
    const int v = vox[p];
    if (v < 0) continue;
    const float val = tile[p * kLdsStride + lane];
    if (val > 0.f) atomicMax(&dst[(size_t)v * kC + lane], __float_as_int(val));
  }
}

template <int CIN>
__global__ __launch_bounds__(kVfeThreads) void vfe_layer1_kernel(
    const float* __restrict__ points, const int32_t* __restrict__ coors4, int P,
    const int32_t* __restrict__ pt2vox, const long long* __restrict__ sums,
    const int32_t* __restrict__ cnt, VfeGeom g, const float* __restrict__ w1t,
    const float* __restrict__ scale1, const float* __restrict__ shift1, int* __restrict__ vmax1) {
  __shared__ float tile[kVfeThreads * kLdsStride];
  __shared__ int vox[kVfeThreads];
  const int t = threadIdx.x, i = blockIdx.x * kVfeThreads + t;
  const int v = i < P ? pt2vox[i] : -1;
  vox[t] = v;
  if (v >= 0) {
    float f[CIN + 6], h[kC];
    vfe_point_features<CIN>(points + (size_t)i * CIN, reinterpret_cast<const int4*>(coors4)[i], v, sums,
                            cnt, g, f);
    vfe_layer1<CIN + 6>(f, w1t, scale1, shift1, h);
#pragma unroll
    for (int o = 0; o < kC; ++o) tile[t * kLdsStride + o] = h[o];
  }
  __syncthreads();
  const int wave = t >> 6, lane = t & 63;
  vfe_wave_max_rows(tile + wave * 64 * kLdsStride, vox + wave * 64, lane, vmax1);
}

// u[v][o] = sum_k vmax1[v][k] * w2bt[k][o]
__global__ __launch_bounds__(kVfeThreads) void vfe_voxel_term_kernel(const float* __restrict__ vmax1,
                                                                      const int* __restrict__ nvox,
                                                                      const float* __restrict__ w2bt,
                                                                      float* __restrict__ u) {
  __shared__ float tile[kVfeThreads * kLdsStride];
  const int N = *nvox;
  const int t = threadIdx.x, v0 = blockIdx.x * kVfeThreads;
  if (v0 >= N) return;
  // coalesced load of this block's rows (channel-per-lane), transposed into row-per-thread via LDS
  for (int idx = t; idx < kVfeThreads * kC; idx += kVfeThreads) {
    const int r = idx / kC, k = idx % kC;
    tile[r * kLdsStride + k] = (v0 + r < N) ? vmax1[(size_t)(v0 + r) * kC + k] : 0.f;
  }
  __syncthreads();
  float acc[kC];
#pragma unroll
  for (int o = 0; o < kC; ++o) acc[o] = 0.f;
  for (int k = 0; k < kC; ++k) {
    const float gk = tile[t * kLdsStride + k];
#pragma unroll
    for (int o = 0; o < kC; ++o) acc[o] = fmaf(gk, w2bt[k * kC + o], acc[o]);
  }
  __syncthreads();
#pragma unroll
  for (int o = 0; o < kC; ++o) tile[t * kLdsStride + o] = acc[o];
  __syncthreads();
  for (int idx = t; idx < kVfeThreads * kC; idx += kVfeThreads) {
    const int r = idx / kC, k = idx % kC;
    if (v0 + r < N) u[(size_t)(v0 + r) * kC + k] = tile[r * kLdsStride + k];
  }
}

template <int CIN>
__global__ __launch_bounds__(kVfeThreads) void vfe_layer2_kernel(
    const float* __restrict__ points, const int32_t* __restrict__ coors4, int P,
    const int32_t* __restrict__ pt2vox, const long long* __restrict__ sums,
    const int32_t* __restrict__ cnt, VfeGeom g, const float* __restrict__ w1t,
    const float* __restrict__ scale1, const float* __restrict__ shift1,
    const float* __restrict__ w2at, const float* __restrict__ u, const float* __restrict__ scale2,
    const float* __restrict__ shift2, int* __restrict__ out) {
  __shared__ float tile[kVfeThreads * kLdsStride];
  __shared__ int vox[kVfeThreads];
  const int t = threadIdx.x, i = blockIdx.x * kVfeThreads + t;
  const int v = i < P ? pt2vox[i] : -1;
  vox[t] = v;
  if (v >= 0) {
    float f[CIN + 6], h[kC];
    vfe_point_features<CIN>(points + (size_t)i * CIN, reinterpret_cast<const int4*>(coors4)[i], v, sums,
                            cnt, g, f);
    vfe_layer1<CIN + 6>(f, w1t, scale1, shift1, h);
    // stage h1 in this thread's LDS row so the k loop can index it at run time
#pragma unroll
    for (int o = 0; o < kC; ++o) tile[t * kLdsStride + o] = h[o];
    float acc[kC];
#pragma unroll
    for (int o = 0; o < kC; ++o) acc[o] = 0.f;
    for (int k = 0; k < kC; ++k) {
      const float gk = tile[t * kLdsStride + k];
#pragma unroll
      for (int o = 0; o < kC; ++o) acc[o] = fmaf(gk, w2at[k * kC + o], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < kC; ++o) tile[t * kLdsStride + o] = acc[o];
  }
  __syncthreads();
  // epilogue channel-per-lane: + voxel term, BN, ReLU, per-voxel max
  const int wave = t >> 6, lane = t & 63;
  const float sc = scale2[lane], sh = shift2[lane];
  const float* wt = tile + wave * 64 * kLdsStride;
  const int* wv = vox + wave * 64;
  for (int p = 0; p < 64; ++p) {
    const int vv = wv[p];
    if (vv < 0) continue;
    float val = wt[p * kLdsStride + lane] + u[(size_t)vv * kC + lane];
    val = fmaf(val, sc, sh);
    if (val > 0.f) atomicMax(&out[(size_t)vv * kC + lane], __float_as_int(val));
  }
}

template <int CIN>
static int vfe_run(Arena& a, const float* points, const int32_t* coors4, int P, int B, const int grid[3],
                   VfeGeom g, const float* w1, const float* scale1, const float* shift1, const float* w2,
                   const float* scale2, const float* shift2, float* voxel_feats, int32_t* voxel_coors,
                   int32_t* pt2vox_out, int* n_host, OccIndex* occ_out, int d_alloc, hipStream_t st) {
  const int F = CIN + 6;
  OccIndex occ;
  // d_alloc > grid z lets the sparse encoder (sparse_shape[0] = grid z + 1) reuse this index for level 0
  ISF_TRY(occ_create(a, &occ, B, d_alloc > grid[2] ? d_alloc : grid[2], grid[1], grid[0], st));
  ISF_TRY(occ_mark_coords4(occ, coors4, P, st));
  ISF_TRY(occ_scan(a, occ, st));
  float *w1t, *w2at, *w2bt;
  ISF_TRY(a.alloc_n(&w1t, (size_t)F * kC));
  ISF_TRY(a.alloc_n(&w2at, (size_t)kC * kC));
  ISF_TRY(a.alloc_n(&w2bt, (size_t)kC * kC));
  hipLaunchKernelGGL(vfe_transpose_kernel, dim3(ceil_div(F * kC, 256)), dim3(256), 0, st, w1, kC, F, 0, F, w1t);
  hipLaunchKernelGGL(vfe_transpose_kernel, dim3(ceil_div(kC * kC, 256)), dim3(256), 0, st, w2, kC, 2 * kC, 0, kC, w2at);
  hipLaunchKernelGGL(vfe_transpose_kernel, dim3(ceil_div(kC * kC, 256)), dim3(256), 0, st, w2, kC, 2 * kC, kC, kC, w2bt);
  int N = 0;
  ISF_TRY(read_int(occ.total, &N, st));  // the one host sync of the VFE: sizes every per-voxel buffer
  *n_host = N;
  if (occ_out) *occ_out = occ;
  if (N == 0) {
    if (pt2vox_out) ISF_HIP_TRY(hipMemsetAsync(pt2vox_out, 0xff, (size_t)P * sizeof(int32_t), st));
    return ISF_OK;
  }
  int32_t* pt2vox = pt2vox_out;
  if (!pt2vox) ISF_TRY(a.alloc_n(&pt2vox, (size_t)P));
  long long* sums;
  int32_t* cnt;
  float *vmax1, *u;
  ISF_TRY(a.alloc_n(&sums, (size_t)N * 3));
  ISF_TRY(a.alloc_n(&cnt, (size_t)N));
  ISF_TRY(a.alloc_n(&vmax1, (size_t)N * kC));
  ISF_TRY(a.alloc_n(&u, (size_t)N * kC));
  ISF_HIP_TRY(hipMemsetAsync(sums, 0, (size_t)N * 3 * sizeof(long long), st));
  ISF_HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)N * sizeof(int32_t), st));
  ISF_HIP_TRY(hipMemsetAsync(vmax1, 0, (size_t)N * kC * sizeof(float), st));
  ISF_HIP_TRY(hipMemsetAsync(voxel_feats, 0, (size_t)N * kC * sizeof(float), st));
  ISF_TRY(occ_compact_coords4(occ, voxel_coors, st));
  hipLaunchKernelGGL(vfe_mean_kernel<CIN>, dim3(ceil_div(P, 256)), dim3(256), 0, st, points, coors4, P,
                     occ.D, occ.H, occ.W, occ.bits, occ.prefix, pt2vox, sums, cnt);
  const int pblocks = ceil_div(P, kVfeThreads);
  hipLaunchKernelGGL(vfe_layer1_kernel<CIN>, dim3(pblocks), dim3(kVfeThreads), 0, st, points, coors4, P,
                     pt2vox, sums, cnt, g, w1t, scale1, shift1, reinterpret_cast<int*>(vmax1));
  hipLaunchKernelGGL(vfe_voxel_term_kernel, dim3(ceil_div(N, kVfeThreads)), dim3(kVfeThreads), 0, st,
                     vmax1, occ.total, w2bt, u);
  hipLaunchKernelGGL(vfe_layer2_kernel<CIN>, dim3(pblocks), dim3(kVfeThreads), 0, st, points, coors4, P,
                     pt2vox, sums, cnt, g, w1t, scale1, shift1, w2at, u, scale2, shift2,
                     reinterpret_cast<int*>(voxel_feats));
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int dynamic_vfe_impl(Arena& a, const float* points, const int32_t* coors4, int P, int Cin, int B,
                     const float vs[3], const float range[6], const float* w1, const float* scale1,
                     const float* shift1, int c1, const float* w2, const float* scale2,
                     const float* shift2, int c2, float* voxel_feats, int32_t* voxel_coors,
                     int32_t* pt2vox, int* num_voxels_host, OccIndex* occ_out, int grid_d_alloc,
                     hipStream_t st) {
  ISF_REQUIRE(c1 == kC && c2 == kC, ISF_ERR_UNSUPPORTED,
              "dynamic_vfe: feat_channels (%d,%d) not built; this build has (64,64)", c1, c2);
  ISF_REQUIRE(Cin == 4 || Cin == 5, ISF_ERR_UNSUPPORTED, "dynamic_vfe: in_channels %d not built (4|5)", Cin);
  *num_voxels_host = 0;
  if (P <= 0) return ISF_OK;
  int grid[3];
  for (int j = 0; j < 3; ++j) grid[j] = (int)roundf((range[3 + j] - range[j]) / vs[j]);
  VfeGeom g;
  g.vx = vs[0]; g.vy = vs[1]; g.vz = vs[2];
  g.ox = vs[0] / 2 + range[0]; g.oy = vs[1] / 2 + range[1]; g.oz = vs[2] / 2 + range[2];
  if (Cin == 5)
    return vfe_run<5>(a, points, coors4, P, B, grid, g, w1, scale1, shift1, w2, scale2, shift2,
                      voxel_feats, voxel_coors, pt2vox, num_voxels_host, occ_out, grid_d_alloc, st);
  return vfe_run<4>(a, points, coors4, P, B, grid, g, w1, scale1, shift1, w2, scale2, shift2, voxel_feats,
                    voxel_coors, pt2vox, num_voxels_host, occ_out, grid_d_alloc, st);
}

}  // namespace isf

extern "C" {

int isf_dynamic_vfe_forward(const float* points, const int32_t* coors4, int num_points, int in_channels,
                            int batch_size, const float voxel_size_host[3],
                            const float coors_range_host[6], const float* w1, const float* scale1,
                            const float* shift1, int c1, const float* w2, const float* scale2,
                            const float* shift2, int c2, float* voxel_feats, int32_t* voxel_coors,
                            int32_t* pt2vox, int* num_voxels_host, isf_stream_t stream) {
  ISF_REQUIRE(num_points >= 0 && batch_size > 0 && num_voxels_host && voxel_size_host && coors_range_host,
              ISF_ERR_ARG, "dynamic_vfe_forward: bad arguments");
  ISF_REQUIRE(num_points == 0 || (points && coors4 && w1 && scale1 && shift1 && w2 && scale2 && shift2 &&
                                  voxel_feats && voxel_coors),
              ISF_ERR_ARG, "dynamic_vfe_forward: null pointer");
  isf::Arena& a = isf::arena_for_current_device();
  ISF_TRY(a.reset());
  return isf::dynamic_vfe_impl(a, points, coors4, num_points, in_channels, batch_size, voxel_size_host,
                               coors_range_host, w1, scale1, shift1, c1, w2, scale2, shift2, c2,
                               voxel_feats, voxel_coors, pt2vox, num_voxels_host, nullptr, 0,
                               isf::as_stream(stream));
}

}  // extern "C"
