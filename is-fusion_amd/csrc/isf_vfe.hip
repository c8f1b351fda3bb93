// isf_vfe.hip -- A4 DynamicVFE.forward fused (voxel_encoder.py:453-547).
//
// Reference data flow: 3x (unique_dim sort + atomic scatter) + 2 dense int64 canvases of B*D*H*W entries
// (663 MB/sample) to map voxels back to points.  Here:
//   mark bitmap -> popcount scan            voxel id = rank, sorted (b,z,y,x) order, no sort, no canvas
//   count / scan / order                    points grouped by voxel (counting sort: 1 int atomic per point)
//   mean        thread per voxel            exact 2^-24 fixed-point int64 sums -> order independent
//   layer 1     128 sorted points / block   11 features -> Linear+BN+ReLU (fp32 VALU, weights through SGPRs)
//                                           -> per-voxel max
//   layer 2     64 sorted points / wave     h1 recomputed (never stored: 307 MB at P=1.2M), [h1 | vmax1[voxel]]
//                                           (128) x W2^T on the f16 matrix cores with the hi/lo split
//                                           arithmetic of isf_spconv16.hip (fp32-class accuracy), BN+ReLU,
//                                           per-voxel max
// Because a voxel's points are contiguous after the counting sort, the per-voxel max is a segmented
// reduction inside the wave (channel per lane, 256-byte row stores); only segments cut by a wave boundary
// fall back to atomics (order independent: max of non-negative floats in their integer view).  Every
// reduction is order independent => the VFE is bit-reproducible run to run.
#include "isf_common.h"

namespace isf {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

static constexpr int kC = 64;  // c1 == c2 == 64 (config); other widths -> ISF_ERR_UNSUPPORTED
static constexpr int kL1Threads = 128;
static constexpr int kLdsStride = kC + 1;
static constexpr double kFix = 16777216.0;  // 2^24 fixed point for the exact coordinate sums

struct VfeGeom {
  float vx, vy, vz, ox, oy, oz;  // voxel size, centre offsets (vs/2 + range_min)
};

__device__ __forceinline__ void vfe_split8(const f32x8 v, uint4& hi, uint4& lo) {
  const h8 h = __builtin_convertvector(v, h8);
  const f32x8 r = v - __builtin_convertvector(h, f32x8);
  const h8 l = __builtin_convertvector(r, h8);
  hi = *reinterpret_cast<const uint4*>(&h);
  lo = *reinterpret_cast<const uint4*>(&l);
}

// ------------------------------------------------------------------------------------------ weight prep
// w2p[kc][nt][hi|lo][lane][8] = split(w2[16nt + (lane&15)][32kc + 8(lane>>4) + jj] * 2^sw), kc = 0..3
// sc2[o] = scale2[o] * 2^-sw
// w1p[nt][hi|lo][lane][4] = split(w1[16nt + (lane&15)][4(lane>>4) + jj] * 2^sw1) (zero for k >= F): B fragments of
// v_mfma_f32_16x16x16_f16;  sc1[o] = scale1[o] * 2^-sw1
__global__ void vfe_prep_kernel(const float* __restrict__ w1, int F, const float* __restrict__ w2,
                                const float* __restrict__ scale1, const float* __restrict__ scale2,
                                uint2* __restrict__ w1p, float* __restrict__ sc1,
                                uint4* __restrict__ w2p, float* __restrict__ sc2) {
  __shared__ float amax_s;
  __shared__ float amax1_s;
  const int t = threadIdx.x;  // 256 threads, one block
  {
    float m1 = 0.f;
    for (int i = t; i < kC * F; i += 256) m1 = fmaxf(m1, fabsf(w1[i]));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m1 = fmaxf(m1, __shfl_xor(m1, d, 64));
    if (t == 0) amax1_s = 0.f;
    __syncthreads();
    if ((t & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(&amax1_s), __float_as_uint(m1));
    __syncthreads();
    int e1 = 0;
    if (amax1_s > 0.f) (void)frexpf(amax1_s, &e1);
    const int sw1 = amax1_s > 0.f ? 13 - e1 : 0;
    const float s1 = ldexpf(1.f, sw1), inv1 = ldexpf(1.f, -sw1);
    if (t < kC) sc1[t] = scale1[t] * inv1;
    for (int i = t; i < 4 * 64; i += 256) {  // (nt, lane)
      const int lane = i & 63, nt = i >> 6;
      _Float16 hi[4], lo[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int k = 4 * (lane >> 4) + jj;
        const float x = k < F ? w1[(size_t)(16 * nt + (lane & 15)) * F + k] * s1 : 0.f;
        hi[jj] = (_Float16)x;
        lo[jj] = (_Float16)(x - (float)hi[jj]);
      }
      w1p[(nt * 2 + 0) * 64 + lane] = *reinterpret_cast<const uint2*>(hi);
      w1p[(nt * 2 + 1) * 64 + lane] = *reinterpret_cast<const uint2*>(lo);
    }
  }
  float m = 0.f;
  for (int i = t; i < kC * 2 * kC; i += 256) m = fmaxf(m, fabsf(w2[i]));
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
  if (t == 0) amax_s = 0.f;
  __syncthreads();
  if ((t & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(&amax_s), __float_as_uint(m));
  __syncthreads();
  const float amax = amax_s;
  int e = 0;
  if (amax > 0.f) (void)frexpf(amax, &e);
  const int sw = amax > 0.f ? 13 - e : 0;
  const float s = ldexpf(1.f, sw), inv = ldexpf(1.f, -sw);
  if (t < kC) sc2[t] = scale2[t] * inv;
  for (int i = t; i < 4 * 4 * 64; i += 256) {  // (kc, nt, lane)
    const int lane = i & 63, nt = (i >> 6) & 3, kc = i >> 8;
    f32x8 v;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
      v[jj] = w2[(size_t)(16 * nt + (lane & 15)) * (2 * kC) + 32 * kc + 8 * (lane >> 4) + jj] * s;
    uint4 hi, lo;
    vfe_split8(v, hi, lo);
    w2p[(size_t)(kc * 4 + nt) * 128 + lane] = hi;
    w2p[(size_t)(kc * 4 + nt) * 128 + 64 + lane] = lo;
  }
}

// ------------------------------------------------------------------------------------------ grouping
__global__ __launch_bounds__(256) void vfe_count_kernel(const int32_t* __restrict__ coors4, int P, int D,
                                                        int H, int W,
                                                        const unsigned long long* __restrict__ bits,
                                                        const uint32_t* __restrict__ prefix,
                                                        int32_t* __restrict__ pt2vox,
                                                        int32_t* __restrict__ slot,
                                                        uint32_t* __restrict__ cnt,
                                                        int32_t* __restrict__ voxel_coors) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int4 c = reinterpret_cast<const int4*>(coors4)[i];
  int v = -1;
  if (c.y >= 0 && c.z >= 0 && c.w >= 0)
    v = occ_lookup(bits, prefix, (((unsigned long long)c.x * D + c.y) * H + c.z) * W + c.w);
  pt2vox[i] = v;
  if (v >= 0) {
    const uint32_t s = atomicAdd(&cnt[v], 1u);
    slot[i] = (int32_t)s;
    if (s == 0) reinterpret_cast<int4*>(voxel_coors)[v] = c;  // exactly one point per voxel draws slot 0
  }
}

// Points are copied ONCE into voxel-sorted 32-byte records {CIN features, voxel id}: the three per-point passes
// (mean, layer 1, layer 2) then stream contiguous records instead of gathering points / coords / voxel ids through
// an index (measured FETCH_SIZE of the gathering version: 146 + 328 + 437 MB for 24 MB of points).
static constexpr int kRec = 8;   // floats per record
template <int CIN>
__global__ __launch_bounds__(256) void vfe_order_kernel(const float* __restrict__ points,
                                                        const int32_t* __restrict__ pt2vox,
                                                        const int32_t* __restrict__ slot, int P,
                                                        const uint32_t* __restrict__ start,
                                                        float* __restrict__ recs) {
  static_assert(CIN + 1 <= kRec, "record too small");
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int v = pt2vox[i];
  if (v < 0) return;
  float r[kRec];
#pragma unroll
  for (int k = 0; k < kRec; ++k) r[k] = 0.f;
#pragma unroll
  for (int k = 0; k < CIN; ++k) r[k] = points[(size_t)i * CIN + k];
  r[kRec - 1] = __int_as_float(v);
  float4* o = reinterpret_cast<float4*>(recs + (size_t)(start[v] + slot[i]) * kRec);
  o[0] = make_float4(r[0], r[1], r[2], r[3]);
  o[1] = make_float4(r[4], r[5], r[6], r[7]);
}

template <int CIN>
__global__ __launch_bounds__(256) void vfe_mean_kernel(const float* __restrict__ recs,
                                                       const uint32_t* __restrict__ start, int N,
                                                       float4* __restrict__ mean4) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= N) return;
  const uint32_t j0 = start[v], j1 = start[v + 1];
  long long sx = 0, sy = 0, sz = 0;
  for (uint32_t j = j0; j < j1; ++j) {
    const float* p = recs + (size_t)j * kRec;
    sx += __double2ll_rn((double)p[0] * kFix);
    sy += __double2ll_rn((double)p[1] * kFix);
    sz += __double2ll_rn((double)p[2] * kFix);
  }
  const double d = kFix * (double)(j1 - j0);
  mean4[v] = make_float4((float)((double)sx / d), (float)((double)sy / d), (float)((double)sz / d),
                         (float)(j1 - j0));
}

// ------------------------------------------------------------------------------------------ per-point math
template <int CIN>
__device__ __forceinline__ void vfe_point_features(const float* __restrict__ p, int4 c, float4 mean, VfeGeom g,
                                                   float (&f)[CIN + 6]) {
#pragma unroll
  for (int k = 0; k < CIN; ++k) f[k] = p[k];
  f[CIN + 0] = __fsub_rn(p[0], mean.x);  // xyz - cluster centre (:500-503)
  f[CIN + 1] = __fsub_rn(p[1], mean.y);
  f[CIN + 2] = __fsub_rn(p[2], mean.z);
  // xyz - voxel centre, centre = idx*vs + (vs/2 + min)  (:505-512); no FMA contraction
  f[CIN + 3] = __fsub_rn(p[0], __fadd_rn(__fmul_rn((float)c.w, g.vx), g.ox));
  f[CIN + 4] = __fsub_rn(p[1], __fadd_rn(__fmul_rn((float)c.z, g.vy), g.oy));
  f[CIN + 5] = __fsub_rn(p[2], __fadd_rn(__fmul_rn((float)c.y, g.vz), g.oz));
}

// Layer 1 on the matrix cores: the wave's 64 points x F (<= 16) features times W1^T [16 x 64] as 4 row groups x
// 4 column tiles of v_mfma_f32_16x16x16_f16 in the same hi/lo split arithmetic as everywhere else (fp32-class).
// The per-lane features go through a 4 KiB wave-private LDS staging area ([point][16 hi | 16 lo] halves) to reach
// the A-fragment layout; the result stays in the C/D layout: acc[rg][nt][t] = sum for point 16 rg + 4 (lane>>4) + t,
// channel 16 nt + (lane & 15), before BatchNorm.  (The VALU version cost 704 FMAs per point fed by 44 scalar
// weight loads per wave.)
typedef _Float16 h4v __attribute__((ext_vector_type(4)));

template <int F>
__device__ __forceinline__ void vfe_layer1_mfma(const float (&f)[F], bool valid, const uint2* __restrict__ w1p,
                                                char* __restrict__ stage, int lane, f32x4 (&acc)[4][4]) {
  static_assert(F <= 16, "layer-1 features must fit one K = 16 MFMA");
  {
    _Float16 hi[16], lo[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float x = (k < F && valid) ? f[k < F ? k : 0] : 0.f;
      hi[k] = (_Float16)x;
      lo[k] = (_Float16)(x - (float)hi[k]);
    }
    uint4* sp = reinterpret_cast<uint4*>(stage + lane * 64);
    sp[0] = *reinterpret_cast<const uint4*>(&hi[0]);
    sp[1] = *reinterpret_cast<const uint4*>(&hi[8]);
    sp[2] = *reinterpret_cast<const uint4*>(&lo[0]);
    sp[3] = *reinterpret_cast<const uint4*>(&lo[8]);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  const int col = lane & 15, kq = lane >> 4;
  uint2 ah[4], al[4];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const char* base = stage + (16 * rg + col) * 64 + kq * 8;
    ah[rg] = *reinterpret_cast<const uint2*>(base);
    al[rg] = *reinterpret_cast<const uint2*>(base + 32);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const uint2 bhu = w1p[(nt * 2 + 0) * 64 + lane];
    const uint2 blu = w1p[(nt * 2 + 1) * 64 + lane];
    const h4v bh = *reinterpret_cast<const h4v*>(&bhu);
    const h4v bl = *reinterpret_cast<const h4v*>(&blu);
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const h4v a_h = *reinterpret_cast<const h4v*>(&ah[rg]);
      const h4v a_l = *reinterpret_cast<const h4v*>(&al[rg]);
      f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
      c = __builtin_amdgcn_mfma_f32_16x16x16f16(a_l, bh, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x16f16(a_h, bl, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x16f16(a_h, bh, c, 0, 0, 0);
      acc[rg][nt] = c;
    }
  }
}

// C/D-layout layer-1 sums -> relu(BN) -> fp32 tile [point - 32 half][kLdsStride] for the points of row groups 2 half
// and 2 half + 1.  The tiles hold HALF a wave's points (8.3 KiB per wave instead of 16.6): LDS is what bounds the
// occupancy of the two VFE kernels (round 2: 8 waves per CU; now 18).
static constexpr int kHalfPts = 32;
__device__ __forceinline__ void vfe_layer1_store_tile(const f32x4 (&acc)[4][4], const float* __restrict__ sc1,
                                                      const float* __restrict__ shift1, float* __restrict__ tile,
                                                      int lane, int half) {
  const int col = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const float sc = sc1[16 * nt + col], sh = shift1[16 * nt + col];
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        tile[(16 * r2 + 4 * kq + t) * kLdsStride + 16 * nt + col] =
            fmaxf(fmaf(half ? acc[2 + r2][nt][t] : acc[r2][nt][t], sc, sh), 0.f);
  }
}

// Segmented per-voxel max over the 64 voxel-sorted points of one wave; lane = channel.  `val(p)` yields this lane's
// channel of point p (>= 0: post-ReLU).  The wave's 64 voxel ids sit in one VGPR (`myvox`, lane p = point p) and are
// broadcast with v_readlane (SGPR, no LDS round trip per point); the values are read from LDS 16 points at a time
// (independent ds_reads in flight) before the serial run logic touches them.  A run that does not reach a wave
// boundary -- or whose neighbour across the boundary belongs to another voxel (`vprev`, `vnext`: the voxel ids of
// records j0-1 and j0+64) -- is written with one 256-byte row store; a run cut by the boundary uses integer
// atomicMax on the zero-initialised destination (identical result for non-negative floats).
// fill(half) puts the points [32 half, 32 half + 32) into the wave's tile before val() reads them (val(p): p in 0..63).
// split_dst != nullptr: whole rows go there in the SPLIT activation format of the sparse-conv kernels (isf_common.h:
// per 32 channels 4 x 8 f16 hi then 4 x 8 f16 lo) instead of fp32 into dst -- lane = channel stores its two halves;
// rows cut by a wave boundary still accumulate in dst (fp32 atomicMax) and are converted by vfe_cut_rows_split_kernel.
__device__ __forceinline__ void vfe_store_split(_Float16* __restrict__ split_dst, int row, int lane, float m) {
  const _Float16 hi = (_Float16)m;
  const _Float16 lo = (_Float16)(m - (float)hi);
  _Float16* p = split_dst + (size_t)row * (2 * kC) + (lane >> 5) * 64 + (lane & 31);
  p[0] = hi;
  p[32] = lo;
}

template <typename FillFn, typename ValFn>
__device__ __forceinline__ void vfe_segmented_max(int myvox, int vprev, int vnext, int lane, float* __restrict__ dst,
                                                  FillFn fill, ValFn val, _Float16* __restrict__ split_dst = nullptr) {
  int run = -1, first = 0;
  float m = 0.f;
  auto flush = [&](int last) {   // run covers points [first, last]
    if (run < 0) return;
    const bool whole = (first > 0 || vprev != run) && (last < 63 || vnext != run);
    if (whole) {
      if (split_dst) vfe_store_split(split_dst, run, lane, m);
      else dst[(size_t)run * kC + lane] = m;
    } else if (m > 0.f) {
      atomicMax(reinterpret_cast<int*>(dst) + (size_t)run * kC + lane, __float_as_int(m));
    }
  };
#pragma unroll
  for (int c = 0; c < 64; c += 16) {
    if (c % kHalfPts == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the previous half has been read
      fill(c / kHalfPts);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    float vals[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) vals[q] = val(c + q);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int v = __builtin_amdgcn_readlane(myvox, c + q);
      if (v < 0) continue;   // past the last valid point (only at the very end of the sorted list)
      if (v != run) {
        flush(c + q - 1);
        run = v;
        first = c + q;
        m = 0.f;
      }
      m = fmaxf(m, vals[q]);
    }
  }
  flush(63);
}

// The segmented max writes a voxel's row with ONE store when the voxel's run of records lies inside a wave, and with
// atomicMax when a wave boundary (every 64 records) cuts it -- which needs a zero row to start from.  Only those rows are
// zeroed: one wave per boundary clears the rows of the voxel that straddles it, in both destination buffers (at most
// P / 64 rows; round 2 zero-filled both [N, 64] buffers: two 92-MB fills per forward).
__global__ __launch_bounds__(256) void vfe_zero_cut_rows_kernel(const float* __restrict__ recs,
                                                                const int* __restrict__ n_valid,
                                                                float* __restrict__ a, float* __restrict__ b) {
  const int lane = threadIdx.x & 63;
  const long long k = (long long)blockIdx.x * 4 + (threadIdx.x >> 6) + 1;   // boundary between records 64k - 1 and 64k
  const long long j = k * 64;
  const long long nv = *n_valid;
  if (j >= nv) return;
  const int v0 = __float_as_int(recs[(size_t)(j - 1) * 8 + 7]), v1 = __float_as_int(recs[(size_t)j * 8 + 7]);
  if (v0 != v1 || v0 < 0) return;
  a[(size_t)v0 * 64 + lane] = 0.f;
  b[(size_t)v0 * 64 + lane] = 0.f;
}

// split output: the rows of the voxels cut by a wave boundary are final in the fp32 buffer only when layer 2 has
// finished; one wave per boundary converts the row of the voxel that straddles it (a voxel cut by several boundaries is
// converted several times, to the same bits)
__global__ __launch_bounds__(256) void vfe_cut_rows_split_kernel(const float* __restrict__ recs,
                                                                 const int* __restrict__ n_valid,
                                                                 const float* __restrict__ rows,
                                                                 _Float16* __restrict__ split_dst) {
  const int lane = threadIdx.x & 63;
  const long long k = (long long)blockIdx.x * 4 + (threadIdx.x >> 6) + 1;
  const long long j = k * 64;
  const long long nv = *n_valid;
  if (j >= nv) return;
  const int v0 = __float_as_int(recs[(size_t)(j - 1) * 8 + 7]), v1 = __float_as_int(recs[(size_t)j * 8 + 7]);
  if (v0 != v1 || v0 < 0) return;
  vfe_store_split(split_dst, v0, lane, rows[(size_t)v0 * 64 + lane]);
}

// voxel id stored in record j (wave-uniform scalar load), -1 outside [0, n)
__device__ __forceinline__ int vfe_record_voxel(const float* __restrict__ recs, long long j, uint32_t n) {
  return (j >= 0 && j < (long long)n) ? __float_as_int(recs[(size_t)j * kRec + kRec - 1]) : -1;
}

template <int CIN>
__global__ __launch_bounds__(kL1Threads) void vfe_layer1_kernel(
    const float* __restrict__ recs, const int32_t* __restrict__ voxel_coors, const int* __restrict__ n_valid,
    const float4* __restrict__ mean4, VfeGeom g, const uint2* __restrict__ w1p, const float* __restrict__ sc1,
    const float* __restrict__ shift1, float* __restrict__ vmax1) {
  __shared__ __attribute__((aligned(16))) float tile[(kL1Threads / 64) * kHalfPts * kLdsStride];
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const long long wj0 = (long long)blockIdx.x * kL1Threads + wave * 64;
  const uint32_t nv = (uint32_t)*n_valid;
  if (wj0 >= (long long)nv) return;   // wave-uniform; the tile is wave-private, no block barrier below
  const uint32_t j = (uint32_t)wj0 + lane;
  int v = -1;
  float f[CIN + 6];
#pragma unroll
  for (int k = 0; k < CIN + 6; ++k) f[k] = 0.f;
  if (j < nv) {
    float rec[kRec];
    *reinterpret_cast<float4*>(rec) = reinterpret_cast<const float4*>(recs + (size_t)j * kRec)[0];
    *reinterpret_cast<float4*>(rec + 4) = reinterpret_cast<const float4*>(recs + (size_t)j * kRec)[1];
    v = __float_as_int(rec[kRec - 1]);
    vfe_point_features<CIN>(rec, reinterpret_cast<const int4*>(voxel_coors)[v], mean4[v], g, f);
  }
  float* wt = tile + wave * kHalfPts * kLdsStride;   // >= the 4 KiB layer-1 staging area
  f32x4 acc[4][4];
  vfe_layer1_mfma<CIN + 6>(f, v >= 0, w1p, reinterpret_cast<char*>(wt), lane, acc);
  vfe_segmented_max(v, vfe_record_voxel(recs, wj0 - 1, nv), vfe_record_voxel(recs, wj0 + 64, nv), lane, vmax1,
                    [&](int half) { vfe_layer1_store_tile(acc, sc1, shift1, wt, lane, half); },
                    [&](int p) { return wt[(p % kHalfPts) * kLdsStride + lane]; });
}

// ------------------------------------------------------------------------------------------ layer 2 (MFMA)
// One wave = 64 sorted points = 4 MFMA row groups; K = 128 = [h1 (64) | vmax1[voxel] (64)] in two halves.
// LDS per wave, 8.3 KiB: the A tile of ONE 32-channel chunk, split format [unit 4][hi|lo][point 64][8 halves] = 8 KiB
// (conflict free for both the point-per-lane writes and the ds_read_b128 fragment reads; the four chunks of K = 128 go
// through it one after the other, each with its B fragments loaded once), reused as the fp32 [32 points][65] tiles of
// the layer-1 hand-over and of the output (two point halves each).
static constexpr int kL2Waves = 2;
static constexpr int kL2WaveBytes = kHalfPts * kLdsStride * 4;  // 8320 >= 8192

template <int CIN>
__global__ __launch_bounds__(64 * kL2Waves) void vfe_layer2_kernel(
    const float* __restrict__ recs, const int32_t* __restrict__ voxel_coors, const int* __restrict__ n_valid,
    const float4* __restrict__ mean4, VfeGeom g, const uint2* __restrict__ w1p, const float* __restrict__ sc1,
    const float* __restrict__ shift1, const float* __restrict__ vmax1, const uint4* __restrict__ w2p,
    const float* __restrict__ sc2, const float* __restrict__ shift2, float* __restrict__ out,
    _Float16* __restrict__ out_split) {
  __shared__ __attribute__((aligned(16))) char smem[kL2Waves * kL2WaveBytes];
  __shared__ int vox_s[kL2Waves * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t j0 = (blockIdx.x * kL2Waves + wave) * 64;
  const uint32_t nv = (uint32_t)*n_valid;
  if (j0 >= nv) return;  // wave-uniform; no block-level barrier below
  uint4* atile = reinterpret_cast<uint4*>(smem + wave * kL2WaveBytes);  // [unit][hi|lo][pt] uint4
  float* ftile = reinterpret_cast<float*>(smem + wave * kL2WaveBytes);  // [pt][65]
  int* vox = vox_s + wave * 64;
  const int col = lane & 15, kg = lane >> 4;

  const uint32_t j = j0 + lane;
  int v = -1;
  float h[kC];
  {
    float f[CIN + 6];
#pragma unroll
    for (int k = 0; k < CIN + 6; ++k) f[k] = 0.f;
    if (j < nv) {
      float rec[kRec];
      *reinterpret_cast<float4*>(rec) = reinterpret_cast<const float4*>(recs + (size_t)j * kRec)[0];
      *reinterpret_cast<float4*>(rec + 4) = reinterpret_cast<const float4*>(recs + (size_t)j * kRec)[1];
      v = __float_as_int(rec[kRec - 1]);
      vfe_point_features<CIN>(rec, reinterpret_cast<const int4*>(voxel_coors)[v], mean4[v], g, f);
    }
    // layer 1 recomputed on the matrix cores (its output [P, 64] never goes to HBM), through the fp32 tile (32 points
    // at a time) back to one-point-per-lane registers for the split-format A tile of layer 2
    f32x4 acc1[4][4];
    vfe_layer1_mfma<CIN + 6>(f, v >= 0, w1p, reinterpret_cast<char*>(ftile), lane, acc1);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      vfe_layer1_store_tile(acc1, sc1, shift1, ftile, lane, half);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      if ((lane >> 5) == half) {
#pragma unroll
        for (int o = 0; o < kC; ++o) h[o] = ftile[(lane & 31) * kLdsStride + o];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
  }
  vox[lane] = v;

  f32x4 acc[4][4];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[rg][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // chunk q of K = 128 (q = 2 * half + kc: 32 channels of x, this lane's point) -> A tile -> 48 MFMAs
  auto chunk = [&](const float (&x)[kC], int half, int kc) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      f32x8 vv;
#pragma unroll
      for (int q = 0; q < 8; ++q) vv[q] = x[(4 * kc + u) * 8 + q];
      uint4 hi, lo;
      vfe_split8(vv, hi, lo);
      atile[(u * 2 + 0) * 64 + lane] = hi;
      atile[(u * 2 + 1) * 64 + lane] = lo;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    uint4 ah[4], al[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      ah[rg] = atile[(kg * 2 + 0) * 64 + rg * 16 + col];
      al[rg] = atile[(kg * 2 + 1) * 64 + rg * 16 + col];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the tile may be overwritten by the next chunk
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const uint4 bhu = w2p[(size_t)((half * 2 + kc) * 4 + nt) * 128 + lane];
      const uint4 blu = w2p[(size_t)((half * 2 + kc) * 4 + nt) * 128 + 64 + lane];
      const h8 bh = *reinterpret_cast<const h8*>(&bhu);
      const h8 bl = *reinterpret_cast<const h8*>(&blu);
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const h8 a_h = *reinterpret_cast<const h8*>(&ah[rg]);
        const h8 a_l = *reinterpret_cast<const h8*>(&al[rg]);
        acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_l, bh, acc[rg][nt], 0, 0, 0);
        acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h, bl, acc[rg][nt], 0, 0, 0);
        acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h, bh, acc[rg][nt], 0, 0, 0);
      }
    }
  };

  // half 0: h1
  chunk(h, 0, 0);
  chunk(h, 0, 1);
  // half 1: the voxel's layer-1 max (map_voxel_center_to_point gather, voxel_encoder.py:541-544)
  {
    float r[kC];
    if (v >= 0) {
      const float4* src = reinterpret_cast<const float4*>(vmax1 + (size_t)v * kC);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float4 t4 = src[q];
        r[4 * q + 0] = t4.x; r[4 * q + 1] = t4.y; r[4 * q + 2] = t4.z; r[4 * q + 3] = t4.w;
      }
    } else {
#pragma unroll
      for (int o = 0; o < kC; ++o) r[o] = 0.f;
    }
    chunk(r, 1, 0);
    chunk(r, 1, 1);
  }
  // accumulators (col = lane&15 -> channel, row = 4*(lane>>4)+t -> point) -> fp32 tile [pt - 32 half][65], half by half
  const float sc = sc2[lane], sh = shift2[lane];
  vfe_segmented_max(vox[lane], vfe_record_voxel(recs, (long long)j0 - 1, nv),
                    vfe_record_voxel(recs, (long long)j0 + 64, nv), lane, out,
                    [&](int half) {
#pragma unroll
                      for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                          for (int t = 0; t < 4; ++t)
                            ftile[(r2 * 16 + 4 * kg + t) * kLdsStride + nt * 16 + col] =
                                half ? acc[2 + r2][nt][t] : acc[r2][nt][t];
                    },
                    [&](int p) { return fmaxf(fmaf(ftile[(p % kHalfPts) * kLdsStride + lane], sc, sh), 0.f); }, out_split);
}

// ------------------------------------------------------------------------------------------ driver
template <int CIN>
static int vfe_run(Arena& a, const float* points, const int32_t* coors4, int P, int B, const int grid[3],
                   VfeGeom g, const float* w1, const float* scale1, const float* shift1, const float* w2,
                   const float* scale2, const float* shift2, float* voxel_feats, int32_t* voxel_coors,
                   int32_t* pt2vox_out, int* n_host, OccIndex* occ_out, int d_alloc, hipStream_t st,
                   hipEvent_t* coords_ready, void* voxel_feats_split, const VoxBatch* voxelize, const VoxGeom* vgeom) {
  const int F = CIN + 6;
  OccIndex occ;
  // d_alloc > grid z lets the sparse encoder (sparse_shape[0] = grid z + 1) reuse this index for level 0
  ISF_TRY(occ_create(a, &occ, B, d_alloc > grid[2] ? d_alloc : grid[2], grid[1], grid[0], st, false));
  if (voxelize)   // the frames are voxelized inside the marking launch; coors4 is written there
    ISF_TRY(occ_voxelize_mark_bytemap(a, occ, points, P, CIN, *vgeom, *voxelize, const_cast<int32_t*>(coors4), st));
  else
    ISF_TRY(occ_mark_coords4_bytemap(a, occ, coors4, P, st));
  ISF_TRY(occ_scan(a, occ, st));
  float *sc1, *sc2;
  uint2* w1p;
  uint4* w2p;
  ISF_TRY(a.alloc_n(&sc1, (size_t)kC));
  ISF_TRY(a.alloc_n(&sc2, (size_t)kC));
  ISF_TRY(a.alloc_n(&w1p, (size_t)4 * 2 * 64));
  ISF_TRY(a.alloc_n(&w2p, (size_t)4 * 4 * 128));
  hipLaunchKernelGGL(vfe_prep_kernel, dim3(1), dim3(256), 0, st, w1, F, w2, scale1, scale2, w1p, sc1, w2p, sc2);
  // The one host round trip of the VFE: N sizes every per-voxel buffer.  The read-back is followed by an event, and
  // everything that does not need N -- the point -> voxel map and the per-voxel point counts (75 us of GPU work at
  // 1.2 M points, counters allocated for the worst case of one voxel per point) -- is queued behind it BEFORE the host
  // waits, so the GPU keeps working while the host wakes up and launches the rest (the plain hipStreamSynchronize left
  // it idle for ~50 us per forward: profiles/r02_call22_timeline_gaps.txt).
  int N = 0;
  unsigned n_ticket = 0;
  ISF_TRY(post_int(a, occ.total, st, &n_ticket));   // (value, ticket) into pinned host memory: no copy command
  int32_t* pt2vox = pt2vox_out;
  if (!pt2vox) ISF_TRY(a.alloc_n(&pt2vox, (size_t)P));
  int32_t* slot;
  float* recs;
  uint32_t *cnt, *start;
  float4* mean4;
  float* vmax1;
  ISF_TRY(a.alloc_n(&slot, (size_t)P));
  ISF_TRY(a.alloc_n(&recs, (size_t)P * kRec));
  ISF_TRY(a.alloc_n(&cnt, (size_t)P + 1));
  ISF_HIP_TRY(hipMemsetAsync(cnt, 0, ((size_t)P + 1) * sizeof(uint32_t), st));
  hipLaunchKernelGGL(vfe_count_kernel, dim3(ceil_div(P, 256)), dim3(256), 0, st, coors4, P, occ.D, occ.H, occ.W,
                     occ.bits, occ.prefix, pt2vox, slot, cnt, voxel_coors);
  ISF_LAUNCH_CHECK();
  if (coords_ready) {   // occupancy index + voxel coords are final here: the encoder's geometry can start now
    ISF_TRY(pooled_event(a, coords_ready));
    ISF_HIP_TRY(hipEventRecord(*coords_ready, st));
  }
  ISF_TRY(wait_int(a, n_ticket, st, &N));
  *n_host = N;
  if (occ_out) *occ_out = occ;
  if (N == 0) return ISF_OK;   // vfe_count_kernel has marked every point -1
  ISF_TRY(a.alloc_n(&start, (size_t)N + 2));
  ISF_TRY(a.alloc_n(&mean4, (size_t)N));
  ISF_TRY(a.alloc_n(&vmax1, (size_t)N * kC));
  ISF_TRY(scan_u32_exclusive(a, cnt, start, (size_t)N, st));  // start[N] = number of in-range points
  hipLaunchKernelGGL(vfe_order_kernel<CIN>, dim3(ceil_div(P, 256)), dim3(256), 0, st, points, pt2vox, slot, P, start,
                     recs);
  hipLaunchKernelGGL(vfe_mean_kernel<CIN>, dim3(ceil_div(N, 256)), dim3(256), 0, st, recs, start, N, mean4);
  const int* n_valid = reinterpret_cast<const int*>(start + N);
  // rows of the voxels cut by a 64-record boundary start from zero (atomicMax), every other row is stored whole
  static_assert(kRec == 8 && kC == 64, "vfe_zero_cut_rows_kernel indexes records / rows with these sizes");
  hipLaunchKernelGGL(vfe_zero_cut_rows_kernel, dim3(ceil_div(ceil_div(P, 64), 4)), dim3(256), 0, st, recs, n_valid, vmax1,
                     voxel_feats);
  hipLaunchKernelGGL(vfe_layer1_kernel<CIN>, dim3(ceil_div(P, kL1Threads)), dim3(kL1Threads), 0, st, recs,
                     voxel_coors, n_valid, mean4, g, w1p, sc1, shift1, vmax1);
  hipLaunchKernelGGL(vfe_layer2_kernel<CIN>, dim3(ceil_div(P, 64 * kL2Waves)), dim3(64 * kL2Waves), 0, st, recs,
                     voxel_coors, n_valid, mean4, g, w1p, sc1, shift1, vmax1, w2p, sc2, shift2, voxel_feats,
                     reinterpret_cast<_Float16*>(voxel_feats_split));
  if (voxel_feats_split)
    hipLaunchKernelGGL(vfe_cut_rows_split_kernel, dim3(ceil_div(ceil_div(P, 64), 4)), dim3(256), 0, st, recs, n_valid,
                       voxel_feats, reinterpret_cast<_Float16*>(voxel_feats_split));
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int dynamic_vfe_impl(Arena& a, const float* points, const int32_t* coors4, int P, int Cin, int B,
                     const float vs[3], const float range[6], const float* w1, const float* scale1,
                     const float* shift1, int c1, const float* w2, const float* scale2,
                     const float* shift2, int c2, float* voxel_feats, int32_t* voxel_coors,
                     int32_t* pt2vox, int* num_voxels_host, OccIndex* occ_out, int grid_d_alloc,
                     hipStream_t st, hipEvent_t* coords_ready, void* voxel_feats_split, const VoxBatch* voxelize) {
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
  const VoxGeom vg = make_geom(vs, range);
  if (Cin == 5)
    return vfe_run<5>(a, points, coors4, P, B, grid, g, w1, scale1, shift1, w2, scale2, shift2,
                      voxel_feats, voxel_coors, pt2vox, num_voxels_host, occ_out, grid_d_alloc, st, coords_ready,
                      voxel_feats_split, voxelize, &vg);
  return vfe_run<4>(a, points, coors4, P, B, grid, g, w1, scale1, shift1, w2, scale2, shift2, voxel_feats,
                    voxel_coors, pt2vox, num_voxels_host, occ_out, grid_d_alloc, st, coords_ready, voxel_feats_split,
                    voxelize, &vg);
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
  isf::Arena& a = isf::arena_for_stream(isf::as_stream(stream));
  ISF_TRY(a.reset());
  return isf::dynamic_vfe_impl(a, points, coors4, num_points, in_channels, batch_size, voxel_size_host,
                               coors_range_host, w1, scale1, shift1, c1, w2, scale2, shift2, c2,
                               voxel_feats, voxel_coors, pt2vox, num_voxels_host, nullptr, 0,
                               isf::as_stream(stream));
}

}  // extern "C"
