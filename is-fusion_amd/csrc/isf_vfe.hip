// isf_vfe.hip -- A4 DynamicVFE.forward fused (voxel_encoder.py:453-547).
//
// Reference data flow: 3x (unique_dim sort + atomic scatter) + 2 dense int64 canvases of B*D*H*W entries
// (663 MB/sample) to map voxels back to points.  Here:
//   mark bitmap -> popcount scan            voxel id = rank, sorted (b,z,y,x) order, no sort, no canvas
//   count / scan / order                    points grouped by voxel (counting sort: 1 int atomic per point)
//   mean        wave per 64 sorted points   exact 2^-24 fixed-point int64 sums -> order independent; segmented scan over
//                                           the lanes, the wave holding a voxel's first record follows its run
//   layer 1     64 sorted points / wave     11 features -> Linear+BN+ReLU on the f16 matrix cores (hi/lo split, fp32-class),
//                                           TRANSPOSED: out^T = W x in^T, so the result's C/D registers are ...
//   layer 2     64 sorted points / wave     ... the B operand of layer 2: h1 recomputed (never stored: 307 MB at
//                                           P=1.2M), [h1 | vmax1[voxel]] (128) x W2^T with W2's fragments in LDS,
//                                           persistent workgroups, the next tile's loads in flight, BN+ReLU,
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
static constexpr double kFix = 16777216.0;  // 2^24 fixed point for the exact coordinate sums

struct VfeGeom {
  float vx, vy, vz, ox, oy, oz;  // voxel size, centre offsets (vs/2 + range_min)
};

// hi = f16(x), lo = f16(x - hi), two values per register: v_cvt_pk_f16_f32 and one v_fma_mix{lo,hi}_f16 per value (the
// mixed-precision FMA takes hi as an f16 operand: f16(fma(hi, -1, x)), x - hi being exact in fp32) -- three
// instructions per pair where convert / convert back / subtract / convert took seven.
__device__ __forceinline__ void vfe_split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  asm("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
      "v_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(hi), "=&v"(lo)
      : "v"(x0), "v"(x1));
}

__device__ __forceinline__ void vfe_split8(const f32x8 v, uint4& hi, uint4& lo) {
  vfe_split2(v[0], v[1], hi.x, lo.x);
  vfe_split2(v[2], v[3], hi.y, lo.y);
  vfe_split2(v[4], v[5], hi.z, lo.z);
  vfe_split2(v[6], v[7], hi.w, lo.w);
}

// ------------------------------------------------------------------------------------------ weight prep
// The two layers run TRANSPOSED on the matrix cores: out^T[channel][point] = W[channel][k] x in^T[k][point], so the
// C/D registers of layer 1 (lane (n, g) = (lane & 15, lane >> 4): point n of the group, channels 16 ct + 4 g + t) ARE
// the B operand of layer 2 -- no trip through LDS, no per-point register rows.  The reduction index of layer 2 is
// permuted to fit: slot (g, jj) of 32-channel chunk c < 2 holds layer-1 channel 16 (2c + (jj >> 2)) + 4 g + (jj & 3);
// chunks 2, 3 (the voxel's layer-1 max) keep the natural order 64 + 32 (c - 2) + 8 g + jj.
// w2p[c][ct][hi|lo][lane][8] = split(w2[16 ct + (lane & 15)][vfe_k2(c, lane >> 4, jj)] * 2^sw): A fragments of
// v_mfma_f32_16x16x32_f16;  sc2[o] = scale2[o] * 2^-sw
// w1p[ct][hi|lo][lane][4] = split(w1[16 ct + (lane & 15)][4 (lane >> 4) + jj] * 2^sw1) (zero for k >= F): A fragments of
// v_mfma_f32_16x16x16_f16;  sc1[o] = scale1[o] * 2^-sw1
__host__ __device__ constexpr int vfe_k2(int c, int g, int jj) {
  return c < 2 ? 16 * (2 * c + (jj >> 2)) + 4 * g + (jj & 3) : 64 + 32 * (c - 2) + 8 * g + jj;
}
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
      v[jj] = w2[(size_t)(16 * nt + (lane & 15)) * (2 * kC) + vfe_k2(kc, lane >> 4, jj)] * s;
    uint4 hi, lo;
    vfe_split8(v, hi, lo);
    w2p[(size_t)(kc * 4 + nt) * 128 + lane] = hi;        // = [(c * 4 + ct) * 2 + 0][lane]
    w2p[(size_t)(kc * 4 + nt) * 128 + 64 + lane] = lo;   //   [(c * 4 + ct) * 2 + 1][lane]
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
    // (75 us at 1.2 M shuffled points, 43 of them this returning atomic: 28 G/s, the same at workgroup scope; the loads
    // and stores around it take 30: profiles/r06_vfe.txt)
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

// Per-voxel means of the sorted records, one wave per 64 records: exact 2^-24 fixed-point sums (order independent) by
// a segmented scan over the wave's lanes; a voxel belongs to the wave that holds its FIRST record, which follows a run
// that leaves its 64 records through the next waves' records until the voxel changes (the long runs are the few voxels
// next to the sensor).  (Round 5: one thread per voxel walking its records -- 32-byte strides, a wave as slow as its
// longest voxel: 45 us at 1.2 M points.)
template <int CIN>
__global__ __launch_bounds__(256) void vfe_mean_kernel(const float* __restrict__ recs,
                                                       const uint32_t* __restrict__ start, int N,
                                                       float4* __restrict__ mean4) {
  const int lane = threadIdx.x & 63;
  const long long j0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
  const long long nv = start[N];   // number of in-range points = records
  if (j0 >= nv) return;
  auto load = [&](long long j, int& v, long long& sx, long long& sy, long long& sz) {
    v = -1;
    sx = sy = sz = 0;
    if (j < nv) {
      const float4 a = reinterpret_cast<const float4*>(recs + (size_t)j * kRec)[0];
      v = __float_as_int(recs[(size_t)j * kRec + kRec - 1]);
      sx = __double2ll_rn((double)a.x * kFix);
      sy = __double2ll_rn((double)a.y * kFix);
      sz = __double2ll_rn((double)a.z * kFix);
    }
  };
  auto write = [&](int v, long long sx, long long sy, long long sz) {
    const uint32_t cnt = start[v + 1] - start[v];
    const double d = kFix * (double)cnt;
    mean4[v] = make_float4((float)((double)sx / d), (float)((double)sy / d), (float)((double)sz / d), (float)cnt);
  };
  int v;
  long long sx, sy, sz;
  load(j0 + lane, v, sx, sy, sz);
  const int vprev = j0 > 0 ? __float_as_int(recs[(size_t)(j0 - 1) * kRec + kRec - 1]) : -1;
  const int vnext = j0 + 64 < nv ? __float_as_int(recs[(size_t)(j0 + 64) * kRec + kRec - 1]) : -1;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int ov = __shfl_up(v, d, 64);
    const long long ox = __shfl_up(sx, d, 64), oy = __shfl_up(sy, d, 64), oz = __shfl_up(sz, d, 64);
    if (lane >= d && ov == v) {
      sx += ox;
      sy += oy;
      sz += oz;
    }
  }
  const int v0 = __builtin_amdgcn_readlane(v, 0), v63 = __builtin_amdgcn_readlane(v, 63);
  const int vn = __shfl_down(v, 1, 64);
  const bool end = v >= 0 && (lane == 63 || vn != v);
  const bool owned = !(v == v0 && vprev == v0);          // the run began in this wave
  const bool leaves = lane == 63 && v >= 0 && vnext == v;   // ... and goes on past it
  if (end && owned && !leaves) write(v, sx, sy, sz);
  // wave-uniform: the last run leaves the wave and is ours
  const bool follow = v63 >= 0 && vnext == v63 && !(v63 == v0 && vprev == v0);
  if (!follow) return;
  long long tx = __shfl(sx, 63, 64), ty = __shfl(sy, 63, 64), tz = __shfl(sz, 63, 64);
  for (long long j = j0 + 64; j < nv; j += 64) {
    int w;
    long long ax, ay, az;
    load(j + lane, w, ax, ay, az);
    if (w != v63) ax = ay = az = 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      ax += __shfl_xor(ax, d, 64);
      ay += __shfl_xor(ay, d, 64);
      az += __shfl_xor(az, d, 64);
    }
    tx += ax;
    ty += ay;
    tz += az;
    if (__builtin_amdgcn_readlane(w, 63) != v63) break;
  }
  if (lane == 0) write(v63, tx, ty, tz);
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

// Layer 1 on the matrix cores, transposed: W1 [64 x 16] (A, 4 channel tiles) times the wave's features^T [16 x 64]
// (B, 4 point groups of 16) as v_mfma_f32_16x16x16_f16 in the same hi/lo split arithmetic as everywhere else
// (fp32-class).  The per-lane features go through a 4 KiB wave-private LDS staging area ([point][16 hi | 16 lo] halves)
// to reach the B-fragment layout; the result stays in the C/D layout: c1[ct][t] = sum for channel 16 ct + 4 (lane >> 4)
// + t of point 16 pg + (lane & 15), before BatchNorm.  (The VALU version cost 704 FMAs per point fed by 44 scalar
// weight loads per wave.)
typedef _Float16 h4v __attribute__((ext_vector_type(4)));

template <int F>
__device__ __forceinline__ void vfe_stage_features(const float (&f)[F], bool valid, char* __restrict__ stage, int lane) {
  static_assert(F <= 16, "layer-1 features must fit one K = 16 MFMA");
  uint32_t hi[8], lo[8];
#pragma unroll
  for (int k = 0; k < 16; k += 2) {
    const float x0 = (k < F && valid) ? f[k < F ? k : 0] : 0.f;
    const float x1 = (k + 1 < F && valid) ? f[k + 1 < F ? k + 1 : 0] : 0.f;
    if (k < F) vfe_split2(x0, x1, hi[k / 2], lo[k / 2]);
    else hi[k / 2] = lo[k / 2] = 0u;
  }
  uint4* sp = reinterpret_cast<uint4*>(stage + lane * 64);
  sp[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
  sp[1] = make_uint4(hi[4], hi[5], hi[6], hi[7]);
  sp[2] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  sp[3] = make_uint4(lo[4], lo[5], lo[6], lo[7]);
}

struct VfeW1 {   // this lane's A fragments of W1, loaded once per wave
  h4v hi[4], lo[4];
  __device__ __forceinline__ void load(const uint2* __restrict__ w1p, int lane) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const uint2 h = w1p[(ct * 2 + 0) * 64 + lane], l = w1p[(ct * 2 + 1) * 64 + lane];
      hi[ct] = *reinterpret_cast<const h4v*>(&h);
      lo[ct] = *reinterpret_cast<const h4v*>(&l);
    }
  }
};

// point group pg of the staged wave: c1[ct] (4 channel tiles)
__device__ __forceinline__ void vfe_layer1_group(const char* __restrict__ stage, int pg, int lane, const VfeW1& w,
                                                 f32x4 (&c1)[4]) {
  const char* base = stage + (16 * pg + (lane & 15)) * 64 + (lane >> 4) * 8;
  const uint2 xh2 = *reinterpret_cast<const uint2*>(base), xl2 = *reinterpret_cast<const uint2*>(base + 32);
  const h4v xh = *reinterpret_cast<const h4v*>(&xh2), xl = *reinterpret_cast<const h4v*>(&xl2);
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(w.hi[ct], xl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(w.lo[ct], xh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(w.hi[ct], xh, c, 0, 0, 0);
    c1[ct] = c;
  }
}

// channel tiles 2 c, 2 c + 1 only (layer 2 takes layer 1 a 32-channel chunk at a time)
__device__ __forceinline__ void vfe_layer1_pair(const char* __restrict__ stage, int pg, int lane, const VfeW1& w, int c,
                                                f32x4 (&c1)[2]) {
  const char* base = stage + (16 * pg + (lane & 15)) * 64 + (lane >> 4) * 8;
  const uint2 xh2 = *reinterpret_cast<const uint2*>(base), xl2 = *reinterpret_cast<const uint2*>(base + 32);
  const h4v xh = *reinterpret_cast<const h4v*>(&xh2), xl = *reinterpret_cast<const h4v*>(&xl2);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
    a = __builtin_amdgcn_mfma_f32_16x16x16f16(w.hi[2 * c + k], xl, a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x16f16(w.lo[2 * c + k], xh, a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x16f16(w.hi[2 * c + k], xh, a, 0, 0, 0);
    c1[k] = a;
  }
}

// this lane's sixteen BatchNorm scales / shifts: channel 16 ct + 4 (lane >> 4) + t
struct VfeBn {
  float4 sc[4], sh[4];
  __device__ __forceinline__ void load(const float* __restrict__ scale, const float* __restrict__ shift, int lane) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      sc[ct] = *reinterpret_cast<const float4*>(scale + 16 * ct + 4 * (lane >> 4));
      sh[ct] = *reinterpret_cast<const float4*>(shift + 16 * ct + 4 * (lane >> 4));
    }
  }
  __device__ __forceinline__ f32x4 relu(const f32x4 x, int ct) const {
    return f32x4{fmaxf(fmaf(x[0], sc[ct].x, sh[ct].x), 0.f), fmaxf(fmaf(x[1], sc[ct].y, sh[ct].y), 0.f),
                 fmaxf(fmaf(x[2], sc[ct].z, sh[ct].z), 0.f), fmaxf(fmaf(x[3], sc[ct].w, sh[ct].w), 0.f)};
  }
};

// C/D-layout values of ONE point group -> fp32 tile [16 points][kTileStride]: this lane's four consecutive channels of a
// tile go out as one ds_write_b128 (16 writes per wave and group; the [32][65] tile of round 5 took 64 conflicting
// ds_write_b32 per half).  The segmented max reads it back channel per lane (consecutive words: conflict free).
static constexpr int kTilePts = 16;
static constexpr int kTileStride = kC + 4;                      // rows 16-byte aligned, 272 bytes apart
static constexpr int kTileBytes = kTilePts * kTileStride * 4;   // 4352
__device__ __forceinline__ void vfe_store_tile(const f32x4 (&a)[4], float* __restrict__ tile, int lane) {
  float* row = tile + (lane & 15) * kTileStride + 4 * (lane >> 4);
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) *reinterpret_cast<f32x4*>(row + 16 * ct) = a[ct];
}

// Segmented per-voxel max over the 64 voxel-sorted points of one wave; lane = channel.  `val(p)` yields this lane's
// channel of point p (>= 0: post-ReLU).  The wave's 64 voxel ids sit in one VGPR (`myvox`, lane p = point p) and are
// broadcast with v_readlane (SGPR, no LDS round trip per point); the values are read from LDS 16 points at a time
// (independent ds_reads in flight) before the serial run logic touches them.  A run that does not reach a wave
// boundary -- or whose neighbour across the boundary belongs to another voxel (`vprev`, `vnext`: the voxel ids of
// records j0-1 and j0+64) -- is written with one 256-byte row store; a run cut by the boundary uses integer
// atomicMax on the zero-initialised destination (identical result for non-negative floats).
// fill(pg) puts the points [16 pg, 16 pg + 16) into the wave's tile before val() reads them (val(p): p in 0..63).
// split_dst != nullptr: whole rows go there in the SPLIT activation format of the sparse-conv kernels (isf_common.h:
// per 32 channels 4 x 8 f16 hi then 4 x 8 f16 lo) instead of fp32 into dst -- lane = channel stores its two halves;
// rows cut by a wave boundary still accumulate in dst (fp32 atomicMax) and are converted by vfe_cut_rows_split_kernel.
__device__ __forceinline__ void vfe_store_split(_Float16* __restrict__ split_dst, int row, int lane, float m) {
  uint32_t hi, lo;   // low halves: f16(m), f16(m - hi)
  asm("v_cvt_f16_f32 %0, %2\n\t"
      "v_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel_hi:[1,0,0]"
      : "=&v"(hi), "=&v"(lo)
      : "v"(m));
  // (row: wave-uniform -> scalar base, 32-bit lane offset)
  char* base = reinterpret_cast<char*>(split_dst + (size_t)row * (2 * kC));
  unsigned short* p = reinterpret_cast<unsigned short*>(base + (uint32_t)(((lane >> 5) * 64 + (lane & 31)) * 2));
  p[0] = (unsigned short)hi;
  p[32] = (unsigned short)lo;
}

// The control flow is scalar: the run ends of the wave's 64 points are one 64-bit mask (ballot of "my voxel differs from
// the next lane's"), bit p tested with a constant index inside the unrolled point loop -- one VALU max, one scalar test
// and one branch per point (round 5 read every point's voxel id back with v_readlane and compared it twice).
template <typename FillFn, typename ValFn>
__device__ __forceinline__ void vfe_segmented_max(int myvox, int vprev, int vnext, int lane, float* __restrict__ dst,
                                                  FillFn fill, ValFn val, _Float16* __restrict__ split_dst = nullptr) {
  const int nxt = __shfl_down(myvox, 1, 64);
  // invalid points (voxel -1) only follow the last valid one: they never end a run and what they add to m is dropped
  const unsigned long long ends = __ballot(myvox >= 0 && (lane == 63 || nxt != myvox));
  const int v0 = __builtin_amdgcn_readlane(myvox, 0), v63 = __builtin_amdgcn_readlane(myvox, 63);
  // runs cut by the wave's boundaries accumulate with atomicMax (on rows zeroed by vfe_zero_cut_rows_kernel)
  unsigned long long cut = 0;
  if (v0 >= 0 && vprev == v0) cut |= ends & (0ull - ends);   // the first run's end
  if (v63 >= 0 && vnext == v63) cut |= 1ull << 63;
  float m = 0.f;
#pragma unroll
  for (int c = 0; c < 64; c += 16) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the previous group has been read
    fill(c / kTilePts);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    float vals[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) vals[q] = val(c + q);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      m = fmaxf(m, vals[q]);
      if ((ends >> (c + q)) & 1ull) {   // wave-uniform
        const int run = __builtin_amdgcn_readlane(myvox, c + q);
        if ((cut >> (c + q)) & 1ull) {
          if (m > 0.f) {
            char* base = reinterpret_cast<char*>(dst + (size_t)run * kC);
            atomicMax(reinterpret_cast<int*>(base + (uint32_t)(lane * 4)), __float_as_int(m));
          }
        } else if (split_dst) {
          vfe_store_split(split_dst, run, lane, m);
        } else {
          char* base = reinterpret_cast<char*>(dst + (size_t)run * kC);
          *reinterpret_cast<float*>(base + (uint32_t)(lane * 4)) = m;
        }
        m = 0.f;
      }
    }
  }
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

// one sorted record per lane -> the staged layer-1 features of a wave's 64 points; returns the record's voxel id
struct VfeRecord {
  float4 a, b;   // {x, y, z, f3}, {f4, -, -, voxel id}
  bool live;     // false past the last valid record
  // Every load of the VFE kernels is UNCONDITIONAL (addresses clamped into the buffers, the voxel id patched afterwards):
  // a load inside a divergent branch makes the compiler wait for all outstanding loads at the join (it cannot count
  // them), which is exactly what the tile pipeline of layer 2 must not do.  What a dead lane computes from the clamped
  // row never leaves its own matrix column, and the segmented max skips it.
  __device__ __forceinline__ void load(const float* __restrict__ recs, uint32_t j, uint32_t n_valid) {
    const uint32_t jc = j < n_valid ? j : n_valid - 1;
    a = reinterpret_cast<const float4*>(recs + (size_t)jc * kRec)[0];
    b = reinterpret_cast<const float4*>(recs + (size_t)jc * kRec)[1];
    live = j < n_valid;
  }
  // (not evaluated inside load(): the first use of a loaded value is where the wave waits for it)
  __device__ __forceinline__ int voxel() const { return live ? __float_as_int(b.w) : -1; }   // -1: no point
  __device__ __forceinline__ int row() const { return __float_as_int(b.w); }   // always a row that exists
};

template <int CIN>
__device__ __forceinline__ void vfe_stage_record(const VfeRecord& r, int4 c, float4 mean, VfeGeom g,
                                                 char* __restrict__ stage, int lane) {
  const float rec[kRec] = {r.a.x, r.a.y, r.a.z, r.a.w, r.b.x, r.b.y, r.b.z, r.b.w};
  float f[CIN + 6];
  vfe_point_features<CIN>(rec, c, mean, g, f);
  vfe_stage_features<CIN + 6>(f, r.voxel() >= 0, stage, lane);
}

template <int CIN>
__global__ __launch_bounds__(kL1Threads) void vfe_layer1_kernel(
    const float* __restrict__ recs, const int32_t* __restrict__ voxel_coors, const int* __restrict__ n_valid,
    const float4* __restrict__ mean4, VfeGeom g, const uint2* __restrict__ w1p, const float* __restrict__ sc1,
    const float* __restrict__ shift1, float* __restrict__ vmax1) {
  __shared__ __attribute__((aligned(16))) char smem[(kL1Threads / 64) * (4096 + kTileBytes)];
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const long long wj0 = (long long)blockIdx.x * kL1Threads + wave * 64;
  const uint32_t nv = (uint32_t)*n_valid;
  if (wj0 >= (long long)nv) return;   // wave-uniform; the LDS areas are wave-private, no block barrier below
  const uint32_t j = (uint32_t)wj0 + lane;
  char* stage = smem + wave * (4096 + kTileBytes);
  float* wt = reinterpret_cast<float*>(stage + 4096);
  VfeRecord r;
  r.load(recs, j, nv);
  const int v = r.voxel();
  const int4 c = reinterpret_cast<const int4*>(voxel_coors)[r.row()];
  const float4 m = mean4[r.row()];
  VfeW1 w1;
  w1.load(w1p, lane);
  VfeBn bn;
  bn.load(sc1, shift1, lane);
  vfe_stage_record<CIN>(r, c, m, g, stage, lane);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  vfe_segmented_max(v, vfe_record_voxel(recs, wj0 - 1, nv), vfe_record_voxel(recs, wj0 + 64, nv), lane, vmax1,
                    [&](int pg) {   // layer 1 of one point group, just in time: sixteen live sums instead of sixty-four
                      f32x4 h1[4];
                      vfe_layer1_group(stage, pg, lane, w1, h1);
#pragma unroll
                      for (int ct = 0; ct < 4; ++ct) h1[ct] = bn.relu(h1[ct], ct);
                      vfe_store_tile(h1, wt, lane);
                    },
                    [&](int p) { return wt[(p % kTilePts) * kTileStride + lane]; });
}

// ------------------------------------------------------------------------------------------ layer 2 (MFMA)
// One wave = 64 sorted points = 4 point groups; K = 128 = [h1 (64) | vmax1[voxel] (64)] in four 32-channel chunks.
// Layer 1 is recomputed (its [P, 64] output never goes to HBM) and handed to layer 2 in registers (see vfe_k2).
// Workgroups are persistent (two per CU): W2's 32 KiB of A fragments sit in LDS, loaded once, and every wave walks
// over its share of the 64-point tiles with the loads of the NEXT tile in flight: its records are requested at the top
// of a tile, its voxel coordinates / means after the first two chunks, its features are staged before the segmented
// max; the voxel-max rows of chunk 2 are requested before layer 1, those of chunk 3 before chunk 2 multiplies.  (At
// two waves per SIMD nothing else hides a load: with every load waited for where it was issued the parts of a tile
// simply added up -- records 21 us, voxel-max rows 17, layer 1 12, layer 2 24, tile + segmented max 41 of 126 us at
// 1.2 M points: profiles/r06_vfe.txt.)  LDS per wave: staging area 4 KiB, one [16][68] fp32 tile for the segmented
// max, the wave's 64 voxel ids.
// (Round 5's version turned the layer-1 output into one point per lane -- 64 + 64 registers of rows, 128 ds_reads, four
// 8-KiB trips through an A tile: 204 registers, 181 us.)
static constexpr int kL2Waves = 4;
static constexpr int kL2WaveBytes = 4096 + kTileBytes + 256;    // 8704
static constexpr int kL2W2Bytes = 4 * 4 * 2 * 64 * 16;          // 32768
static constexpr int kL2Lds = kL2W2Bytes + kL2Waves * kL2WaveBytes;

template <int CIN>
__global__ __launch_bounds__(64 * kL2Waves) __attribute__((amdgpu_waves_per_eu(2, 2))) void vfe_layer2_kernel(
    const float* __restrict__ recs, const int32_t* __restrict__ voxel_coors, const int* __restrict__ n_valid,
    const float4* __restrict__ mean4, VfeGeom g, const uint2* __restrict__ w1p, const float* __restrict__ sc1,
    const float* __restrict__ shift1, const float* __restrict__ vmax1, const uint4* __restrict__ w2p,
    const float* __restrict__ sc2, const float* __restrict__ shift2, float* __restrict__ out,
    _Float16* __restrict__ out_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint4* w2s = reinterpret_cast<uint4*>(smem);   // [(c * 4 + ct) * 2 + hi|lo][lane]
  for (int i = threadIdx.x; i < kL2W2Bytes / 16; i += 64 * kL2Waves) w2s[i] = w2p[i];
  __syncthreads();   // the only block barrier; everything below is wave-private
  char* stage = smem + kL2W2Bytes + wave * kL2WaveBytes;
  float* ftile = reinterpret_cast<float*>(stage + 4096);   // [16][68]
  int* vox = reinterpret_cast<int*>(stage + 4096 + kTileBytes);
  const int n = lane & 15, kg = lane >> 4;
  const uint32_t nv = (uint32_t)*n_valid;
  const uint32_t stride = gridDim.x * kL2Waves * 64;
  uint32_t j0 = (blockIdx.x * kL2Waves + wave) * 64;
  if (j0 >= nv) return;
  VfeW1 w1;
  w1.load(w1p, lane);
  VfeBn bn;
  bn.load(sc1, shift1, lane);
  const float sc = sc2[lane], sh = shift2[lane];

  int v;
  {   // prologue: the first tile's records, staged
    VfeRecord r;
    r.load(recs, j0 + lane, nv);
    v = r.voxel();
    const int4 c = reinterpret_cast<const int4*>(voxel_coors)[r.row()];
    const float4 m = mean4[r.row()];
    vfe_stage_record<CIN>(r, c, m, g, stage, lane);
  }

  for (;;) {
    const uint32_t jn = j0 + stride;
    const bool more = jn < nv;   // wave-uniform
    VfeRecord rn;
    rn.load(recs, jn + lane, nv);   // (1) the next tile's records (the last tile re-reads the last record)
    vox[lane] = v;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    int vp[4];
#pragma unroll
    for (int pg = 0; pg < 4; ++pg) vp[pg] = vox[16 * pg + n];
    // the voxel's layer-1 max (map_voxel_center_to_point gather, voxel_encoder.py:541-544): this lane's eight channels
    // 32 c + 8 kg .. + 8 of the row of point 16 pg + n's voxel
    auto vmax_rows = [&](int c, f32x8 (&r)[4]) {
#pragma unroll
      for (int pg = 0; pg < 4; ++pg) {
        const float4* src = reinterpret_cast<const float4*>(vmax1 + (size_t)(vp[pg] < 0 ? 0 : vp[pg]) * kC + 32 * c + 8 * kg);
        const float4 r0 = src[0], r1 = src[1];
        r[pg] = f32x8{r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
      }
    };
    f32x8 vm2[4];
    vmax_rows(0, vm2);   // (2) requested before layer 1
    __builtin_amdgcn_sched_barrier(0);

    f32x4 acc[4][4];   // [point group][channel tile]
#pragma unroll
    for (int pg = 0; pg < 4; ++pg)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[pg][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto chunk = [&](int c, const uint4 (&bh)[4], const uint4 (&bl)[4]) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const uint4 ahu = w2s[((c * 4 + ct) * 2 + 0) * 64 + lane];
        const uint4 alu = w2s[((c * 4 + ct) * 2 + 1) * 64 + lane];
        const h8 a_h = *reinterpret_cast<const h8*>(&ahu);
        const h8 a_l = *reinterpret_cast<const h8*>(&alu);
#pragma unroll
        for (int pg = 0; pg < 4; ++pg) {
          const h8 b_h = *reinterpret_cast<const h8*>(&bh[pg]);
          const h8 b_l = *reinterpret_cast<const h8*>(&bl[pg]);
          acc[pg][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h, b_l, acc[pg][ct], 0, 0, 0);
          acc[pg][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_l, b_h, acc[pg][ct], 0, 0, 0);
          acc[pg][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_h, b_h, acc[pg][ct], 0, 0, 0);
        }
      }
    };
    // chunks 0, 1: layer 1's channel tiles 2 c, 2 c + 1 of every point group -> B fragments -> 48 MFMAs
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      uint4 bh[4], bl[4];
#pragma unroll
      for (int pg = 0; pg < 4; ++pg) {
        f32x4 c1[2];
        vfe_layer1_pair(stage, pg, lane, w1, c, c1);
        const f32x4 a = bn.relu(c1[0], 2 * c), b = bn.relu(c1[1], 2 * c + 1);
        vfe_split8(f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}, bh[pg], bl[pg]);
      }
      chunk(c, bh, bl);
    }
    // (3) the next tile's voxel coordinates and means (its records have arrived under the multiplies above)
    __builtin_amdgcn_sched_barrier(0);
    int vrow = rn.row();
    asm volatile("" : "+v"(vrow));   // the wait for the records belongs here, not where the compiler would extend the index
    const int vn = rn.live ? vrow : -1;
    const int4 cn = reinterpret_cast<const int4*>(voxel_coors)[vrow];
    const float4 mn = mean4[vrow];
    {
      f32x8 vm3[4];
      vmax_rows(1, vm3);   // (4) requested before chunk 2 multiplies
      __builtin_amdgcn_sched_barrier(0);
      uint4 bh[4], bl[4];
#pragma unroll
      for (int pg = 0; pg < 4; ++pg) vfe_split8(vm2[pg], bh[pg], bl[pg]);
      chunk(2, bh, bl);
#pragma unroll
      for (int pg = 0; pg < 4; ++pg) vfe_split8(vm3[pg], bh[pg], bl[pg]);
      chunk(3, bh, bl);
    }
    // (5) the next tile's features into the staging area (layer 1 of this tile has read it)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("" : "+v"(rn.a.x), "+v"(rn.a.y), "+v"(rn.a.z), "+v"(rn.a.w), "+v"(rn.b.x));   // (nothing of it earlier)
    vfe_stage_record<CIN>(rn, cn, mn, g, stage, lane);

    // accumulators (lane (n, kg): channels 16 ct + 4 kg + t of point 16 pg + n) -> fp32 tile, one point group at a
    // time; BatchNorm + ReLU where the segmented max reads them back channel per lane
    vfe_segmented_max(v, vfe_record_voxel(recs, (long long)j0 - 1, nv), vfe_record_voxel(recs, (long long)j0 + 64, nv),
                      lane, out,
                      [&](int pg) {
                        switch (pg) {   // constant after unrolling
                          case 0: vfe_store_tile(acc[0], ftile, lane); break;
                          case 1: vfe_store_tile(acc[1], ftile, lane); break;
                          case 2: vfe_store_tile(acc[2], ftile, lane); break;
                          default: vfe_store_tile(acc[3], ftile, lane); break;
                        }
                      },
                      [&](int p) { return fmaxf(fmaf(ftile[(p % kTilePts) * kTileStride + lane], sc, sh), 0.f); },
                      out_split);
    if (!more) break;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    v = vn;
    j0 = jn;
  }
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
  // one 256-thread workgroup, 14 us of latency: on the workspace's side stream, beside the index scan and the counting
  // sort; the layer kernels wait for it
  hipStream_t sprep = nullptr;
  ISF_TRY(side_stream(a, &sprep));
  ISF_TRY(stream_wait_stream(a, sprep, st));   // the weights are ready where the caller's stream stands
  hipLaunchKernelGGL(vfe_prep_kernel, dim3(1), dim3(256), 0, sprep, w1, F, w2, scale1, scale2, w1p, sc1, w2p, sc2);
  hipEvent_t prep_done;
  ISF_TRY(pooled_event(a, &prep_done));
  ISF_HIP_TRY(hipEventRecord(prep_done, sprep));
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
  ISF_HIP_TRY(hipStreamWaitEvent(st, prep_done, 0));   // (before any return: the packed weights live in this call's arena)
  if (N == 0) return ISF_OK;   // vfe_count_kernel has marked every point -1
  ISF_TRY(a.alloc_n(&start, (size_t)N + 2));
  ISF_TRY(a.alloc_n(&mean4, (size_t)N));
  ISF_TRY(a.alloc_n(&vmax1, (size_t)N * kC));
  ISF_TRY(scan_u32_exclusive(a, cnt, start, (size_t)N, st));  // start[N] = number of in-range points
  hipLaunchKernelGGL(vfe_order_kernel<CIN>, dim3(ceil_div(P, 256)), dim3(256), 0, st, points, pt2vox, slot, P, start,
                     recs);
  hipLaunchKernelGGL(vfe_mean_kernel<CIN>, dim3(ceil_div(P, 256)), dim3(256), 0, st, recs, start, N, mean4);
  const int* n_valid = reinterpret_cast<const int*>(start + N);
  // rows of the voxels cut by a 64-record boundary start from zero (atomicMax), every other row is stored whole
  static_assert(kRec == 8 && kC == 64, "vfe_zero_cut_rows_kernel indexes records / rows with these sizes");
  hipLaunchKernelGGL(vfe_zero_cut_rows_kernel, dim3(ceil_div(ceil_div(P, 64), 4)), dim3(256), 0, st, recs, n_valid, vmax1,
                     voxel_feats);
  hipLaunchKernelGGL(vfe_layer1_kernel<CIN>, dim3(ceil_div(P, kL1Threads)), dim3(kL1Threads), 0, st, recs,
                     voxel_coors, n_valid, mean4, g, w1p, sc1, shift1, vmax1);
  {
    static bool lds_set[2] = {false, false};   // per instantiation; the attribute is idempotent, a race is harmless
    if (!lds_set[CIN - 4]) {
      ISF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&vfe_layer2_kernel<CIN>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kL2Lds));
      lds_set[CIN - 4] = true;
    }
    const int tiles = ceil_div(P, 64 * kL2Waves);
    hipLaunchKernelGGL(vfe_layer2_kernel<CIN>, dim3(tiles < 512 ? tiles : 512), dim3(64 * kL2Waves), kL2Lds, st, recs,
                       voxel_coors, n_valid, mean4, g, w1p, sc1, shift1, vmax1, w2p, sc2, shift2, voxel_feats,
                       reinterpret_cast<_Float16*>(voxel_feats_split));
  }
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
