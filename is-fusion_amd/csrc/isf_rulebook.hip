// isf_rulebook.hip -- A5 sparse-conv rulebook as an OUTPUT-stationary neighbour table.
//
// Reference (spconv 1.x, indice.cu.h:22-203 / geometry.h): a dense int32 grid of B*D*H*W entries per
// call (340 MB/sample at level 0) + per-tap (in,out) pair lists built with atomics, and a sort-unique
// for strided convs.  Here the active set of a level lives in a 1-bit-per-cell occupancy index whose
// popcount rank IS the row number (sorted (b,z,y,x) order), so
//   - SubM:    nbr[k][o] = rank(coord(o) + offset(k))                   (27 bit tests per voxel)
//   - strided: mark out = (in + pad - k)/stride in a fresh bitmap -> scan -> rows; nbr by lookup.
// Rows are sorted by (b,z,y,x), so consecutive threads probe neighbouring words: the probes are
// L2-local.  The table layout nbr[K][stride] makes the conv kernel's per-tap tile loads contiguous.
#include "isf_common.h"

namespace isf {

struct RbGeom {
  int ks[3], st[3], pd[3];
  int in_shape[3], out_shape[3];
  int batch;   // rows whose batch index is outside [0, batch) have no neighbours
};

__global__ void rb_perm_kernel(const int32_t* __restrict__ coors4, int n, int B, int D, int H, int W,
                               const unsigned long long* __restrict__ bits,
                               const uint32_t* __restrict__ prefix, int32_t* __restrict__ perm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 c = reinterpret_cast<const int4*>(coors4)[i];
  // rows outside the grid (dynamic voxelize emits -1 for invalid points; callers may pass anything) own no cell:
  // the same guard as occ_mark_coords4, so the unsigned cell index can never leave the bitmap
  if (c.x < 0 || c.y < 0 || c.z < 0 || c.w < 0 || c.x >= B || c.y >= D || c.z >= H || c.w >= W) return;
  const int r = occ_lookup(bits, prefix, (((unsigned long long)c.x * D + c.y) * H + c.z) * W + c.w);
  if (r >= 0) perm[r] = i;
}

// one thread per (output row, (kz, ky) line of taps): grid (ceil(stride/256), ks[0] * ks[1]).  The ks[2] taps of a line
// probe x-adjacent cells -- the same 64-bit bitmap word (and prefix entry) except when the line crosses a word
// boundary -- so a line costs one coordinate load and ONE pair of dependent loads instead of ks[2] of each (the
// (row, tap) grid of round 1 read every coordinate 27 times: 101 us for 44 MB at level 0).  Rows are sorted, so the
// 256 probes of a block hit neighbouring bitmap words.  Rows >= n_out up to nbr_stride are filled with -1.
__global__ __launch_bounds__(256) void rb_nbr_kernel(const int32_t* __restrict__ out_coors4, int n_out,
                                                     RbGeom g,
                                                     const unsigned long long* __restrict__ in_bits,
                                                     const uint32_t* __restrict__ in_prefix,
                                                     const int32_t* __restrict__ perm,
                                                     int32_t* __restrict__ nbr, int nbr_stride,
                                                     uint32_t* __restrict__ block_pairs) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  const int line = blockIdx.y;
  const int ky = line % g.ks[1], kz = line / g.ks[1];
  const int nx = g.ks[2];                                   // <= 3
  int r[3] = {-1, -1, -1};
  if (o < n_out) {
    const int4 c = reinterpret_cast<const int4*>(out_coors4)[o];
    const int iz = c.y * g.st[0] - g.pd[0] + kz;
    const int iy = c.z * g.st[1] - g.pd[1] + ky;
    const int ix0 = c.w * g.st[2] - g.pd[2];
    if (c.x >= 0 && c.x < g.batch && iz >= 0 && iz < g.in_shape[0] && iy >= 0 && iy < g.in_shape[1]) {
      const long long base = (((long long)c.x * g.in_shape[0] + iz) * g.in_shape[1] + iy) * g.in_shape[2];
      long long wcur = -1;
      unsigned long long word = 0;
      uint32_t pre = 0;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ix0 + kx;
        if (kx < nx && ix >= 0 && ix < g.in_shape[2]) {
          const long long cell = base + ix;
          if ((cell >> 6) != wcur) {
            wcur = cell >> 6;
            word = in_bits[wcur];
            pre = in_prefix[wcur];
          }
          const unsigned long long bit = 1ull << (cell & 63);
          if (word & bit) {
            const int v = (int)(pre + (uint32_t)__popcll(word & (bit - 1)));
            r[kx] = perm ? perm[v] : v;
          }
        }
      }
    }
  }
  int found = 0;
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    if (kx < nx) {
      if (o < nbr_stride) nbr[(size_t)(line * nx + kx) * nbr_stride + o] = r[kx];
      found += r[kx] >= 0;
    }
  }
  if (block_pairs) {  // per-block partial, no atomics: a single counter word serialises at ~88 updates/us
    __shared__ int wsum[4];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) found += __shfl_xor(found, d, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = found;
    __syncthreads();
    if (threadIdx.x == 0)
      block_pairs[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = (uint32_t)(wsum[0] + wsum[1] + wsum[2] + wsum[3]);
  }
}

__global__ __launch_bounds__(1024) void rb_sum_pairs_kernel(const uint32_t* __restrict__ block_pairs, int n,
                                                            unsigned long long* __restrict__ pair_count) {
  __shared__ unsigned long long ws[16];
  unsigned long long s = 0;
  for (int i = threadIdx.x; i < n; i += 1024) s += block_pairs[i];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < 16; ++w) t += ws[w];
    *pair_count = t;
  }
}

// strided conv: every (input voxel, tap) marks the output it feeds.  One thread per (input row, kz): the ks[1] * ks[2]
// taps of a plane are arithmetic on ONE coordinate load (the (row, tap) grid of round 1 loaded it 27 times), and with
// stride 2 at most two taps per axis land on an output at all.  grid (ceil(n_in/256), ks[0])
__global__ __launch_bounds__(256) void rb_mark_out_kernel(const int32_t* __restrict__ in_coors4, int n_in,
                                                          RbGeom g,
                                                          unsigned long long* __restrict__ out_bits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_in) return;
  const int kz = blockIdx.y;
  const int4 c = reinterpret_cast<const int4*>(in_coors4)[i];
  if (c.x < 0 || c.x >= g.batch || c.y < 0 || c.z < 0 || c.w < 0 || c.y >= g.in_shape[0] || c.z >= g.in_shape[1] ||
      c.w >= g.in_shape[2])
    return;
  const int tz = c.y + g.pd[0] - kz;
  if (tz < 0 || tz % g.st[0]) return;
  const int oz = tz / g.st[0];
  if (oz >= g.out_shape[0]) return;
  for (int ky = 0; ky < g.ks[1]; ++ky) {
    const int ty = c.z + g.pd[1] - ky;
    if (ty < 0 || ty % g.st[1] || ty / g.st[1] >= g.out_shape[1]) continue;
    const unsigned long long line =
        (((unsigned long long)c.x * g.out_shape[0] + oz) * g.out_shape[1] + ty / g.st[1]) * g.out_shape[2];
    for (int kx = 0; kx < g.ks[2]; ++kx) {
      const int tx = c.w + g.pd[2] - kx;
      if (tx < 0 || tx % g.st[2] || tx / g.st[2] >= g.out_shape[2]) continue;
      const unsigned long long cell = line + tx / g.st[2];
      const unsigned long long bit = 1ull << (cell & 63);
      unsigned long long* p = out_bits + (cell >> 6);
      if (!(*p & bit)) atomicOr(p, bit);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// LINE-COMPRESSED neighbour table.  The ks[2] taps of a (kz, ky) line probe x-adjacent cells, and rows are sorted by
// (b, z, y, x): the neighbours a row has through the taps of one line are CONSECUTIVE rows.  So a row needs one int32
// per line -- the row of its first present neighbour -- and one bit per tap: lines [ks0 * ks1][stride] + mask [stride]
// = 40 bytes per row for a 3 x 3 x 3 kernel instead of 108, and nbr[k][o] = mask bit k ? lines[k / nx][o] +
// popcount(mask bits of the line below k) : -1.  The narrow layers (levels 0 / 1, isf_spconv_dma.hip) read their table
// in this form: their table bytes were as many as a 32-channel row's.  One thread per output row (the nine lines share
// the coordinate load).  Only for tables in rank order (no permutation).
__global__ __launch_bounds__(256) void rb_lines_kernel(const int32_t* __restrict__ out_coors4, int n_out, RbGeom g,
                                                       const unsigned long long* __restrict__ in_bits,
                                                       const uint32_t* __restrict__ in_prefix,
                                                       int32_t* __restrict__ lines, uint32_t* __restrict__ mask,
                                                       int stride, uint32_t* __restrict__ block_pairs) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  const int nl = g.ks[0] * g.ks[1], nx = g.ks[2];
  unsigned m = 0;
  int4 c = make_int4(-1, 0, 0, 0);
  if (o < n_out) c = reinterpret_cast<const int4*>(out_coors4)[o];
  const bool live = o < n_out && c.x >= 0 && c.x < g.batch;
  const int ix0 = c.w * g.st[2] - g.pd[2];
#pragma unroll
  for (int line = 0; line < 9; ++line) {
    if (line >= nl) continue;
    const int ky = line % g.ks[1], kz = line / g.ks[1];
    const int iz = c.y * g.st[0] - g.pd[0] + kz;
    const int iy = c.z * g.st[1] - g.pd[1] + ky;
    int first = -1;
    if (live && iz >= 0 && iz < g.in_shape[0] && iy >= 0 && iy < g.in_shape[1]) {
      const long long base = (((long long)c.x * g.in_shape[0] + iz) * g.in_shape[1] + iy) * g.in_shape[2];
      long long wcur = -1;
      unsigned long long word = 0;
      uint32_t pre = 0;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ix0 + kx;
        if (kx < nx && ix >= 0 && ix < g.in_shape[2]) {
          const long long cell = base + ix;
          if ((cell >> 6) != wcur) {
            wcur = cell >> 6;
            word = in_bits[wcur];
            pre = in_prefix[wcur];
          }
          const unsigned long long bit = 1ull << (cell & 63);
          if (word & bit) {
            if (first < 0) first = (int)(pre + (uint32_t)__popcll(word & (bit - 1)));
            m |= 1u << (line * nx + kx);
          }
        }
      }
    }
    if (o < stride) lines[(size_t)line * stride + o] = first;
  }
  if (o < stride) mask[o] = m;
  if (block_pairs) {
    __shared__ int wsum[4];
    int found = __popc(m);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) found += __shfl_xor(found, d, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = found;
    __syncthreads();
    if (threadIdx.x == 0) block_pairs[blockIdx.x] = (uint32_t)(wsum[0] + wsum[1] + wsum[2] + wsum[3]);
  }
}

// full table -> line-compressed (rows >= n_out: no neighbours).  flag (optional): set to 1 when a line's neighbours are
// NOT consecutive rows (a permuted or hand-made table): the compressed form cannot express it.
__global__ __launch_bounds__(256) void rb_to_lines_kernel(const int32_t* __restrict__ nbr, int nbr_stride, int K, int nx,
                                                          int n_out, int32_t* __restrict__ lines,
                                                          uint32_t* __restrict__ mask, int* __restrict__ flag) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= nbr_stride) return;
  unsigned m = 0;
  const int nl = K / nx;
  for (int line = 0; line < nl; ++line) {
    int first = -1, cnt = 0;
    for (int kx = 0; kx < nx; ++kx) {
      const int k = line * nx + kx;
      const int v = o < n_out ? nbr[(size_t)k * nbr_stride + o] : -1;
      if (v >= 0) {
        if (first < 0) first = v;
        else if (v != first + cnt && flag) *flag = 1;
        ++cnt;
        m |= 1u << k;
      }
    }
    lines[(size_t)line * nbr_stride + o] = first;
  }
  mask[o] = m;
}

__global__ __launch_bounds__(256) void rb_from_lines_kernel(const int32_t* __restrict__ lines,
                                                            const uint32_t* __restrict__ mask, int nbr_stride, int K,
                                                            int nx, int32_t* __restrict__ nbr) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= nbr_stride) return;
  const unsigned m = mask[o];
  for (int k = 0; k < K; ++k) {
    const int line = k / nx, first = line * nx;
    const unsigned below = m & ((1u << k) - 1u) & ~((1u << first) - 1u);
    nbr[(size_t)k * nbr_stride + o] = ((m >> k) & 1u) ? lines[(size_t)line * nbr_stride + o] + __popc(below) : -1;
  }
}

static RbGeom make_rb_geom(const int in_shape[3], const int ks[3], const int st[3], const int pd[3],
                           bool subm, int batch) {
  RbGeom g;
  g.batch = batch;
  for (int j = 0; j < 3; ++j) {
    g.ks[j] = ks[j];
    g.st[j] = subm ? 1 : st[j];
    g.pd[j] = subm ? ks[j] / 2 : pd[j];  // spconv_ops.h:76-79
    g.in_shape[j] = in_shape[j];
    g.out_shape[j] = subm ? in_shape[j] : (in_shape[j] + 2 * g.pd[j] - (ks[j] - 1) - 1) / g.st[j] + 1;
  }
  return g;
}

int launch_nbr(Arena& a, const int32_t* out_coors4, int n_out, const int in_shape[3], const int ks[3],
               const int st[3], const int pd[3], bool subm, const OccIndex& in_occ, const int32_t* perm,
               int32_t* nbr, int nbr_stride, unsigned long long* pair_count, hipStream_t st_) {
  const RbGeom g = make_rb_geom(in_shape, ks, st, pd, subm, in_occ.B);
  const int lines = ks[0] * ks[1], nbx = ceil_div(nbr_stride, 256);
  ISF_REQUIRE(ks[2] >= 1 && ks[2] <= 3, ISF_ERR_UNSUPPORTED, "rulebook: kernel width %d (1..3)", ks[2]);
  uint32_t* block_pairs = nullptr;
  if (pair_count) ISF_TRY(a.alloc_n(&block_pairs, (size_t)lines * nbx));
  hipLaunchKernelGGL(rb_nbr_kernel, dim3(nbx, lines), dim3(256), 0, st_, out_coors4, n_out, g, in_occ.bits,
                     in_occ.prefix, perm, nbr, nbr_stride, block_pairs);
  if (pair_count)
    hipLaunchKernelGGL(rb_sum_pairs_kernel, dim3(1), dim3(1024), 0, st_, block_pairs, lines * nbx, pair_count);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int launch_nbr_lines(Arena& a, const int32_t* out_coors4, int n_out, const int in_shape[3], const int ks[3],
                     const int st[3], const int pd[3], bool subm, const OccIndex& in_occ, int32_t* lines, uint32_t* mask,
                     int stride, unsigned long long* pair_count, hipStream_t st_) {
  const RbGeom g = make_rb_geom(in_shape, ks, st, pd, subm, in_occ.B);
  const int nbx = ceil_div(stride, 256);
  ISF_REQUIRE(ks[2] >= 1 && ks[2] <= 3 && ks[0] * ks[1] <= 9 && ks[0] * ks[1] * ks[2] <= 27, ISF_ERR_UNSUPPORTED,
              "rulebook lines: kernel %d x %d x %d", ks[0], ks[1], ks[2]);
  uint32_t* block_pairs = nullptr;
  if (pair_count) ISF_TRY(a.alloc_n(&block_pairs, (size_t)nbx));
  hipLaunchKernelGGL(rb_lines_kernel, dim3(nbx), dim3(256), 0, st_, out_coors4, n_out, g, in_occ.bits, in_occ.prefix, lines,
                     mask, stride, block_pairs);
  if (pair_count) hipLaunchKernelGGL(rb_sum_pairs_kernel, dim3(1), dim3(1024), 0, st_, block_pairs, nbx, pair_count);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int launch_mark_out(const int32_t* in_coors4, int n_in, const int in_shape[3], const int ks[3],
                    const int st[3], const int pd[3], const OccIndex& out_occ, hipStream_t st_) {
  if (n_in <= 0) return ISF_OK;
  const RbGeom g = make_rb_geom(in_shape, ks, st, pd, false, out_occ.B);
  hipLaunchKernelGGL(rb_mark_out_kernel, dim3(ceil_div(n_in, 256), ks[0]), dim3(256), 0, st_, in_coors4, n_in, g,
                     out_occ.bits);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int build_perm(Arena& a, const OccIndex& occ, const int32_t* coors4, int n, int32_t** perm_out,
               hipStream_t st) {
  int32_t* perm = nullptr;
  ISF_TRY(a.alloc_n(&perm, (size_t)(n > 0 ? n : 1)));
  if (n > 0) {
    hipLaunchKernelGGL(rb_perm_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, coors4, n, occ.B, occ.D,
                       occ.H, occ.W, occ.bits, occ.prefix, perm);
    ISF_LAUNCH_CHECK();
  }
  *perm_out = perm;
  return ISF_OK;
}

// ------------------------------------------------------------------------------- spconv-1 interchange
// one workgroup per tap: ordered compaction of the valid rows of nbr[k][:]
__global__ __launch_bounds__(1024) void rb_to_pairs_kernel(const int32_t* __restrict__ nbr, int nbr_stride,
                                                           int n_out, int n_in,
                                                           int32_t* __restrict__ pairs,
                                                           int32_t* __restrict__ num) {
  __shared__ int wave_cnt[16];
  __shared__ int base_s;
  const int k = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int32_t* pin = pairs + ((size_t)k * 2 + 0) * n_in;
  int32_t* pout = pairs + ((size_t)k * 2 + 1) * n_in;
  if (threadIdx.x == 0) base_s = 0;
  __syncthreads();
  for (int o0 = 0; o0 < n_out; o0 += 1024) {
    const int o = o0 + threadIdx.x;
    const int v = o < n_out ? nbr[(size_t)k * nbr_stride + o] : -1;
    const unsigned long long m = __ballot(v >= 0);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    if (v >= 0) {
      const int pos = off + __popcll(m & ((1ull << lane) - 1));
      pin[pos] = v;
      pout[pos] = o;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < 16; ++w) t += wave_cnt[w];
      base_s += t;
    }
    __syncthreads();
  }
  const int total = base_s;
  for (int s = total + threadIdx.x; s < n_in; s += 1024) { pin[s] = -1; pout[s] = -1; }
  if (threadIdx.x == 0) num[k] = total;
}

__global__ void rb_from_pairs_kernel(const int32_t* __restrict__ pairs, const int32_t* __restrict__ num,
                                     int K, int n_in, int n_out, int32_t* __restrict__ nbr,
                                     int nbr_stride) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)K * n_in) return;
  const int k = (int)(t / n_in), s = (int)(t % n_in);
  if (s >= num[k]) return;
  const int i = pairs[((size_t)k * 2 + 0) * n_in + s];
  const int o = pairs[((size_t)k * 2 + 1) * n_in + s];
  if (i >= 0 && o >= 0 && o < n_out) nbr[(size_t)k * nbr_stride + o] = i;
}

}  // namespace isf

extern "C" {

int isf_nbr_stride(int num_rows) { return (int)isf::round_up((size_t)(num_rows > 0 ? num_rows : 1), 128); }

int isf_conv_out_shape(const int in_shape_host[3], const int ksize_host[3], const int stride_host[3],
                       const int padding_host[3], int out_shape_host[3]) {
  if (!in_shape_host || !ksize_host || !stride_host || !padding_host || !out_shape_host) return ISF_ERR_ARG;
  for (int j = 0; j < 3; ++j) {
    if (stride_host[j] <= 0 || ksize_host[j] <= 0) return ISF_ERR_ARG;
    out_shape_host[j] = (in_shape_host[j] + 2 * padding_host[j] - (ksize_host[j] - 1) - 1) / stride_host[j] + 1;
  }
  return ISF_OK;
}

int isf_build_rulebook(const int32_t* indices, int num_in, int batch_size, const int spatial_shape_host[3],
                       const int ksize_host[3], const int stride_host[3], const int padding_host[3],
                       int conv_type, int32_t* out_indices, int out_capacity, int32_t* nbr, int nbr_stride,
                       int* num_out_host, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(num_in >= 0 && batch_size > 0 && spatial_shape_host && ksize_host && stride_host &&
                  padding_host && nbr && num_out_host,
              ISF_ERR_ARG, "build_rulebook: bad arguments");
  const int K = ksize_host[0] * ksize_host[1] * ksize_host[2];
  ISF_REQUIRE(K >= 1 && K <= 27, ISF_ERR_UNSUPPORTED, "build_rulebook: kernel volume %d (max 27)", K);
  hipStream_t st = as_stream(stream);
  Arena& a = arena_for_stream(as_stream(stream));
  ISF_TRY(a.reset());
  const bool subm = conv_type == ISF_CONV_SUBM;
  *num_out_host = 0;
  if (num_in == 0) {
    ISF_HIP_TRY(hipMemsetAsync(nbr, 0xff, (size_t)K * nbr_stride * sizeof(int32_t), st));
    return ISF_OK;
  }
  ISF_REQUIRE(indices, ISF_ERR_ARG, "build_rulebook: null indices");
  OccIndex in_occ;
  ISF_TRY(occ_create(a, &in_occ, batch_size, spatial_shape_host[0], spatial_shape_host[1],
                     spatial_shape_host[2], st));
  ISF_TRY(occ_mark_coords4(in_occ, indices, num_in, st));
  ISF_TRY(occ_scan(a, in_occ, st));
  int32_t* perm = nullptr;
  ISF_TRY(build_perm(a, in_occ, indices, num_in, &perm, st));
  if (subm) {
    ISF_REQUIRE(nbr_stride >= isf_nbr_stride(num_in), ISF_ERR_CAPACITY, "build_rulebook: nbr_stride too small");
    ISF_TRY(launch_nbr(a, indices, num_in, spatial_shape_host, ksize_host, stride_host, padding_host, true,
                       in_occ, perm, nbr, nbr_stride, nullptr, st));
    if (out_indices && out_indices != indices)
      ISF_HIP_TRY(hipMemcpyAsync(out_indices, indices, (size_t)num_in * 4 * sizeof(int32_t),
                                 hipMemcpyDeviceToDevice, st));
    *num_out_host = num_in;
    return ISF_OK;
  }
  int out_shape[3];
  ISF_TRY(isf_conv_out_shape(spatial_shape_host, ksize_host, stride_host, padding_host, out_shape));
  ISF_REQUIRE(out_shape[0] > 0 && out_shape[1] > 0 && out_shape[2] > 0, ISF_ERR_ARG,
              "build_rulebook: empty output shape");
  OccIndex out_occ;
  ISF_TRY(occ_create(a, &out_occ, batch_size, out_shape[0], out_shape[1], out_shape[2], st));
  ISF_TRY(launch_mark_out(indices, num_in, spatial_shape_host, ksize_host, stride_host, padding_host,
                          out_occ, st));
  ISF_TRY(occ_scan(a, out_occ, st));
  int n_out = 0;
  ISF_TRY(read_int(out_occ.total, &n_out, st));
  ISF_REQUIRE(out_indices && n_out <= out_capacity, ISF_ERR_CAPACITY,
              "build_rulebook: %d outputs exceed out_capacity %d", n_out, out_capacity);
  ISF_REQUIRE(nbr_stride >= isf_nbr_stride(n_out), ISF_ERR_CAPACITY, "build_rulebook: nbr_stride too small");
  ISF_TRY(occ_compact_coords4(out_occ, out_indices, st));
  ISF_TRY(launch_nbr(a, out_indices, n_out, spatial_shape_host, ksize_host, stride_host, padding_host, false,
                     in_occ, perm, nbr, nbr_stride, nullptr, st));
  *num_out_host = n_out;
  return ISF_OK;
}

int isf_rulebook_to_indice_pairs(const int32_t* nbr, int nbr_stride, int num_out, int num_taps, int num_in,
                                 int32_t* indice_pairs, int32_t* indice_num, isf_stream_t stream) {
  ISF_REQUIRE(nbr && indice_pairs && indice_num && num_taps > 0 && num_in >= 0 && num_out >= 0, ISF_ERR_ARG,
              "rulebook_to_indice_pairs: bad arguments");
  if (num_in == 0) return ISF_OK;
  hipLaunchKernelGGL(isf::rb_to_pairs_kernel, dim3(num_taps), dim3(1024), 0, isf::as_stream(stream), nbr,
                     nbr_stride, num_out, num_in, indice_pairs, indice_num);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int isf_indice_pairs_to_rulebook(const int32_t* indice_pairs, const int32_t* indice_num, int num_taps,
                                 int num_in, int num_out, int32_t* nbr, int nbr_stride, isf_stream_t stream) {
  ISF_REQUIRE(indice_pairs && indice_num && nbr && num_taps > 0 && num_in >= 0 && num_out >= 0 &&
                  nbr_stride >= isf_nbr_stride(num_out),
              ISF_ERR_ARG, "indice_pairs_to_rulebook: bad arguments");
  hipStream_t st = isf::as_stream(stream);
  ISF_HIP_TRY(hipMemsetAsync(nbr, 0xff, (size_t)num_taps * nbr_stride * sizeof(int32_t), st));
  if (num_in == 0) return ISF_OK;
  hipLaunchKernelGGL(isf::rb_from_pairs_kernel, dim3(isf::ceil_div((long long)num_taps * num_in, 256)),
                     dim3(256), 0, st, indice_pairs, indice_num, num_taps, num_in, num_out, nbr, nbr_stride);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int isf_rulebook_to_lines(const int32_t* nbr, int nbr_stride, int num_taps, int taps_per_line, int num_out,
                          int32_t* lines, uint32_t* mask, int* not_consecutive_flag, isf_stream_t stream) {
  ISF_REQUIRE(nbr && lines && mask && nbr_stride > 0 && num_out >= 0 && num_out <= nbr_stride && num_taps >= 1 &&
                  num_taps <= 27 && (taps_per_line == 1 || taps_per_line == 3) && num_taps % taps_per_line == 0 &&
                  num_taps / taps_per_line <= 9,
              ISF_ERR_ARG, "rulebook_to_lines: bad arguments");
  hipLaunchKernelGGL(isf::rb_to_lines_kernel, dim3(isf::ceil_div(nbr_stride, 256)), dim3(256), 0, isf::as_stream(stream),
                     nbr, nbr_stride, num_taps, taps_per_line, num_out, lines, mask, not_consecutive_flag);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int isf_lines_to_rulebook(const int32_t* lines, const uint32_t* mask, int nbr_stride, int num_taps, int taps_per_line,
                          int32_t* nbr, isf_stream_t stream) {
  ISF_REQUIRE(nbr && lines && mask && nbr_stride > 0 && num_taps >= 1 && num_taps <= 27 &&
                  (taps_per_line == 1 || taps_per_line == 3) && num_taps % taps_per_line == 0,
              ISF_ERR_ARG, "lines_to_rulebook: bad arguments");
  hipLaunchKernelGGL(isf::rb_from_lines_kernel, dim3(isf::ceil_div(nbr_stride, 256)), dim3(256), 0, isf::as_stream(stream),
                     lines, mask, nbr_stride, num_taps, taps_per_line, nbr);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

}  // extern "C"
