// isf_voxelize.hip -- A1 dynamic voxelization, A2 deterministic hard (pillar) voxelization.
//
// A2 design (replaces the reference's O(P^2) point_to_voxelidx_kernel + single-thread
// determin_voxel_num, voxelization_cuda.cu:105-180):
//   1. every point bubbles its index into a per-cell sorted list of the max_points smallest point
//      indices (one atomicMin per slot; order-independent, deterministic result);
//   2. a cell's voxel id in "first appearance" order is the rank of its smallest point index among
//      all cells' smallest indices -> one bitmap over point indices + the popcount-prefix scan;
//   3. one gather pass writes voxels / coors / num_points for ids < max_voxels.
// All of it is HBM/L2-bound integer work: O(P * max_points) atomics worst case, O(P) typical.
#include <cstring>


#include "isf_common.h"

namespace isf {

__global__ void dynamic_voxelize_kernel(const float* __restrict__ points, int P, int C, VoxGeom g,
                                        int32_t* __restrict__ coors, int stride, int col0,
                                        int batch_idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  int cx, cy, cz;
  const bool ok = voxel_of_point(points + (size_t)i * C, g.vx, g.vy, g.vz, g.x0, g.y0, g.z0, g.gx, g.gy,
                                 g.gz, cx, cy, cz);
  int32_t* o = coors + (size_t)i * stride;
  if (col0) o[0] = batch_idx;
  o[col0 + 0] = ok ? cz : -1;
  o[col0 + 1] = ok ? cy : -1;
  o[col0 + 2] = ok ? cx : -1;
}

int dynamic_voxelize_impl(const float* points, int P, int C, const float vs[3], const float range[6],
                          int32_t* coors, int coors_stride, int coors_col0, int batch_idx,
                          hipStream_t st) {
  if (P <= 0) return ISF_OK;
  hipLaunchKernelGGL(dynamic_voxelize_kernel, dim3(ceil_div(P, 256)), dim3(256), 0, st, points, P, C,
                     make_geom(vs, range), coors, coors_stride, coors_col0, batch_idx);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

// ------------------------------------------------------------------------------------------- A2
static constexpr int kEmpty = 0x7f7f7f7f;  // hipMemset byte pattern 0x7f; larger than any point index

// The samples of a batch voxelized in ONE pass (isf_hard_voxelize_batched_device): the points are concatenated, sample b owns
// the point indices [off[b], off[b + 1]) and the grid planes [b * gz, (b + 1) * gz) of a grid stacked along z.  Point indices
// ascend with the sample, so "rank of a cell's first point" numbers sample 0's voxels first, then sample 1's ...
static constexpr int kHvMaxBatch = 16;
struct HvBatch {
  int n;
  int off[kHvMaxBatch + 1];
};
__device__ __forceinline__ int hv_sample_of(const HvBatch& hb, int i) {
  int b = 0;
#pragma unroll
  for (int k = 1; k < kHvMaxBatch; ++k) b += (k < hb.n && i >= hb.off[k]) ? 1 : 0;
  return b;
}

__global__ void hv_mark_cells_kernel(const float* __restrict__ points, int P, int C, VoxGeom g, HvBatch hb,
                                     unsigned long long* __restrict__ cbits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  int cx, cy, cz;
  if (!voxel_of_point(points + (size_t)i * C, g.vx, g.vy, g.vz, g.x0, g.y0, g.z0, g.gx, g.gy, g.gz, cx,
                      cy, cz))
    return;
  cz += hv_sample_of(hb, i) * g.gz;
  const unsigned long long cell = ((unsigned long long)cz * g.gy + cy) * g.gx + cx;
  const unsigned long long bit = 1ull << (cell & 63);
  unsigned long long* p = cbits + (cell >> 6);
  if (!(*p & bit)) atomicOr(p, bit);
}

// byte-map marking for small grids (pillars: 32 k cells): plain byte stores instead of device-scope atomicOr on a few
// hundred hot words (memory-side atomics serialise per address: 257 us for 300 k points), then one pack pass
__global__ void hv_mark_bytes_kernel(const float* __restrict__ points, int P, int C, VoxGeom g, HvBatch hb,
                                     unsigned char* __restrict__ seen) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  int cx, cy, cz;
  if (!voxel_of_point(points + (size_t)i * C, g.vx, g.vy, g.vz, g.x0, g.y0, g.z0, g.gx, g.gy, g.gz, cx,
                      cy, cz))
    return;
  cz += hv_sample_of(hb, i) * g.gz;
  seen[((size_t)cz * g.gy + cy) * g.gx + cx] = 1;
}

__global__ void hv_pack_bytes_kernel(const unsigned char* __restrict__ seen, size_t ncells, size_t nwords,
                                     unsigned long long* __restrict__ bits) {
  const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  unsigned long long v = 0;
  for (int j = 0; j < 64; ++j) {
    const size_t c = w * 64 + j;
    if (c < ncells && seen[c]) v |= 1ull << j;
  }
  bits[w] = v;
}

// points of a cell = one segment of the point indices STABLY sorted by the cell's rank (radix sort over the rank's bits,
// hv_stable_sort below): inside a segment the indices ascend, so a cell's T smallest point indices -- what the reference's
// sequential scan keeps (voxelization_cpu.cpp:54-69) -- are the first T of its segment.  Deterministic, no atomics.
// History: v1 atomicMin bubble insertion (720 us per 300 k points in pillars), v2 count -> scan -> fill -> select with
// one atomic per point per pass (90 + 16 + 90 + 55 us: the pillar grid's hot cells serialise the atomics).
__global__ void hv_keys_kernel(const float* __restrict__ points, int P, int C, VoxGeom g, HvBatch hb,
                               const unsigned long long* __restrict__ cbits, const uint32_t* __restrict__ cprefix,
                               uint32_t none, uint32_t* __restrict__ keys, int* __restrict__ idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  int cx, cy, cz;
  uint32_t key = none;   // points outside the grid sort behind every cell (none = row capacity > any rank)
  if (voxel_of_point(points + (size_t)i * C, g.vx, g.vy, g.vz, g.x0, g.y0, g.z0, g.gx, g.gy, g.gz, cx, cy, cz))
    key = (uint32_t)occ_lookup(cbits, cprefix,
                               ((unsigned long long)(cz + hv_sample_of(hb, i) * g.gz) * g.gy + cy) * g.gx + cx);
  keys[i] = key;
  if (idx) idx[i] = i;     // (the sort below takes the identity permutation implicitly)
}

// ---------------------------------------------------------------------------------------------------------------------
// The stable sort itself (round 6; hand-written, replaces rocprim::radix_sort_pairs -- 8 launches, 270 us per two-sample
// forward, one of them 214 us: a library tuned for arrays a thousand times larger).  LSD radix sort, `bits` (<= 10) key
// bits per pass, WAVE MULTI-SPLIT ranking: a wave owns a tile of kRsTile consecutive elements and walks it in order, 64 at
// a time; `bits` ballots tell every lane which lanes hold the same digit (mask &= my bit ? ballot : ~ballot), so its rank
// among them is a popcount and the group's first lane keeps the wave's running count of the digit in LDS -- no atomics,
// no sorting network, stable by construction (lanes, sub-tiles and tiles are all taken in index order).
//   count pass   per (digit, tile) counts                -> hist [digits][tiles]   (digit-major: the order of the output)
//   row scan     per digit: exclusive prefix over its tiles, and its total
//   scatter pass out[first position of the digit + hist[digit][tile] + running count in the tile + rank in the sub-tile]
constexpr int kRsTile = 1024;     // elements per wave (16 sub-tiles): 293 waves for 300 k points
constexpr int kRsWaves = 4;       // waves (tiles) per workgroup

template <bool SCATTER>
__global__ __launch_bounds__(64 * kRsWaves) void hv_radix_pass_kernel(const uint32_t* __restrict__ keys_in,
                                                                      const int* __restrict__ vals_in /* nullptr: identity */,
                                                                      int n, int shift, int bits, int tiles,
                                                                      uint32_t* __restrict__ hist,
                                                                      const uint32_t* __restrict__ totals /* [digits] */,
                                                                      uint32_t* __restrict__ keys_out, int* __restrict__ vals_out) {
  __shared__ uint32_t cnt_s[kRsWaves][1024];
  __shared__ uint32_t dbase_s[1024];                 // SCATTER: first output position of every digit
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x * kRsWaves + wave;
  const int nd = 1 << bits;
  if (SCATTER) {                                     // exclusive prefix of the digit totals (<= 1024 values: one wave)
    if (wave == 0) {
      uint32_t run = 0u;
      for (int b0 = 0; b0 < nd; b0 += 64) {
        const uint32_t v = totals[b0 + lane];
        uint32_t x = v;
#pragma unroll
        for (int dlt = 1; dlt < 64; dlt <<= 1) {
          const uint32_t o = __shfl_up(x, dlt, 64);
          if (lane >= dlt) x += o;
        }
        dbase_s[b0 + lane] = run + x - v;
        run += __shfl(x, 63, 64);
      }
    }
    __syncthreads();
  }
  if (tile >= tiles) return;
  volatile uint32_t* cnt = cnt_s[wave];
  for (int d = lane; d < nd; d += 64) cnt[d] = 0u;
  __builtin_amdgcn_wave_barrier();
  const unsigned long long lt = (1ull << lane) - 1ull;
  // the tile's keys (and values) are requested up front: kRsTile / 64 independent loads in flight per lane instead of one
  // exposed round trip per sub-tile (147 waves walking 32 dependent round trips each made a pass 50 us)
  constexpr int NSUB = kRsTile / 64;
  uint32_t kreg[NSUB];
  int vreg[NSUB];
#pragma unroll
  for (int sub = 0; sub < NSUB; ++sub) {
    const int i = tile * kRsTile + sub * 64 + lane;
    kreg[sub] = i < n ? keys_in[i] : 0u;
    vreg[sub] = (SCATTER && i < n) ? (vals_in ? vals_in[i] : i) : 0;
  }
#pragma unroll
  for (int sub = 0; sub < NSUB; ++sub) {
    const int i = tile * kRsTile + sub * 64 + lane;
    const bool valid = i < n;
    const uint32_t key = kreg[sub];
    const uint32_t d = (key >> shift) & (uint32_t)(nd - 1);
    unsigned long long m = __ballot(valid);
    for (int b = 0; b < bits; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    const int rank = __popcll(m & lt), group = __popcll(m);
    uint32_t base = 0u;
    if (valid) base = cnt[d];                       // every lane of a digit group reads the count before ...
    __builtin_amdgcn_wave_barrier();
    if (valid && rank == 0) cnt[d] = base + (uint32_t)group;   // ... its first lane advances it (one address per group)
    __builtin_amdgcn_wave_barrier();
    if (SCATTER && valid) {
      const uint32_t pos = dbase_s[d] + hist[(size_t)d * tiles + tile] + base + (uint32_t)rank;
      keys_out[pos] = key;
      vals_out[pos] = vreg[sub];
    }
  }
  if (!SCATTER)
    for (int d = lane; d < nd; d += 64) hist[(size_t)d * tiles + tile] = cnt[d];
}

// per digit (one wave each): exclusive prefix of its row of per-tile counts in place (coalesced, ceil(tiles / 64) rounds)
// and the digit's total; the scatter pass turns the totals into the digits' first output positions itself.  (A single
// workgroup scanning all digits x tiles entries with a chunk per thread was 100 us of dependent, uncoalesced loads.)
__global__ __launch_bounds__(256) void hv_radix_rowscan_kernel(uint32_t* __restrict__ hist, int nd, int tiles,
                                                               uint32_t* __restrict__ totals) {
  const int lane = threadIdx.x & 63, d = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= nd) return;
  uint32_t* row = hist + (size_t)d * tiles;
  uint32_t run = 0u;
  for (int b0 = 0; b0 < tiles; b0 += 64) {
    const int t = b0 + lane;
    const uint32_t v = t < tiles ? row[t] : 0u;
    uint32_t x = v;
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
      const uint32_t o = __shfl_up(x, dlt, 64);
      if (lane >= dlt) x += o;
    }
    if (t < tiles) row[t] = run + x - v;
    run += __shfl(x, 63, 64);
  }
  if (lane == 0) totals[d] = run;
}

// (keys, 0 .. n - 1) stably sorted by the low `key_bits` bits of the keys -> (keys_sorted, idx_sorted); keys_tmp / idx_tmp:
// scratch of n entries each, hist: (1 << 10) * tiles entries.  The sorted data always ends in keys_sorted / idx_sorted.
static int hv_stable_sort(const uint32_t* keys, uint32_t* keys_tmp, uint32_t* keys_sorted, int* idx_tmp, int* idx_sorted,
                          uint32_t* hist /* 1024 * (tiles + 1) */, int n, int key_bits, hipStream_t st) {
  const int passes = (key_bits + 9) / 10, bits = (key_bits + passes - 1) / passes;
  const int tiles = ceil_div(n, kRsTile), blocks = ceil_div(tiles, kRsWaves);
  const uint32_t* kin = keys;
  const int* vin = nullptr;
  for (int p = 0; p < passes; ++p) {
    // ping-pong so that the LAST pass writes the caller's output buffers
    const bool to_out = ((passes - 1 - p) & 1) == 0;
    uint32_t* kout = to_out ? keys_sorted : keys_tmp;
    int* vout = to_out ? idx_sorted : idx_tmp;
    uint32_t* totals = hist + (size_t)1024 * tiles;
    hipLaunchKernelGGL(hv_radix_pass_kernel<false>, dim3(blocks), dim3(64 * kRsWaves), 0, st, kin, vin, n, p * bits, bits, tiles,
                       hist, totals, kout, vout);
    hipLaunchKernelGGL(hv_radix_rowscan_kernel, dim3(ceil_div(1 << bits, 4)), dim3(256), 0, st, hist, 1 << bits, tiles, totals);
    hipLaunchKernelGGL(hv_radix_pass_kernel<true>, dim3(blocks), dim3(64 * kRsWaves), 0, st, kin, vin, n, p * bits, bits, tiles,
                       hist, totals, kout, vout);
    kin = kout;
    vin = vout;
  }
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

// the same sort for other callers (the row sort of the deep conv launches, isf_spconv16.hip): idx_sorted [n] = the stable
// order of the low `key_bits` bits of keys [n]; scratch from the arena
int stable_sort_u32_impl(Arena& a, const uint32_t* keys, int n, int key_bits, int* idx_sorted, hipStream_t st) {
  ISF_REQUIRE(keys && idx_sorted && n > 0 && key_bits >= 1 && key_bits <= 32, ISF_ERR_ARG, "stable_sort: bad arguments");
  uint32_t *keys_tmp = nullptr, *keys_sorted = nullptr, *hist = nullptr;
  int* idx_tmp = nullptr;
  ISF_TRY(a.alloc_n(&keys_tmp, (size_t)n));
  ISF_TRY(a.alloc_n(&keys_sorted, (size_t)n));
  ISF_TRY(a.alloc_n(&idx_tmp, (size_t)n));
  ISF_TRY(a.alloc_n(&hist, (size_t)1024 * (ceil_div(n, kRsTile) + 1)));
  return hv_stable_sort(keys, keys_tmp, keys_sorted, idx_tmp, idx_sorted, hist, n, key_bits, st);
}

// first sorted position of every cell
__global__ void hv_segment_heads_kernel(const uint32_t* __restrict__ keys, int P, uint32_t none,
                                        uint32_t* __restrict__ start) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= P) return;
  const uint32_t k = keys[j];
  if (k != none && (j == 0 || keys[j - 1] != k)) start[k] = (uint32_t)j;
}

__global__ void hv_segment_slots_kernel(const uint32_t* __restrict__ keys, const int* __restrict__ idx, int P,
                                        const uint32_t* __restrict__ start, uint32_t none, int T,
                                        int* __restrict__ slots /*[rows][T], pre-filled kEmpty*/) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= P) return;
  const uint32_t k = keys[j];
  if (k == none) return;
  const uint32_t t = (uint32_t)j - start[k];
  if (t < (uint32_t)T) slots[(size_t)k * T + t] = idx[j];
}

__global__ void hv_mark_first_kernel(const int* __restrict__ slots, const int* __restrict__ nrows,
                                     int T, unsigned long long* __restrict__ pbits) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= *nrows) return;
  const int first = slots[(size_t)r * T];
  atomicOr(pbits + (first >> 6), 1ull << (first & 63));
}

// number of set bits before position x of an occupancy index over point indices (x <= number of positions)
__device__ __forceinline__ int hv_rank_before(const unsigned long long* __restrict__ bits, const uint32_t* __restrict__ prefix,
                                              int x, int npos, const int* __restrict__ total) {
  if (x >= npos) return *total;
  const unsigned long long w = bits[x >> 6];
  return (int)prefix[x >> 6] + __popcll(w & ((1ull << (x & 63)) - 1ull));
}

// batched (hb.n > 1 or coors4 output): voxel ids are local to the sample (id among ALL first points - first points of the
// earlier samples), capped per sample, and the samples' voxels are written one behind the other (row = voxels kept by the
// earlier samples + local id) with (sample, z, y, x) coordinates -- what the reference's per-sample loop + cat + pad produce
__global__ void hv_gather_kernel(const float* __restrict__ points, int C, const int* __restrict__ slots,
                                 const int* __restrict__ nrows, const int32_t* __restrict__ cell_coors4,
                                 int T, const unsigned long long* __restrict__ pbits,
                                 const uint32_t* __restrict__ pprefix, int max_voxels,
                                 float* __restrict__ voxels, int32_t* __restrict__ coors,
                                 int32_t* __restrict__ num_points, int fill_empty, HvBatch hb, int gz, int batched,
                                 const int* __restrict__ ptotal) {
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= (long long)(*nrows) * T) return;
  const int r = (int)(tid / T), t = (int)(tid % T);
  const int* s = slots + (size_t)r * T;
  int vid = occ_lookup(pbits, pprefix, (unsigned long long)s[0]);
  int b = 0;
  if (batched) {
    b = hv_sample_of(hb, s[0]);
    const int npos = hb.off[hb.n];
    int row0 = 0, before = 0;   // voxels kept by the earlier samples; first points of the earlier samples
    for (int k = 0; k < b; ++k) {
      const int nxt = hv_rank_before(pbits, pprefix, hb.off[k + 1], npos, ptotal);
      const int cnt = nxt - before;
      row0 += cnt < max_voxels ? cnt : max_voxels;
      before = nxt;
    }
    vid -= before;
    if (vid < 0 || vid >= max_voxels) return;
    vid += row0;
  } else if (vid < 0 || vid >= max_voxels) {
    return;
  }
  const int pi = s[t];
  if (pi != kEmpty) {
    const float* p = points + (size_t)pi * C;
    float* o = voxels + ((size_t)vid * T + t) * C;
    for (int k = 0; k < C; ++k) o[k] = p[k];
  } else if (fill_empty) {                       // the caller did not zero the output: write the padding slots here
    float* o = voxels + ((size_t)vid * T + t) * C;
    for (int k = 0; k < C; ++k) o[k] = 0.f;
  }
  if (t == 0) {
    int n = 0;
    for (int k = 0; k < T; ++k) n += (s[k] != kEmpty);
    num_points[vid] = n;
    const int4 c = reinterpret_cast<const int4*>(cell_coors4)[r];  // (0, z, y, x); batched: z = sample * gz + z
    if (batched) {
      coors[(size_t)vid * 4 + 0] = b;
      coors[(size_t)vid * 4 + 1] = c.y - b * gz;
      coors[(size_t)vid * 4 + 2] = c.z;
      coors[(size_t)vid * 4 + 3] = c.w;
    } else {
      coors[(size_t)vid * 3 + 0] = c.y;
      coors[(size_t)vid * 3 + 1] = c.z;
      coors[(size_t)vid * 3 + 2] = c.w;
    }
  }
}

__global__ void hv_publish_count_kernel(const int* __restrict__ total, int max_voxels, int32_t* __restrict__ out) {
  *out = *total < max_voxels ? *total : max_voxels;
}

// batched: out[b] = voxels kept of sample b, out[n] = their sum (= rows written)
__global__ void hv_publish_counts_kernel(const unsigned long long* __restrict__ pbits, const uint32_t* __restrict__ pprefix,
                                         const int* __restrict__ total, HvBatch hb, int max_voxels,
                                         int32_t* __restrict__ out) {
  int before = 0, sum = 0;
  for (int b = 0; b < hb.n; ++b) {
    const int nxt = hv_rank_before(pbits, pprefix, hb.off[b + 1], hb.off[hb.n], total);
    const int cnt = nxt - before < max_voxels ? nxt - before : max_voxels;
    out[b] = cnt;
    sum += cnt;
    before = nxt;
  }
  out[hb.n] = sum;
}

// voxel_num_dev != nullptr: DEVICE-RESIDENT COUNT -- no host read-back (the caller sized the outputs for max_voxels and
// did not zero them: rows [0, *voxel_num_dev) are written completely, padding slots included; the rest is untouched)
// batch != nullptr (with voxel_num_dev): the samples of a batch in one pass -- `points` holds them one behind the other,
// coors is [rows, 4] = (sample, z, y, x), voxel_num_dev has n + 1 entries (per sample, then the sum); see HvBatch
int hard_voxelize_impl(Arena& a, const float* points, int P, int C, const float vs[3],
                       const float range[6], int max_points, int max_voxels, float* voxels,
                       int32_t* coors, int32_t* num_points, int* voxel_num_host, hipStream_t st,
                       int32_t* voxel_num_dev = nullptr, const HvBatch* batch = nullptr) {
  const VoxGeom g = make_geom(vs, range);
  ISF_REQUIRE(g.gx > 0 && g.gy > 0 && g.gz > 0, ISF_ERR_ARG, "hard_voxelize: empty grid");
  if (voxel_num_host) *voxel_num_host = 0;
  HvBatch hb;
  for (int k = 0; k <= kHvMaxBatch; ++k) hb.off[k] = P;
  hb.n = 1;
  hb.off[0] = 0;
  if (batch) hb = *batch;
  const int nb = hb.n;
  if (P <= 0) {
    if (voxel_num_dev) ISF_HIP_TRY(hipMemsetAsync(voxel_num_dev, 0, (batch ? nb + 1 : 1) * sizeof(int32_t), st));
    return ISF_OK;
  }
  const long long cells = (long long)g.gx * g.gy * g.gz * nb;
  const int row_cap = (int)(cells < P ? cells : P);
  OccIndex cocc;  // occupied cells of this sample
  OccIndex pocc;  // bitmap over POINT indices: rank of a cell's first point = its voxel id
  int* slots = nullptr;
  ISF_TRY(a.alloc_n(&slots, (size_t)row_cap * max_points));
  const bool bytes_path = cells <= (1ll << 22);
  if (bytes_path) {
    // the four fills of the pass -- cell index, byte map, point index (zero: incl. the padding words the scans read) and the
    // slot table (kEmpty) -- in one launch
    ISF_TRY(occ_create(a, &cocc, 1, g.gz * nb, g.gy, g.gx, st, false));
    ISF_TRY(occ_create(a, &pocc, 1, 1, 1, P, st, false));
    unsigned char* seen = nullptr;
    ISF_TRY(a.alloc_n(&seen, (size_t)cells));
    void* fp[4] = {cocc.bits, seen, pocc.bits, slots};
    const size_t fb[4] = {occ_bits_bytes(cocc), (size_t)cells, occ_bits_bytes(pocc), (size_t)row_cap * max_points * sizeof(int)};
    const unsigned char fv[4] = {0, 0, 0, 0x7f};
    ISF_TRY(fill_many(st, 4, fp, fb, fv));
    hipLaunchKernelGGL(hv_mark_bytes_kernel, dim3(ceil_div(P, 256)), dim3(256), 0, st, points, P, C, g, hb, seen);
    hipLaunchKernelGGL(hv_pack_bytes_kernel, dim3(ceil_div((long long)cocc.nwords, 256)), dim3(256), 0, st, seen,
                       (size_t)cells, cocc.nwords, cocc.bits);
  } else {
    ISF_TRY(occ_create(a, &cocc, 1, g.gz * nb, g.gy, g.gx, st));
    hipLaunchKernelGGL(hv_mark_cells_kernel, dim3(ceil_div(P, 256)), dim3(256), 0, st, points, P, C, g, hb,
                       cocc.bits);
  }
  ISF_LAUNCH_CHECK();
  ISF_TRY(occ_scan(a, cocc, st));
  int32_t* cell_coors = nullptr;
  ISF_TRY(a.alloc_n(&cell_coors, (size_t)row_cap * 4));
  if (!bytes_path) {
    ISF_HIP_TRY(hipMemsetAsync(slots, 0x7f, (size_t)row_cap * max_points * sizeof(int), st));
    ISF_TRY(occ_create(a, &pocc, 1, 1, 1, P, st));
  }
  ISF_TRY(occ_compact_coords4(cocc, cell_coors, st));
  {
    uint32_t *keys = nullptr, *keys_sorted = nullptr, *start = nullptr;
    int* idx_sorted = nullptr;
    ISF_TRY(a.alloc_n(&keys, (size_t)P));
    ISF_TRY(a.alloc_n(&keys_sorted, (size_t)P));
    ISF_TRY(a.alloc_n(&idx_sorted, (size_t)P));
    ISF_TRY(a.alloc_n(&start, (size_t)row_cap + 1));
    const uint32_t none = (uint32_t)row_cap;           // ranks are < row_cap
    int key_bits = 1;
    while ((1ull << key_bits) <= (unsigned long long)none) ++key_bits;
    hipLaunchKernelGGL(hv_keys_kernel, dim3(ceil_div(P, 256)), dim3(256), 0, st, points, P, C, g, hb, cocc.bits,
                       cocc.prefix, none, keys, (int*)nullptr);
    ISF_LAUNCH_CHECK();
    uint32_t* keys_tmp = nullptr;
    int* idx_tmp = nullptr;
    uint32_t* hist = nullptr;
    ISF_TRY(a.alloc_n(&keys_tmp, (size_t)P));
    ISF_TRY(a.alloc_n(&idx_tmp, (size_t)P));
    ISF_TRY(a.alloc_n(&hist, (size_t)1024 * (ceil_div(P, kRsTile) + 1)));
    ISF_TRY(hv_stable_sort(keys, keys_tmp, keys_sorted, idx_tmp, idx_sorted, hist, P, key_bits, st));
    hipLaunchKernelGGL(hv_segment_heads_kernel, dim3(ceil_div(P, 256)), dim3(256), 0, st, keys_sorted, P, none, start);
    hipLaunchKernelGGL(hv_segment_slots_kernel, dim3(ceil_div(P, 256)), dim3(256), 0, st, keys_sorted, idx_sorted, P,
                       start, none, max_points, slots);
  }
  hipLaunchKernelGGL(hv_mark_first_kernel, dim3(ceil_div(row_cap, 256)), dim3(256), 0, st, slots,
                     cocc.total, max_points, pocc.bits);
  ISF_LAUNCH_CHECK();
  ISF_TRY(occ_scan(a, pocc, st));
  hipLaunchKernelGGL(hv_gather_kernel, dim3(ceil_div((long long)row_cap * max_points, 256)), dim3(256),
                     0, st, points, C, slots, cocc.total, cell_coors, max_points, pocc.bits,
                     pocc.prefix, max_voxels, voxels, coors, num_points, voxel_num_dev ? 1 : 0, hb, g.gz, batch ? 1 : 0,
                     pocc.total);
  ISF_LAUNCH_CHECK();
  if (voxel_num_dev) {
    if (batch)
      hipLaunchKernelGGL(hv_publish_counts_kernel, dim3(1), dim3(1), 0, st, pocc.bits, pocc.prefix, pocc.total, hb,
                         max_voxels, voxel_num_dev);
    else
      hipLaunchKernelGGL(hv_publish_count_kernel, dim3(1), dim3(1), 0, st, pocc.total, max_voxels, voxel_num_dev);
    ISF_LAUNCH_CHECK();
    return ISF_OK;
  }
  int total = 0;
  ISF_TRY(read_int(pocc.total, &total, st));
  *voxel_num_host = total < max_voxels ? total : max_voxels;
  return ISF_OK;
}

}  // namespace isf

extern "C" {

int isf_dynamic_voxelize(const float* points, int num_points, int num_features,
                         const float voxel_size_host[3], const float coors_range_host[6],
                         int32_t* coors, isf_stream_t stream) {
  ISF_REQUIRE(num_points >= 0 && num_features >= 3 && voxel_size_host && coors_range_host,
              ISF_ERR_ARG, "dynamic_voxelize: bad arguments");
  ISF_REQUIRE(num_points == 0 || (points && coors), ISF_ERR_ARG, "dynamic_voxelize: null pointer");
  return isf::dynamic_voxelize_impl(points, num_points, num_features, voxel_size_host,
                                    coors_range_host, coors, 3, 0, 0, isf::as_stream(stream));
}

int isf_dynamic_voxelize_batched(const float* points, const int64_t* point_offsets_host,
                                 int batch_size, int num_features, const float voxel_size_host[3],
                                 const float coors_range_host[6], int32_t* coors4,
                                 isf_stream_t stream) {
  ISF_REQUIRE(point_offsets_host && batch_size >= 0 && num_features >= 3, ISF_ERR_ARG,
              "dynamic_voxelize_batched: bad arguments");
  for (int b = 0; b < batch_size; ++b) {
    const int64_t lo = point_offsets_host[b], hi = point_offsets_host[b + 1];
    ISF_REQUIRE(hi >= lo && hi - lo < (1ll << 31), ISF_ERR_ARG, "dynamic_voxelize_batched: bad offsets");
    ISF_TRY(isf::dynamic_voxelize_impl(points + lo * num_features, (int)(hi - lo), num_features,
                                       voxel_size_host, coors_range_host, coors4 + lo * 4, 4, 1, b,
                                       isf::as_stream(stream)));
  }
  return ISF_OK;
}

int isf_hard_voxelize(const float* points, int num_points, int num_features,
                      const float voxel_size_host[3], const float coors_range_host[6],
                      int max_points, int max_voxels, float* voxels, int32_t* coors,
                      int32_t* num_points_per_voxel, int* voxel_num_host, isf_stream_t stream) {
  ISF_REQUIRE(num_points >= 0 && num_features >= 3 && max_points > 0 && max_voxels > 0 &&
                  voxel_num_host && voxel_size_host && coors_range_host,
              ISF_ERR_ARG, "hard_voxelize: bad arguments");
  ISF_REQUIRE(num_points == 0 || (points && voxels && coors && num_points_per_voxel), ISF_ERR_ARG,
              "hard_voxelize: null pointer");
  isf::Arena& a = isf::arena_for_stream(isf::as_stream(stream));
  ISF_TRY(a.reset());
  return isf::hard_voxelize_impl(a, points, num_points, num_features, voxel_size_host,
                                 coors_range_host, max_points, max_voxels, voxels, coors,
                                 num_points_per_voxel, voxel_num_host, isf::as_stream(stream));
}

int isf_hard_voxelize_device(const float* points, int num_points, int num_features,
                             const float voxel_size_host[3], const float coors_range_host[6],
                             int max_points, int max_voxels, float* voxels, int32_t* coors,
                             int32_t* num_points_per_voxel, int32_t* voxel_num_device, isf_stream_t stream) {
  ISF_REQUIRE(num_points >= 0 && num_features >= 3 && max_points > 0 && max_voxels > 0 &&
                  voxel_num_device && voxel_size_host && coors_range_host,
              ISF_ERR_ARG, "hard_voxelize_device: bad arguments");
  ISF_REQUIRE(num_points == 0 || (points && voxels && coors && num_points_per_voxel), ISF_ERR_ARG,
              "hard_voxelize_device: null pointer");
  isf::Arena& a = isf::arena_for_stream(isf::as_stream(stream));
  ISF_TRY(a.reset());
  return isf::hard_voxelize_impl(a, points, num_points, num_features, voxel_size_host,
                                 coors_range_host, max_points, max_voxels, voxels, coors,
                                 num_points_per_voxel, nullptr, isf::as_stream(stream), voxel_num_device);
}

int isf_hard_voxelize_batched_device(const float* points, const int64_t* point_offsets_host, int batch_size,
                                     int num_features, const float voxel_size_host[3], const float coors_range_host[6],
                                     int max_points, int max_voxels, float* voxels, int32_t* coors4,
                                     int32_t* num_points_per_voxel, int32_t* voxel_num_device, isf_stream_t stream) {
  ISF_REQUIRE(point_offsets_host && batch_size >= 1 && batch_size <= isf::kHvMaxBatch && num_features >= 3 &&
                  max_points > 0 && max_voxels > 0 && voxel_num_device && voxel_size_host && coors_range_host,
              ISF_ERR_ARG, "hard_voxelize_batched_device: bad arguments (batch 1..%d)", isf::kHvMaxBatch);
  isf::HvBatch hb;
  hb.n = batch_size;
  ISF_REQUIRE(point_offsets_host[0] == 0, ISF_ERR_ARG, "hard_voxelize_batched_device: offsets start at 0");
  for (int b = 0; b <= isf::kHvMaxBatch; ++b) {
    const int64_t o = point_offsets_host[b < batch_size ? b : batch_size];
    ISF_REQUIRE(o >= 0 && o < (1ll << 31) && (b == 0 || b > batch_size || o >= point_offsets_host[b - 1]), ISF_ERR_ARG,
                "hard_voxelize_batched_device: bad offsets");
    hb.off[b] = (int)o;
  }
  const int P = hb.off[batch_size];
  ISF_REQUIRE(P == 0 || (points && voxels && coors4 && num_points_per_voxel), ISF_ERR_ARG,
              "hard_voxelize_batched_device: null pointer");
  isf::Arena& a = isf::arena_for_stream(isf::as_stream(stream));
  ISF_TRY(a.reset());
  return isf::hard_voxelize_impl(a, points, P, num_features, voxel_size_host, coors_range_host, max_points, max_voxels,
                                 voxels, coors4, num_points_per_voxel, nullptr, isf::as_stream(stream), voxel_num_device,
                                 &hb);
}

}  // extern "C"
