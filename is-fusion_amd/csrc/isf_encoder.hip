// isf_encoder.hip -- A7: dense BEV write-out, SparseEncoder.forward and the fused LiDAR branch.
#include "isf_common.h"

#include <algorithm>

namespace isf {

// ----------------------------------------------------------------------------------------- dense BEV
// out[b, c*D + z, y, x] = feats[row(b,z,y,x), c] or 0.  Dense-stationary: a workgroup owns 64 consecutive
// x cells of one (b, y) line, looks the <= 64*D rows up in the occupancy index, transposes 64x64 blocks
// through LDS and writes 256-B runs; every output element is written exactly once (no memset pass).
static constexpr int kDenseX = 64;

template <int FMT>   // 0 fp32 rows, 1 split rows, 2 f16 rows
__global__ __launch_bounds__(256) void dense_bev_kernel(const void* __restrict__ feats_v, int C, int D,
                                                        int H, int W,
                                                        const unsigned long long* __restrict__ bits,
                                                        const uint32_t* __restrict__ prefix,
                                                        const int32_t* __restrict__ perm,
                                                        float* __restrict__ out) {
  __shared__ float tile[64][kDenseX + 1];
  __shared__ int rows[kDenseX];
  const int xt = blockIdx.x, y = blockIdx.y, b = blockIdx.z;
  const int x0 = xt * kDenseX;
  const int t = threadIdx.x;
  for (int z = 0; z < D; ++z) {
    __syncthreads();
    if (t < kDenseX) {
      int r = -1;
      const int x = x0 + t;
      if (x < W) {
        r = occ_lookup(bits, prefix, (((unsigned long long)b * D + z) * H + y) * W + x);
        if (r >= 0 && perm) r = perm[r];
      }
      rows[t] = r;
    }
    __syncthreads();
    for (int c0 = 0; c0 < C; c0 += 64) {
      if (FMT != 0) {
        // split format: row = C/8 units of (hi8 | lo8) f16 (f16 rows: hi8 only); thread -> (row xx = t/8 + 32*j, unit t%8)
        const uint4* fs = reinterpret_cast<const uint4*>(feats_v);
        const int u = t & 7;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int xx = (t >> 3) + 32 * j;
          const int r = rows[xx];
          float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (r >= 0 && c0 + u * 8 < C) {
            const size_t o = FMT == 2 ? (size_t)r * (C >> 3) + (c0 >> 3) + u : split_hi_index((size_t)r, C >> 3, (c0 >> 3) + u);
            const uint4 hi = fs[o], lo = FMT == 2 ? make_uint4(0, 0, 0, 0) : fs[o + 4];
            const _Float16* h = reinterpret_cast<const _Float16*>(&hi);
            const _Float16* l = reinterpret_cast<const _Float16*>(&lo);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (float)h[q] + (float)l[q];
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) tile[u * 8 + q][xx] = v[q];
        }
      } else {
        // load: thread -> (row xx = t/16 + 16*j, 4 channels c0 + 4*(t%16))
        const float* feats = reinterpret_cast<const float*>(feats_v);
        const int c4 = (t & 15) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int xx = (t >> 4) + 16 * j;
          const int r = rows[xx];
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (r >= 0 && c0 + c4 < C) v = *reinterpret_cast<const float4*>(feats + (size_t)r * C + c0 + c4);
          tile[c4 + 0][xx] = v.x;
          tile[c4 + 1][xx] = v.y;
          tile[c4 + 2][xx] = v.z;
          tile[c4 + 3][xx] = v.w;
        }
      }
      __syncthreads();
      // store: lane -> x, wave -> channel stripe
      const int lane = t & 63, wv = t >> 6;
      if (x0 + lane < W) {
        for (int cc = wv; cc < 64 && c0 + cc < C; cc += 4) {
          const size_t ch = (size_t)(c0 + cc) * D + z;
          out[(((size_t)b * C * D + ch) * H + y) * W + x0 + lane] = tile[cc][lane];
        }
      }
      __syncthreads();
    }
  }
}

int sparse_to_dense_bev_impl(Arena& a, const void* feats, int fmt, const int32_t* indices, int n, int C,
                             int B, int D, int H, int W, float* out, const OccIndex* occ_in, hipStream_t st) {
  ISF_REQUIRE(C % (fmt ? 32 : 4) == 0, ISF_ERR_UNSUPPORTED, "dense: channels %d not a multiple of %d", C,
              fmt ? 32 : 4);
  OccIndex occ;
  const int32_t* perm = nullptr;
  if (occ_in) {
    occ = *occ_in;  // rows already in rank order
  } else {
    ISF_TRY(occ_create(a, &occ, B, D, H, W, st));
    ISF_TRY(occ_mark_coords4(occ, indices, n, st));
    ISF_TRY(occ_scan(a, occ, st));
    int32_t* p = nullptr;
    ISF_TRY(build_perm(a, occ, indices, n, &p, st));
    perm = p;
  }
  dim3 grid(ceil_div(W, kDenseX), H, B);
  if (fmt == 1)
    hipLaunchKernelGGL(dense_bev_kernel<1>, grid, dim3(256), 0, st, feats, C, D, H, W, occ.bits, occ.prefix,
                       perm, out);
  else if (fmt == 2)
    hipLaunchKernelGGL(dense_bev_kernel<2>, grid, dim3(256), 0, st, feats, C, D, H, W, occ.bits, occ.prefix,
                       perm, out);
  else
    hipLaunchKernelGGL(dense_bev_kernel<0>, grid, dim3(256), 0, st, feats, C, D, H, W, occ.bits, occ.prefix,
                       perm, out);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

// BEV map of the last level as SPLIT-FORMAT TOKEN MATRICES (isf_encoder_options.bev_format = 1): what the fusion encoder's
// 3 x 3 convolutions read (dense_conv.SplitMap) -- no fp32 [B, C*D, H, W] map, no NCHW -> split pass behind it.  Group g
// (256 channels) = matrix [B*H*W, 256] at out + g * B*H*W * 64 (uint4), token = (b*H + y)*W + x, channel = c*D + z
// (the reference's view(N, C*D, H, W), sparse_encoder.py:137-139).  Thread -> (token, 8-channel unit); every unit written.
__global__ __launch_bounds__(256) void bev_split_kernel(const uint4* __restrict__ fs, int C, int D, int H, int W,
                                                        const unsigned long long* __restrict__ bits,
                                                        const uint32_t* __restrict__ prefix, long long ntok,
                                                        uint4* __restrict__ out) {
  const int upt = C * D / 8;                                   // 8-channel units per token
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= ntok * upt) return;
  const long long tok = t / upt;
  const int j = (int)(t - tok * upt);
  const int hw = H * W;
  const int b = (int)(tok / hw), pos = (int)(tok - (long long)b * hw);
  const int y = pos / W, x = pos - y * W;
  _Float16 hi[8], lo[8];
  if (D == 2) {   // channels 8 j .. 8 j + 7 = (c, z) = (4 j, 0), (4 j, 1), (4 j + 1, 0), ...: four channels of two rows
    const int r0 = occ_lookup(bits, prefix, (((unsigned long long)b * 2 + 0) * H + y) * W + x);
    const int r1 = occ_lookup(bits, prefix, (((unsigned long long)b * 2 + 1) * H + y) * W + x);
    uint2 h0 = make_uint2(0, 0), l0 = h0, h1 = h0, l1 = h0;
    const int uc = j >> 1, e = j & 1;                          // the rows' 8-channel unit, its lower / upper four channels
    if (r0 >= 0) {
      const uint2* p = reinterpret_cast<const uint2*>(fs + split_hi_index((size_t)r0, C >> 3, uc));
      h0 = p[e];
      l0 = p[8 + e];                                           // lo piece: 4 uint4 = 8 uint2 further
    }
    if (r1 >= 0) {
      const uint2* p = reinterpret_cast<const uint2*>(fs + split_hi_index((size_t)r1, C >> 3, uc));
      h1 = p[e];
      l1 = p[8 + e];
    }
    const _Float16 *a0 = reinterpret_cast<const _Float16*>(&h0), *a1 = reinterpret_cast<const _Float16*>(&h1);
    const _Float16 *b0 = reinterpret_cast<const _Float16*>(&l0), *b1 = reinterpret_cast<const _Float16*>(&l1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      hi[2 * q] = a0[q]; hi[2 * q + 1] = a1[q];
      lo[2 * q] = b0[q]; lo[2 * q + 1] = b1[q];
    }
  } else {
    for (int q = 0; q < 8; ++q) {
      const int ch = 8 * j + q, c = ch / D, z = ch - c * D;
      const int r = occ_lookup(bits, prefix, (((unsigned long long)b * D + z) * H + y) * W + x);
      hi[q] = lo[q] = (_Float16)0.f;
      if (r >= 0) {
        const _Float16* p = reinterpret_cast<const _Float16*>(fs + split_hi_index((size_t)r, C >> 3, c >> 3));
        hi[q] = p[c & 7];
        lo[q] = p[32 + (c & 7)];                               // 4 uint4 = 32 halves further
      }
    }
  }
  const int g = j >> 5, ju = j & 31;                           // 256-channel group, unit inside it
  uint4* o = out + (size_t)g * ntok * 64 + split_hi_index((size_t)tok, 32, ju);
  o[0] = *reinterpret_cast<const uint4*>(hi);
  o[4] = *reinterpret_cast<const uint4*>(lo);
}

int sparse_to_bev_split_impl(const void* feats_split, int C, int B, int D, int H, int W, const OccIndex& occ, void* out,
                             hipStream_t st) {
  ISF_REQUIRE(C % 32 == 0 && (C * D) % 256 == 0, ISF_ERR_UNSUPPORTED,
              "bev (split token matrices): %d x %d channels are not whole 256-channel groups", C, D);
  const long long ntok = (long long)B * H * W, total = ntok * (C * D / 8);
  hipLaunchKernelGGL(bev_split_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, reinterpret_cast<const uint4*>(feats_split),
                     C, D, H, W, occ.bits, occ.prefix, ntok, reinterpret_cast<uint4*>(out));
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

__global__ void check_rank_order_kernel(const int32_t* __restrict__ coors4, int n, int B, int D, int H, int W,
                                        const unsigned long long* __restrict__ bits,
                                        const uint32_t* __restrict__ prefix, int* __restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 c = reinterpret_cast<const int4*>(coors4)[i];
  int r = -1;
  if (c.x >= 0 && c.y >= 0 && c.z >= 0 && c.w >= 0 && c.x < B && c.y < D && c.z < H && c.w < W)
    r = occ_lookup(bits, prefix, (((unsigned long long)c.x * D + c.y) * H + c.z) * W + c.w);
  if (r != i) *flag = 1;
}

__global__ void gather_rows_kernel(const float* __restrict__ x, const int32_t* __restrict__ perm, int n, int C,
                                   float* __restrict__ xs) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * C) return;
  xs[t] = x[(size_t)perm[t / C] * C + (t % C)];
}

// ----------------------------------------------------------------------------------------- encoder
struct StageTables {
  uint16_t* slots = nullptr;
  int32_t* ulist = nullptr;
  int32_t* ucount = nullptr;
};

struct LevelState {
  int shape[3];
  int n;                  // active rows (host)
  const int32_t* coors;   // [n,4] sorted
  OccIndex occ;
  bool has_occ;
  // SubM rulebook cache for this level (all SubM convs of one level share the active set)
  int cache_ks[3];
  int32_t* cache_nbr;
  int cache_stride;
  long long cache_pairs;
  StageTables cache_stage;   // staging tables of cache_nbr (slots == nullptr: not built)
  const int32_t* cache_order;   // tile order / table of cache_nbr for a (cache_order_cin -> cache_order_cout) launch, or nullptr
  int cache_order_cin, cache_order_cout;
  bool cache_order_is_table;
  const uint32_t* cache_lmask;  // non-null: cache_nbr is LINE-COMPRESSED (lines [ks0 * ks1][stride]) with these tap masks
  ConvCuPlan cache_cu;          // unit plan of cache_nbr for the one-workgroup-per-CU kernel (n_out == 0: not built)
  const int32_t* cache_rowmap;  // ROW SORT of cache_nbr (conv_row_sort_impl): position -> row, or nullptr
  const int32_t* cache_nbr_sorted;   // cache_nbr by position
  const uint32_t* cache_lmask_sorted;   // cache_lmask by position (line-compressed tables)
  int cache_sort_part_rows;     // the launch plan's rows per part the sort was cut for
};

// unit plan of one neighbour table (isf_spconv_cu.hip), built behind it on the geometry stream
static int build_cu_plan(Arena& a, const int32_t* nbr, int stride, int K, int n_out, ConvCuPlan* plan, hipStream_t sg,
                         int cap = 16) {
  int32_t* buf = nullptr;
  ISF_TRY(a.alloc_n(&buf, conv_cu_plan_ints(n_out)));
  return conv_cu_plan_impl(nbr, stride, K, n_out, buf, plan, sg, cap);
}

// tile order / tile table of one conv launch over a neighbour table, built behind the table on the geometry stream.  A
// launch of the tile kernel that is resident in one round gets a TILE TABLE (conv16_table_part: the groups dealt to the
// compute units by equal work; *is_table = true, run the conv with mode | 1024); the LDS-DMA kernel's launches keep the
// permutation of uniform tiles (conv16_tile_order_impl); *order stays nullptr when neither applies.
static int build_tile_order(Arena& a, const isf_conv_layer& ly, int K, const int32_t* nbr, int stride, int n_out,
                            int mode, bool dma, const int32_t** order, hipStream_t sg, const uint32_t* lmask = nullptr,
                            bool* is_table = nullptr, bool tables = true,
                            const int32_t* coors_out = nullptr /* [n_out][4] (b, z, y, x) of the output rows */, int band = 0) {
  *order = nullptr;
  if (is_table) *is_table = false;
  Conv16LaunchInfo info;
  if (dma)
    ISF_TRY(sparse_conv_forward_dma_impl(nullptr, ly.c_in, ly.packed16, K, ly.c_out, nbr, stride, n_out, nullptr,
                                         nullptr, nullptr, 0, nullptr, mode, sg, nullptr, &info));
  else
    ISF_TRY(sparse_conv_forward_f16x3_impl(nullptr, ly.c_in, ly.packed16, K, ly.c_out, nbr, stride, n_out, nullptr,
                                           nullptr, nullptr, 0, nullptr, mode, sg, nullptr, &info));
  if (!dma && is_table && tables && !lmask && conv16_table_applies(info) && n_out >= 16 * info.cus_per_xcd) {
    const int ng = ceil_div(n_out, 16);
    int32_t *masks = nullptr, *work = nullptr, *table = nullptr;
    ISF_TRY(a.alloc_n(&masks, (size_t)ng));
    ISF_TRY(a.alloc_n(&work, (size_t)ng));
    ISF_TRY(a.alloc_n(&table, (size_t)conv16_table_ints(info)));
    ISF_TRY(conv_group_masks_impl(nbr, stride, K, n_out, masks, work, sg));
    ISF_TRY(conv16_tile_table_impl(work, n_out, info, table, sg));
    *order = table;
    *is_table = true;
    return ISF_OK;
  }
  if (!conv16_order_applies(info)) {
    // a launch of several rounds: its tiles band by band in y (conv16_band_order_kernel), so that the rows the dz = +-1
    // taps gather are still in the XCD's L2
    if (band > 0 && coors_out && conv16_band_order_applies(info)) {
      int32_t* ord = nullptr;
      ISF_TRY(a.alloc_n(&ord, (size_t)conv16_order_parts(info) * conv16_order_tiles(info)));
      ISF_TRY(conv16_band_order_impl(coors_out, n_out, info, band, ord, sg));
      *order = ord;
    }
    return ISF_OK;
  }
  const size_t n = (size_t)conv16_order_parts(info) * conv16_order_tiles(info);
  int32_t *work = nullptr, *ord = nullptr;
  ISF_TRY(a.alloc_n(&work, n));
  ISF_TRY(a.alloc_n(&ord, n));
  ISF_TRY(conv16_tile_order_impl(nbr, stride, K, n_out, info, work, ord, sg, lmask));
  *order = ord;
  return ISF_OK;
}

// staging tables of one neighbour table (isf_spconv_stage.hip), built behind it on the geometry stream
static int build_stage_tables(Arena& a, const int32_t* nbr, int stride, int K, StageTables* t, hipStream_t sg) {
  const int units = stride / isf_stage_unit_rows();
  ISF_TRY(a.alloc_n(&t->slots, (size_t)K * stride));
  ISF_TRY(a.alloc_n(&t->ulist, (size_t)units * isf_stage_unit_cap()));
  ISF_TRY(a.alloc_n(&t->ucount, (size_t)units));
  return stage_tables_impl(nbr, stride, K, t->slots, t->ulist, t->ucount, sg);
}

// LDS rows per 128-row tile a layer stages by default (0 = the gather kernel): measured choice per channel shape
// (profiles/r03_*).
static int default_stage_rows(const isf_conv_layer& ly) {
  (void)ly;
  return 0;
}

static int ensure_occ(Arena& a, LevelState& L, int B, hipStream_t st) {
  if (L.has_occ) return ISF_OK;
  ISF_TRY(occ_create(a, &L.occ, B, L.shape[0], L.shape[1], L.shape[2], st));
  ISF_TRY(occ_mark_coords4(L.occ, L.coors, L.n, st));
  ISF_TRY(occ_scan(a, L.occ, st));
  L.has_occ = true;
  return ISF_OK;
}

int sparse_encoder_forward_impl(Arena& a, const float* x0, const int32_t* coors0, int n0, int B,
                                const int shape0[3], const OccIndex* occ0, const isf_conv_layer* layers,
                                int num_layers, float* spatial_features, int out_shape[4],
                                isf_encoder_stats* stats, int time_layers, const isf_encoder_options* opt,
                                hipStream_t st, hipEvent_t geometry_ready = nullptr,
                                const void* x0_split = nullptr /* x0 already in the split format (DynamicVFE wrote it) */) {
  const int precision = opt ? opt->precision : 0, diagnostic = opt ? opt->diagnostic : 0;
  const int dg = diagnostic & ~(32 | 64 | 128 | 256 | 512 | (15 << 10) | 16384 | 32768 | 65536 | 131072 | 262144 | 524288 | 1048576 | 2097152 | 4194304 | 8388608 | 16777216 | 33554432 | 67108864 | 134217728 | 268435456 | 536870912);   // 256: isf_lidar_branch_forward's VFE hand-over, not the encoder's   // bits 32 (uniform conv tiles) and 64 (tiles in launch order) combine with the others
  const bool tile_order = (diagnostic & 64) == 0;
  const bool dma_gather = (diagnostic & 128) == 0;   // bit 128: the narrow layers on the gather kernel as well
  const bool tile_tables = (diagnostic & 32768) != 0;      // bit 32768 (opt-in; measured slower): equal-work
                                                           // tile tables instead of uniform tiles + LPT order
  const bool line_tables = (diagnostic & 256 * 64) == 0;   // bit 16384: full neighbour tables for the narrow layers too
  const bool mailbox = (diagnostic & 131072) == 0;         // bit 131072: data-dependent counts through hipMemcpyAsync + sync
  // band order of the launches of several rounds (conv16_band_order_kernel): OPT-IN with bit 16777216 -- measured slower on
  // the benchmark's scenes (one dominant ground plane per frame: 64 -> 32 0.135 -> 0.154, 32 -> 32 0.27 -> 0.29, 64 -> 64
  // 0.694 -> 0.709 ms per step, profiles/r06_band_order.txt): row order already keeps a plane's y-neighbours together
  const bool band_order = (diagnostic & 16777216) != 0;
  auto band_of = [&](const int shape[3]) -> int { return band_order ? std::max(8, shape[1] / 45) : 0; };
  // row sort of the deep SubM launches (conv_row_sort_impl): rows of a tile / a 16-row group with the same tap mask;
  // bit 33554432 switches it off (A/B; results are bit-identical either way)
  const bool row_sort = (diagnostic & 33554432) == 0;
  // ... of the NARROW layers too (LDS-DMA kernel, line-compressed tables): opt-in with bit 67108864 -- 33 % fewer tile-taps at
  // level 0 on paper, no gain measured (64 -> 64 0.703 -> 0.719, 32 -> 32 0.272 -> 0.276 ms per step: those layers are bound
  // by their gathers, and like rows are further apart), profiles/r06_row_sort.txt
  const bool narrow_sort = (diagnostic & 67108864) != 0;
  // CHUNK SPLIT of the 256-column layers (conv mode 524288, isf_spconv16.hip: two workgroups per tile, each over half of
  // the 32-channel chunks, the second to arrive adds the other's accumulators and runs the epilogue): OPT-IN with bit
  // 536870912.  Built on the reading that the small deep levels are ONE round of workgroups that ends with its longest tile
  // (216 steps against an average of 136) -- and measured slower: 256 -> 256 0.223 -> 0.256 ms per launch, 1 030 -> 980
  // frames/s (gpurun_out / profiles/EXPERIMENTS.md): the launch is bound by what its workgroups move through the CUs'
  // vector-memory paths in total, not by its longest chain; a second prologue per tile and the exchange add to that.
  // Deterministic (a + b == b + a), not the bits of the unsplit kernel (the sum over chunks is (lower half) + (upper half)).
  const bool ksplit_on = (diagnostic & 536870912) != 0;
  constexpr int sort_min_rows = 4096;   // (configs[2] at B = 2: 7.05 ms sorted from 4 096 rows up, 7.10 from 32 768, 7.10 unsorted)
  // key of the sort: 1 = six coarse bits (one radix pass) everywhere.  Sixteen taps of the planes above / below (key 3,
  // two passes) made the 256-column launches 0.8 % faster (1.131 vs 1.140 ms per step) and their fabric traffic 37 % larger
  // (FETCH_SIZE 226 -> 311 MB per launch: finer key groups scatter a tile's rows over the grid) -- not kept;
  // bit 134217728: coarse | in-plane taps (key 2; A/B)
  const bool sort_key_ab = (diagnostic & 134217728) != 0;
  const bool cu_units = (diagnostic & 512) != 0;     // bit 512 (opt-in; measured slower): the 256-column
                                                     // layers on the one-workgroup-per-CU kernel
  const int cu_cap = conv_cu_variant_cap((diagnostic >> 10) & 15);   // unit shape of the requested kernel variant
  ISF_REQUIRE(precision >= 0 && precision <= 2 && diagnostic >= 0 &&
                  (dg == 0 || dg == 2 || dg == 4 || dg == 6 || dg == 8 || dg == 16) && !(precision == 2 && dg != 0),
              ISF_ERR_ARG, "sparse_encoder: options (precision %d, diagnostic %d)", precision, diagnostic);
  // precision 2: f16 storage + single-pass f16 arithmetic (mode 257 of the conv kernel)
  const int conv_mode = precision == 2 ? (257 | (diagnostic & 32)) : (diagnostic & ~(64 | 128 | 256 | 512 | (15 << 10) | 16384 | 32768 | 65536 | 131072 | 262144 | 524288 | 1048576 | 2097152 | 4194304 | 8388608 | 16777216 | 33554432 | 67108864 | 134217728 | 268435456 | 536870912));
  // bits 262144 / 524288: the 256-column layers as one column block (conv mode 4096 / 8192; isf_spconv16.hip)
  const int wide_cols = (diagnostic & 262144 ? 4096 : 0) | (diagnostic & 524288 ? 8192 : 0);
  const int stagger = ((diagnostic & 1048576) ? 65536 : 0) | ((diagnostic & 2097152) ? 131072 : 0) | ((diagnostic & 4194304) ? 262144 : 0) |
                      ((diagnostic & 8388608) ? 32768 : 0) | ((diagnostic & 268435456) ? 64 : 0);   // bit 8388608: the deep layers on isf_spconv_deep.hip (opt-in: LDS-DMA gathers + one instruction stream per step; bit-identical, 8 % slower)   // bit 4194304: gathered rows two steps ahead (A2 loop); bit 2097152: round 4's issue phase in the deep layers (A/B)
    // bit 1048576: staggered issue phases in the deep layers' workgroups
  const bool f16io = precision == 2;
  const int stage_opt = opt ? opt->stage_rows : 0;
  const unsigned stage_mask = opt ? (unsigned)opt->stage_mask : 0u;
  ISF_REQUIRE(stage_opt >= -1, ISF_ERR_ARG, "sparse_encoder: stage_rows %d", stage_opt);
  ISF_REQUIRE(num_layers > 0 && num_layers <= 32, ISF_ERR_ARG, "sparse_encoder: %d layers (1..32)", num_layers);
  // Geometry (occupancy indexes, output sets, neighbour tables: small integer kernels + the host syncs that size
  // the next level) runs on a side stream and overlaps the convolutions of the previous level on `st`; a
  // convolution waits for the event recorded behind its level's table.  `sg` first waits for everything the
  // caller enqueued on `st` -- or only for `geometry_ready`, the point where the VFE's coordinates are final.
  hipStream_t sg = nullptr;
  ISF_TRY(side_stream(a, &sg));
  if (geometry_ready) ISF_HIP_TRY(hipStreamWaitEvent(sg, geometry_ready, 0));   // coords final: overlap the VFE tail too
  else ISF_TRY(stream_wait_stream(a, sg, st));
  LevelState L;
  for (int j = 0; j < 3; ++j) L.shape[j] = shape0[j];
  L.n = n0;
  L.coors = coors0;
  L.has_occ = false;
  if (occ0) { L.occ = *occ0; L.has_occ = true; }
  L.cache_nbr = nullptr;
  L.cache_order = nullptr;
  L.cache_order_cin = L.cache_order_cout = 0;
  L.cache_order_is_table = false;
  L.cache_cu = ConvCuPlan();
  L.cache_lmask = nullptr;
  L.cache_rowmap = L.cache_nbr_sorted = nullptr;
  L.cache_lmask_sorted = nullptr;
  L.cache_sort_part_rows = 0;
  // precision: f16x3 split MFMA when every layer was packed for it (and not overridden), else fp32 MFMA
  bool use16 = precision != 1;
  for (int i = 0; i < num_layers; ++i)
    use16 = use16 && layers[i].packed16 && sparse_conv_f16x3_supported(layers[i].c_in, layers[i].c_out);
  const void* outputs[32];
  int out_rows[32];
  const void* x = x0;
  const void* x_in0 = x0;
  ISF_REQUIRE(!x0_split || (use16 && !f16io), ISF_ERR_ARG, "sparse_encoder: split input without the split kernels");
  if (x0_split) {
    x = x0_split;
    x_in0 = x0_split;
  } else if (use16) {  // inter-layer activations live in the split (hi8|lo8 f16) format: same bytes as fp32
    void* xs0 = nullptr;
    ISF_TRY(a.alloc(&xs0, (size_t)std::max(n0, 1) * layers[0].c_in * 4));
    if (f16io) ISF_TRY(f32_to_half_impl(x0, (size_t)n0 * layers[0].c_in, xs0, st));
    else ISF_TRY(f32_to_split_impl(x0, (size_t)n0 * layers[0].c_in, xs0, st));
    x = xs0;
    x_in0 = xs0;
  }
  unsigned long long* pair_counts = nullptr;   // per-rulebook pair totals: statistics only (8 + 1 launches a caller without
  if (stats) {                                   // `stats` does not pay for)
    ISF_TRY(a.alloc_n(&pair_counts, 32));
    ISF_HIP_TRY(hipMemsetAsync(pair_counts, 0, 32 * sizeof(unsigned long long), sg));
  }
  std::vector<hipEvent_t> ev;
  if (time_layers && stats) {
    ev.resize((size_t)num_layers * 2);
    for (auto& e : ev) ISF_HIP_TRY(hipEventCreate(&e));
  }
  int c_last = layers[0].c_in;
  for (int i = 0; i < num_layers; ++i) {
    const isf_conv_layer& ly = layers[i];
    const int K = ly.ksize[0] * ly.ksize[1] * ly.ksize[2];
    ISF_REQUIRE(K >= 1 && K <= 27, ISF_ERR_UNSUPPORTED, "sparse_encoder: layer %d has %d taps", i, K);
    ISF_REQUIRE(ly.c_in == c_last, ISF_ERR_ARG, "sparse_encoder: layer %d expects %d channels, got %d", i, ly.c_in, c_last);
    int32_t* nbr = nullptr;
    int stride = 0;
    int n_out = L.n;
    int n_in = L.n;
    // LDS rows this layer stages (0 = gather kernel); the timing diagnostics exist on the gather kernel only
    int srows = 0;
    if (use16 && dg == 0 && !f16io && stage_opt >= 0 && sparse_conv_f16x3_supported(ly.c_in, ly.c_out))
      srows = stage_opt == 0 ? default_stage_rows(ly)
                             : ((stage_mask == 0 || ((stage_mask >> i) & 1u)) ? stage_opt : 0);
    StageTables stg;
    const int32_t* order = nullptr;
    bool order_is_table = false;
    // 256-column layers: one workgroup per CU over units of equal work (isf_spconv_cu.hip; fp32-class mode only)
    const bool cu = use16 && cu_units && srows == 0 && conv_mode == 0 && sparse_conv_cu_supported(ly.c_in, ly.c_out);
    ConvCuPlan cu_plan;
    const bool want_order = use16 && tile_order && srows == 0 && !cu;
    const int ksbit = (ksplit_on && use16 && !f16io && !cu && srows == 0 && dg == 0 && ly.c_out == 256 && ly.c_in >= 128 &&
                       conv_mode == (conv_mode & 32) && wide_cols == 0 && stagger == 0) ? 524288 : 0;
    // narrow layers: LDS-DMA gathers (isf_spconv_dma.hip); the timing diagnostics and mode 16 exist on the gather kernel
    const bool dma = use16 && dma_gather && srows == 0 && dg == 0 && sparse_conv_dma_supported(ly.c_in, ly.c_out);
    const uint32_t* lmask = nullptr;   // non-null: `nbr` is the line-compressed table of this layer
    const int nx = ly.ksize[2];
    // a rulebook may be line-compressed when every layer that reads it runs the LDS-DMA kernel: for a SubM table, all the
    // SubM layers of this level with the same kernel (up to the next strided conv); for a strided conv, itself
    auto lines_ok = [&](int first) -> bool {
      if (!use16 || !dma_gather || !line_tables || dg != 0 || stage_opt > 0 || (nx != 1 && nx != 3) ||
          ly.ksize[0] * ly.ksize[1] > 9)
        return false;
      for (int j = first; j < num_layers; ++j) {
        const isf_conv_layer& q = layers[j];
        if (q.conv_type != ISF_CONV_SUBM) break;
        if (q.ksize[0] != ly.ksize[0] || q.ksize[1] != ly.ksize[1] || q.ksize[2] != ly.ksize[2]) continue;
        if (!sparse_conv_dma_supported(q.c_in, q.c_out)) return false;
      }
      return true;
    };
    // ROW SORT (round 6): a SubM layer computes its rows in the order of their tap masks (deep layers on the tile kernel,
    // narrow layers on the LDS-DMA kernel, full or line-compressed table)
    // (the f16-storage mode too, round 6: the sort only renames positions; its epilogue reads / writes through the row map)
    const bool sort_mode_ok = conv_mode == (conv_mode & 32) || conv_mode == (257 | (conv_mode & 32));
    const bool sort_ok = row_sort && use16 && !cu && srows == 0 && ly.conv_type == ISF_CONV_SUBM && K == 27 &&
                         sort_mode_ok && wide_cols == 0 && (stagger & ~64) == 0 && L.n >= sort_min_rows &&
                         (dma ? narrow_sort : ly.c_out >= 128);
    const int32_t* rowmap = nullptr;
    auto launch_info = [&](const int32_t* table, int tstride, int rows, Conv16LaunchInfo* info) -> int {
      if (dma)
        return sparse_conv_forward_dma_impl(nullptr, ly.c_in, ly.packed16, K, ly.c_out, table, tstride, rows, nullptr, nullptr,
                                            nullptr, 0, nullptr, conv_mode, sg, nullptr, info);
      return sparse_conv_forward_f16x3_impl(nullptr, ly.c_in, ly.packed16, K, ly.c_out, table, tstride, rows, nullptr, nullptr,
                                            nullptr, 0, nullptr, conv_mode | ksbit | (ly.c_out >= 128 ? stagger : 0), sg, nullptr, info);
    };
    auto ensure_row_sort = [&](const int32_t* table, int tstride, int rows, bool* built) -> int {
      *built = false;
      if (!sort_ok || L.cache_rowmap) return ISF_OK;   // (sorted for another plan: this layer keeps the plain table)
      Conv16LaunchInfo info;
      ISF_TRY(launch_info(table, tstride, rows, &info));
      int32_t *rm = nullptr, *ns = nullptr;
      ISF_TRY(a.alloc_n(&rm, (size_t)tstride));
      if (L.cache_lmask) {
        const int nl = ly.ksize[0] * ly.ksize[1];
        uint32_t* lms = nullptr;
        ISF_TRY(a.alloc_n(&ns, (size_t)nl * tstride));
        ISF_TRY(a.alloc_n(&lms, (size_t)tstride));
        // the input level's rows have few, varied neighbours (6 of 27 on the benchmark geometry): all 27 bits decide
        ISF_TRY(conv_row_sort_lines_impl(a, table, L.cache_lmask, tstride, nl, rows, info.part_rows, L.coors == coors0, rm, ns,
                                         lms, sg));
        L.cache_lmask_sorted = lms;
      } else {
        ISF_TRY(a.alloc_n(&ns, (size_t)K * tstride));
        ISF_TRY(conv_row_sort_impl(a, table, tstride, K, rows, info.part_rows, rm, ns, sg, sort_key_ab ? 2 : 1));
      }
      L.cache_rowmap = rm;
      L.cache_nbr_sorted = ns;
      L.cache_sort_part_rows = info.part_rows;
      *built = true;
      return ISF_OK;
    };
    auto sort_applies = [&](const int32_t* table, int tstride, int rows) -> bool {   // this layer's plan == the sort's plan
      if (!sort_ok || !L.cache_rowmap) return false;
      Conv16LaunchInfo info;
      if (launch_info(table, tstride, rows, &info) != ISF_OK) return false;
      return info.part_rows == L.cache_sort_part_rows;
    };
    if (ly.conv_type == ISF_CONV_SUBM) {
      const bool hit = L.cache_nbr && L.cache_ks[0] == ly.ksize[0] && L.cache_ks[1] == ly.ksize[1] &&
                       L.cache_ks[2] == ly.ksize[2];
      if (!hit) {
        ISF_TRY(ensure_occ(a, L, B, sg));
        stride = isf_nbr_stride(L.n);
        L.cache_lmask = nullptr;
        if (sparse_conv_dma_supported(ly.c_in, ly.c_out) && lines_ok(i)) {
          const int nl = ly.ksize[0] * ly.ksize[1];
          uint32_t* lm = nullptr;
          ISF_TRY(a.alloc_n(&nbr, (size_t)nl * stride));
          ISF_TRY(a.alloc_n(&lm, (size_t)stride));
          ISF_TRY(launch_nbr_lines(a, L.coors, L.n, L.shape, ly.ksize, ly.stride, ly.padding, true, L.occ, nbr, lm, stride,
                                   stats ? pair_counts + i : nullptr, sg));
          L.cache_lmask = lm;
        } else {
          ISF_TRY(a.alloc_n(&nbr, (size_t)K * stride));
          ISF_TRY(launch_nbr(a, L.coors, L.n, L.shape, ly.ksize, ly.stride, ly.padding, true, L.occ, nullptr, nbr,
                             stride, stats ? pair_counts + i : nullptr, sg));
        }
        L.cache_nbr = nbr;
        L.cache_stride = stride;
        for (int j = 0; j < 3; ++j) L.cache_ks[j] = ly.ksize[j];
        L.cache_pairs = -(long long)i - 1;  // pairs live in pair_counts[i]; resolved after the final sync
        L.cache_stage = StageTables();
        if (srows > 0) ISF_TRY(build_stage_tables(a, nbr, stride, K, &L.cache_stage, sg));
        L.cache_order = nullptr;
        L.cache_order_cin = L.cache_order_cout = 0;
        L.cache_order_is_table = false;
        L.cache_cu = ConvCuPlan();
        L.cache_rowmap = L.cache_nbr_sorted = nullptr;
        L.cache_lmask_sorted = nullptr;
        L.cache_sort_part_rows = 0;
        {
          bool built = false;
          ISF_TRY(ensure_row_sort(nbr, stride, n_out, &built));
        }
        if (want_order) {
          ISF_TRY(build_tile_order(a, ly, K, sort_applies(nbr, stride, n_out) ? L.cache_nbr_sorted : nbr, stride, n_out,
                                   conv_mode | ksbit | (ly.c_out >= 128 && conv_mode == (conv_mode & 32) ? stagger : 0), dma, &L.cache_order, sg,
                                   (L.cache_lmask && sort_applies(nbr, stride, n_out)) ? L.cache_lmask_sorted : L.cache_lmask,
                                   &L.cache_order_is_table, tile_tables, L.coors, band_of(L.shape)));
          L.cache_order_cin = ly.c_in;
          L.cache_order_cout = ly.c_out;
        }
        if (cu) ISF_TRY(build_cu_plan(a, nbr, stride, K, n_out, &L.cache_cu, sg, cu_cap));
        ISF_TRY(stream_wait_stream(a, st, sg));   // this level's convolutions wait for its table
      } else {
        nbr = L.cache_nbr;
        stride = L.cache_stride;
        if (srows > 0 && !L.cache_stage.slots) {   // an earlier layer of the level ran unstaged: build the tables now
          ISF_TRY(build_stage_tables(a, nbr, stride, K, &L.cache_stage, sg));
          ISF_TRY(stream_wait_stream(a, st, sg));
        }
        {   // an earlier layer of the level did not sort (another kernel): sort now
          bool built = false;
          ISF_TRY(ensure_row_sort(nbr, stride, n_out, &built));
          if (built) ISF_TRY(stream_wait_stream(a, st, sg));
        }
        if (want_order && (L.cache_order_cin != ly.c_in || L.cache_order_cout != ly.c_out)) {   // another launch shape
          ISF_TRY(build_tile_order(a, ly, K, sort_applies(nbr, stride, n_out) ? L.cache_nbr_sorted : nbr, stride, n_out,
                                   conv_mode | ksbit | (ly.c_out >= 128 && conv_mode == (conv_mode & 32) ? stagger : 0), dma, &L.cache_order, sg,
                                   (L.cache_lmask && sort_applies(nbr, stride, n_out)) ? L.cache_lmask_sorted : L.cache_lmask,
                                   &L.cache_order_is_table, tile_tables, L.coors, band_of(L.shape)));
          L.cache_order_cin = ly.c_in;
          L.cache_order_cout = ly.c_out;
          ISF_TRY(stream_wait_stream(a, st, sg));
        }
        if (cu && L.cache_cu.n_out == 0) {   // an earlier layer of the level did not need the unit plan
          ISF_TRY(build_cu_plan(a, nbr, stride, K, n_out, &L.cache_cu, sg, cu_cap));
          ISF_TRY(stream_wait_stream(a, st, sg));
        }
      }
      cu_plan = L.cache_cu;
      lmask = L.cache_lmask;
      if (sort_applies(nbr, stride, n_out)) {   // positions instead of rows: the sorted table + the row map
        nbr = const_cast<int32_t*>(L.cache_nbr_sorted);
        if (lmask) lmask = L.cache_lmask_sorted;
        rowmap = L.cache_rowmap;
      }
      ISF_REQUIRE(!lmask || dma, ISF_ERR_UNSUPPORTED, "sparse_encoder: layer %d cannot read a line-compressed table", i);
      stg = L.cache_stage;
      if (want_order) {
        order = L.cache_order;
        order_is_table = L.cache_order_is_table;
      }
      if (stats) stats->pairs[i] = hit ? L.cache_pairs : -(long long)i - 1;
    } else {
      ISF_TRY(ensure_occ(a, L, B, sg));
      LevelState Nx;
      ISF_TRY(isf_conv_out_shape(L.shape, ly.ksize, ly.stride, ly.padding, Nx.shape));
      ISF_REQUIRE(Nx.shape[0] > 0 && Nx.shape[1] > 0 && Nx.shape[2] > 0, ISF_ERR_ARG,
                  "sparse_encoder: layer %d output shape empty", i);
      ISF_TRY(occ_create(a, &Nx.occ, B, Nx.shape[0], Nx.shape[1], Nx.shape[2], sg));
      ISF_TRY(launch_mark_out(L.coors, L.n, L.shape, ly.ksize, ly.stride, ly.padding, Nx.occ, sg));
      ISF_TRY(occ_scan(a, Nx.occ, sg));
      if (mailbox) {                               // host wait without a copy command (post_int / wait_int)
        unsigned ticket = 0;
        ISF_TRY(post_int(a, Nx.occ.total, sg, &ticket));
        ISF_TRY(wait_int(a, ticket, sg, &Nx.n));
      } else {
        ISF_TRY(read_int(Nx.occ.total, &Nx.n, sg));  // host sync: sizes the next level's buffers / grids
      }
      Nx.has_occ = true;
      int32_t* nc = nullptr;
      ISF_TRY(a.alloc_n(&nc, (size_t)std::max(Nx.n, 1) * 4));
      if (Nx.n > 0) ISF_TRY(occ_compact_coords4(Nx.occ, nc, sg));
      Nx.coors = nc;
      stride = isf_nbr_stride(Nx.n);
      if (dma && lines_ok(num_layers)) {     // a strided conv's table has one reader
        uint32_t* lm = nullptr;
        ISF_TRY(a.alloc_n(&nbr, (size_t)ly.ksize[0] * ly.ksize[1] * stride));
        ISF_TRY(a.alloc_n(&lm, (size_t)stride));
        ISF_TRY(launch_nbr_lines(a, Nx.coors, Nx.n, L.shape, ly.ksize, ly.stride, ly.padding, false, L.occ, nbr, lm, stride,
                                 stats ? pair_counts + i : nullptr, sg));
        lmask = lm;
      } else {
        ISF_TRY(a.alloc_n(&nbr, (size_t)K * stride));
        ISF_TRY(launch_nbr(a, Nx.coors, Nx.n, L.shape, ly.ksize, ly.stride, ly.padding, false, L.occ, nullptr, nbr,
                           stride, stats ? pair_counts + i : nullptr, sg));
      }
      if (srows > 0) ISF_TRY(build_stage_tables(a, nbr, stride, K, &stg, sg));
      // ROW SORT of a deep strided convolution's own table (its rows want 10 of 27 taps, in many patterns: -12 % tile-taps,
      // -26 % MFMA blocks at level 3 on the benchmark geometry, profiles/r06_row_sort.txt)
      if (row_sort && use16 && !dma && !cu && srows == 0 && !lmask && ly.c_out >= 128 && K == 27 &&
          sort_mode_ok && wide_cols == 0 && (stagger & ~64) == 0 && Nx.n >= sort_min_rows) {
        Conv16LaunchInfo info;
        ISF_TRY(sparse_conv_forward_f16x3_impl(nullptr, ly.c_in, ly.packed16, K, ly.c_out, nbr, stride, Nx.n, nullptr, nullptr,
                                               nullptr, 0, nullptr, conv_mode | ksbit | stagger, sg, nullptr, &info));
        int32_t *rm = nullptr, *ns = nullptr;
        ISF_TRY(a.alloc_n(&rm, (size_t)stride));
        ISF_TRY(a.alloc_n(&ns, (size_t)K * stride));
        ISF_TRY(conv_row_sort_impl(a, nbr, stride, K, Nx.n, info.part_rows, rm, ns, sg, 1));
        nbr = ns;
        rowmap = rm;
      }
      if (want_order)
        ISF_TRY(build_tile_order(a, ly, K, nbr, stride, Nx.n, conv_mode | ksbit | (ly.c_out >= 128 && conv_mode == (conv_mode & 32) ? stagger : 0), dma, &order, sg, lmask, &order_is_table, tile_tables,
                                 Nx.coors, band_of(Nx.shape)));
      if (cu && Nx.n > 0) ISF_TRY(build_cu_plan(a, nbr, stride, K, Nx.n, &cu_plan, sg, cu_cap));
      if (stats) stats->pairs[i] = -(long long)i - 1;
      Nx.cache_cu = ConvCuPlan();
      Nx.cache_lmask = nullptr;
      Nx.cache_rowmap = Nx.cache_nbr_sorted = nullptr;
      Nx.cache_lmask_sorted = nullptr;
      Nx.cache_sort_part_rows = 0;
      Nx.cache_nbr = nullptr;
      Nx.cache_order = nullptr;
      Nx.cache_order_cin = Nx.cache_order_cout = 0;
      Nx.cache_order_is_table = false;
      n_out = Nx.n;
      L = Nx;
      ISF_TRY(stream_wait_stream(a, st, sg));
    }
    void* y = nullptr;
    ISF_TRY(a.alloc(&y, (size_t)std::max(n_out, 1) * ly.c_out * 4));
    const void* res = nullptr;
    if (ly.residual_from == -1) res = x_in0;
    else if (ly.residual_from >= 0) {
      ISF_REQUIRE(ly.residual_from < i && out_rows[ly.residual_from] == n_out, ISF_ERR_ARG,
                  "sparse_encoder: layer %d residual source %d incompatible", i, ly.residual_from);
      res = outputs[ly.residual_from];
    }
    if (!ev.empty()) ISF_HIP_TRY(hipEventRecord(ev[2 * i], st));
    if (use16 && srows > 0)
      ISF_TRY(sparse_conv_forward_staged_impl(x, ly.c_in, ly.packed16, K, ly.c_out, stg.slots, stride, stg.ulist,
                                              stg.ucount, n_out, ly.scale, ly.shift, res, ly.relu, y, srows, conv_mode,
                                              st));
    else if (cu && n_out > 0 && ((cu_plan.variant = (diagnostic >> 10) & 15), true))
      ISF_TRY(sparse_conv_forward_cu_impl(x, ly.c_in, ly.packed16, K, ly.c_out, nbr, stride, n_out, ly.scale, ly.shift, res,
                                          ly.relu, y, cu_plan, st));
    else if (dma)
      ISF_TRY(sparse_conv_forward_dma_impl(x, ly.c_in, ly.packed16, K, ly.c_out, nbr, stride, n_out, ly.scale, ly.shift,
                                           res, ly.relu, y, conv_mode, st, order, nullptr, lmask, nx, rowmap));
    else if (use16)
      ISF_TRY(sparse_conv_forward_f16x3_impl(x, ly.c_in, ly.packed16, K, ly.c_out, nbr, stride, n_out, ly.scale,
                                             ly.shift, res, ly.relu, y,
                                             conv_mode | ksbit | (order_is_table ? 1024 : 0) | (ly.c_out == 256 && conv_mode == (conv_mode & 32) ? wide_cols : 0) |
                                                 (ly.c_out >= 128 && conv_mode == (conv_mode & 32) ? stagger : 0),
                                             st, order, nullptr, rowmap));
    else if (sparse_conv_mfma_supported(ly.c_in, ly.c_out))
      ISF_TRY(sparse_conv_forward_packed_impl(reinterpret_cast<const float*>(x), n_in, ly.c_in, ly.packed, K,
                                              ly.c_out, nbr, stride, n_out, ly.scale, ly.shift,
                                              reinterpret_cast<const float*>(res), ly.relu,
                                              reinterpret_cast<float*>(y), st));
    else
      ISF_TRY(sparse_conv_forward_generic_impl(reinterpret_cast<const float*>(x), ly.c_in, ly.packed, K, ly.c_out,
                                               nbr, stride, n_out, ly.scale, ly.shift,
                                               reinterpret_cast<const float*>(res), ly.relu,
                                               reinterpret_cast<float*>(y), st));
    if (!ev.empty()) ISF_HIP_TRY(hipEventRecord(ev[2 * i + 1], st));
    outputs[i] = y;
    out_rows[i] = n_out;
    if (stats) { stats->num_in[i] = n_in; stats->num_out[i] = n_out; stats->ms[i] = 0.f; }
    x = y;
    c_last = ly.c_out;
  }
  // dense BEV of the last level
  ISF_TRY(ensure_occ(a, L, B, sg));
  ISF_TRY(stream_wait_stream(a, st, sg));
  if (opt && opt->bev_format == 1) {   // split-format token matrices (same byte count as the fp32 map)
    ISF_REQUIRE(use16 && !f16io, ISF_ERR_UNSUPPORTED, "sparse_encoder: bev_format 1 needs the split-precision path");
    ISF_TRY(sparse_to_bev_split_impl(x, c_last, B, L.shape[0], L.shape[1], L.shape[2], L.occ, spatial_features, st));
  } else {
    ISF_REQUIRE(!opt || opt->bev_format == 0, ISF_ERR_ARG, "sparse_encoder: bev_format %d (0 fp32 map, 1 split token matrices)",
                opt->bev_format);
    ISF_TRY(sparse_to_dense_bev_impl(a, x, use16 ? (f16io ? 2 : 1) : 0, L.coors, L.n, c_last, B, L.shape[0], L.shape[1],
                                     L.shape[2], spatial_features, &L.occ, st));
  }
  if (stats) stats->precision = use16 ? 1 : 0;
  if (out_shape) { out_shape[0] = c_last * L.shape[0]; out_shape[1] = L.shape[1]; out_shape[2] = L.shape[2]; out_shape[3] = L.n; }
  if (stats) {
    unsigned long long h_pairs[32];
    ISF_HIP_TRY(hipMemcpyAsync(h_pairs, pair_counts, sizeof(h_pairs), hipMemcpyDeviceToHost, st));
    ISF_HIP_TRY(hipStreamSynchronize(st));
    stats->num_layers = num_layers;
    for (int i = 0; i < num_layers; ++i)
      if (stats->pairs[i] < 0) stats->pairs[i] = (long long)h_pairs[-(stats->pairs[i] + 1)];
    for (int i = 0; i < num_layers && !ev.empty(); ++i)
      ISF_HIP_TRY(hipEventElapsedTime(&stats->ms[i], ev[2 * i], ev[2 * i + 1]));
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  return ISF_OK;
}

}  // namespace isf

extern "C" {

int isf_sparse_to_dense_bev(const float* features, const int32_t* indices, int num_rows, int channels,
                            int batch_size, int D, int H, int W, float* out, isf_stream_t stream) {
  ISF_REQUIRE(num_rows >= 0 && channels > 0 && batch_size > 0 && D > 0 && H > 0 && W > 0 && out, ISF_ERR_ARG,
              "sparse_to_dense_bev: bad arguments");
  ISF_REQUIRE(num_rows == 0 || (features && indices), ISF_ERR_ARG, "sparse_to_dense_bev: null pointer");
  isf::Arena& a = isf::arena_for_stream(isf::as_stream(stream));
  ISF_TRY(a.reset());
  return isf::sparse_to_dense_bev_impl(a, features, 0, indices, num_rows, channels, batch_size, D, H, W, out,
                                       nullptr, isf::as_stream(stream));
}

int isf_sparse_encoder_forward(const float* voxel_features, const int32_t* coors, int num_voxels,
                               int batch_size, const int sparse_shape_host[3],
                               const isf_conv_layer* layers_host, int num_layers, float* spatial_features,
                               int out_shape_host[4], isf_encoder_stats* stats_host, int time_layers,
                               const isf_encoder_options* options, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(num_voxels > 0 && batch_size > 0 && sparse_shape_host && layers_host && spatial_features &&
                  voxel_features && coors && num_layers > 0,
              ISF_ERR_ARG, "sparse_encoder_forward: bad arguments");
  hipStream_t st = as_stream(stream);
  Arena& a = arena_for_stream(as_stream(stream));
  ISF_TRY(a.reset());
  // The kernels want rows in rank ((b,z,y,x)-sorted) order, which is what DynamicVFE emits.  Verify it on
  // the device; arbitrary / duplicated input rows (e.g. the reference's shape test feeds random coords)
  // are re-ranked once: x_sorted[rank(coord_i)] = x[i] (a duplicate coordinate keeps one of its rows).
  OccIndex occ0;
  ISF_TRY(occ_create(a, &occ0, batch_size, sparse_shape_host[0], sparse_shape_host[1], sparse_shape_host[2], st));
  ISF_TRY(occ_mark_coords4(occ0, coors, num_voxels, st));
  ISF_TRY(occ_scan(a, occ0, st));
  int* flag = nullptr;
  ISF_TRY(a.alloc_n(&flag, 64));
  ISF_HIP_TRY(hipMemsetAsync(flag, 0, sizeof(int), st));
  hipLaunchKernelGGL(check_rank_order_kernel, dim3(ceil_div(num_voxels, 256)), dim3(256), 0, st, coors,
                     num_voxels, occ0.B, occ0.D, occ0.H, occ0.W, occ0.bits, occ0.prefix, flag);
  ISF_LAUNCH_CHECK();
  int h[2] = {0, 0};
  ISF_HIP_TRY(hipMemcpyAsync(&h[0], flag, sizeof(int), hipMemcpyDeviceToHost, st));
  ISF_HIP_TRY(hipMemcpyAsync(&h[1], occ0.total, sizeof(int), hipMemcpyDeviceToHost, st));
  ISF_HIP_TRY(hipStreamSynchronize(st));
  const float* x = voxel_features;
  const int32_t* c = coors;
  int n = num_voxels;
  if (h[0] != 0 || h[1] != num_voxels) {
    n = h[1];
    ISF_REQUIRE(n > 0, ISF_ERR_ARG, "sparse_encoder_forward: no coordinate inside sparse_shape");
    int32_t* perm = nullptr;
    ISF_TRY(build_perm(a, occ0, coors, num_voxels, &perm, st));
    int32_t* cs = nullptr;
    float* xs = nullptr;
    const int C = layers_host[0].c_in;
    ISF_TRY(a.alloc_n(&cs, (size_t)n * 4));
    ISF_TRY(a.alloc_n(&xs, (size_t)n * C));
    ISF_TRY(occ_compact_coords4(occ0, cs, st));
    hipLaunchKernelGGL(gather_rows_kernel, dim3(ceil_div((long long)n * C, 256)), dim3(256), 0, st,
                       voxel_features, perm, n, C, xs);
    ISF_LAUNCH_CHECK();
    x = xs;
    c = cs;
  }
  return sparse_encoder_forward_impl(a, x, c, n, batch_size, sparse_shape_host, &occ0, layers_host, num_layers,
                                     spatial_features, out_shape_host, stats_host, time_layers, options, st);
}

// does sparse_encoder_forward_impl run these layers on the split-format kernels (fp32-class f16x3 arithmetic)?
static bool encoder_takes_split_input(const isf_conv_layer* layers, int num_layers, const isf_encoder_options* opt) {
  if ((opt ? opt->precision : 0) != 0 || num_layers <= 0 || num_layers > 32) return false;
  for (int i = 0; i < num_layers; ++i)
    if (!layers[i].packed16 || !isf::sparse_conv_f16x3_supported(layers[i].c_in, layers[i].c_out)) return false;
  return true;
}

int isf_lidar_branch_forward(const float* points, const int64_t* point_offsets_host, int batch_size,
                             const isf_vfe_params* vfe_host, const int sparse_shape_host[3],
                             const isf_conv_layer* layers_host, int num_layers, float* spatial_features,
                             int out_shape_host[4], isf_encoder_stats* stats_host, int time_layers,
                             const isf_encoder_options* options, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(points && point_offsets_host && batch_size > 0 && vfe_host && sparse_shape_host && layers_host &&
                  spatial_features,
              ISF_ERR_ARG, "lidar_branch_forward: bad arguments");
  hipStream_t st = as_stream(stream);
  Arena& a = arena_for_stream(as_stream(stream));
  ISF_TRY(a.reset());
  const int64_t P64 = point_offsets_host[batch_size];
  ISF_REQUIRE(P64 > 0 && P64 < (1ll << 31), ISF_ERR_ARG, "lidar_branch_forward: bad point count");
  const int P = (int)P64;
  const int Cin = vfe_host->in_channels;
  int32_t* coors4 = nullptr;
  ISF_TRY(a.alloc_n(&coors4, (size_t)P * 4));
  // batches of up to 8 frames: voxelized inside the byte-map marking launch of the VFE (one launch instead of B + 1 and
  // one pass over the coordinates less); diagnostic 65536 and larger batches: one dynamic_voxelize launch per frame
  VoxBatch vb;
  vb.B = batch_size;
  const bool fused_vox = batch_size <= kVoxMaxBatch && !(options && (options->diagnostic & 65536));
  for (int b = 0; b < batch_size; ++b) {
    const int64_t lo = point_offsets_host[b], hi = point_offsets_host[b + 1];
    ISF_REQUIRE(hi >= lo, ISF_ERR_ARG, "lidar_branch_forward: bad offsets");
    if (fused_vox) {
      vb.off[b] = lo;
      vb.off[b + 1] = hi;
    } else {
      ISF_TRY(dynamic_voxelize_impl(points + lo * Cin, (int)(hi - lo), Cin, vfe_host->voxel_size,
                                    vfe_host->coors_range, coors4 + lo * 4, 4, 1, b, st));
    }
  }
  float* vf = nullptr;
  int32_t* vc = nullptr;
  ISF_TRY(a.alloc_n(&vf, (size_t)P * vfe_host->c2));
  ISF_TRY(a.alloc_n(&vc, (size_t)P * 4));
  int n0 = 0;
  OccIndex occ0;
  hipEvent_t coords_ready = nullptr;
  // the voxel rows go straight into the split format the first convolution reads (whole rows from the VFE's segmented
  // max, the rows cut by a wave boundary from a fix-up pass): no [N, 64] fp32 -> split conversion pass in between
  void* vf_split = nullptr;
  if (encoder_takes_split_input(layers_host, num_layers, options) && vfe_host->c2 == layers_host[0].c_in &&
      !(options && (options->diagnostic & 256)))   // diagnostic 256: fp32 rows + the conversion pass (same bits)
    ISF_TRY(a.alloc(&vf_split, (size_t)P * vfe_host->c2 * 4));
  ISF_TRY(dynamic_vfe_impl(a, points, coors4, P, Cin, batch_size, vfe_host->voxel_size, vfe_host->coors_range,
                           vfe_host->w1, vfe_host->scale1, vfe_host->shift1, vfe_host->c1, vfe_host->w2,
                           vfe_host->scale2, vfe_host->shift2, vfe_host->c2, vf, vc, nullptr, &n0, &occ0,
                           sparse_shape_host[0], st, &coords_ready, vf_split, fused_vox ? &vb : nullptr));
  ISF_REQUIRE(n0 > 0, ISF_ERR_ARG, "lidar_branch_forward: no point falls inside the voxel grid");
  const bool occ_ok = occ0.D == sparse_shape_host[0] && occ0.H == sparse_shape_host[1] &&
                      occ0.W == sparse_shape_host[2];
  return sparse_encoder_forward_impl(a, vf, vc, n0, batch_size, sparse_shape_host, occ_ok ? &occ0 : nullptr,
                                     layers_host, num_layers, spatial_features, out_shape_host, stats_host,
                                     time_layers, options, st, occ_ok ? coords_ready : nullptr, vf_split);
}

}  // extern "C"
