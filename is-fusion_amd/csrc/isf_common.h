// isf_common.h -- internal helpers shared by the HIP translation units of libisf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <vector>

#include "isf_hip.h"

namespace isf {

// ----------------------------------------------------------------------------- error handling
void set_error(const char* fmt, ...);

#define ISF_HIP_TRY(expr)                                                                   \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      ::isf::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return ISF_ERR_HIP;                                                                   \
    }                                                                                       \
  } while (0)

#define ISF_REQUIRE(cond, code, ...)  \
  do {                                \
    if (!(cond)) {                    \
      ::isf::set_error(__VA_ARGS__);  \
      return (code);                  \
    }                                 \
  } while (0)

#define ISF_TRY(expr)          \
  do {                         \
    int _r = (expr);           \
    if (_r != ISF_OK) return _r; \
  } while (0)

#define ISF_LAUNCH_CHECK() ISF_HIP_TRY(hipGetLastError())

// ----------------------------------------------------------------------------- per-stream workspace
// One workspace per (device, caller stream): a bump allocator over a few large hipMalloc'd blocks plus the helper
// objects a call needs (side stream, recycled events, the persistent byte maps of the level-0 occupancy index).  reset()
// at the start of every top-level API call; if more than one block had to be created the arena is coalesced into one
// block at the next reset (after a device sync), so the steady state is exactly one block and zero hipMalloc calls.
// Calls on DIFFERENT streams (or from different host threads on different streams) never share workspace memory;
// calls on ONE stream are ordered by the stream, so reusing its workspace from call to call is safe.  The registry is
// mutex-protected; a single workspace is not: issue the calls of one stream from one host thread at a time.
struct ByteMaps {
  unsigned char* fine = nullptr;
  unsigned char* coarse = nullptr;
  size_t words = 0;
};

class Arena {
 public:
  int reset();                              // start of a top-level call
  int alloc(void** out, size_t bytes);      // 256-byte aligned
  template <typename T>
  int alloc_n(T** out, size_t n) { return alloc(reinterpret_cast<void**>(out), n * sizeof(T)); }
  int release();                            // frees blocks, byte maps, side stream, events
  size_t capacity() const;
  int list_blocks(unsigned long long* base_cap, int max_pairs, int* num_pairs) const;   // diagnostic
  // helper objects owned by the workspace
  hipStream_t side = nullptr;               // geometry / rulebook work of the sparse encoder overlaps the convolutions
  std::vector<hipEvent_t> events;           // recycled hipEventDisableTiming events
  size_t next_event = 0;
  ByteMaps bytemaps;
  // host mailbox (post_int / wait_int): a ring of (value, ticket) pairs in pinned, device-mapped host memory
  int* mailbox_host = nullptr;
  int* mailbox_dev = nullptr;
  unsigned mailbox_seq = 0;
  // kZeroInts ints that are ZERO between top-level calls: a kernel chain that counts into them puts the zeros back itself
  // (its last reader), which saves the memset in front of the chain (zero_ints())
  int* zero_pool = nullptr;
  // CHUNK-SPLIT convolutions (isf_spconv16.hip, conv mode bit 524288): the partial accumulator tiles the two workgroups of
  // a tile exchange, and one arrival counter per tile (zero between launches: the second arriver puts the zero back)
  float* ks_scratch = nullptr;
  size_t ks_scratch_bytes = 0;
  unsigned* ks_count = nullptr;
  int ks_count_cap = 0;
 private:
  struct Block { char* base; size_t cap; size_t off; };
  std::vector<Block> blocks_;
};

Arena& arena_for_stream(hipStream_t st);    // workspace of (current device, st); created on first use

static inline hipStream_t as_stream(isf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline size_t round_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// ----------------------------------------------------------------------------- occupancy index
// Minimal perfect hash of the active voxel set: one bit per grid cell + an exclusive popcount prefix
// per 64-bit word.  rank(cell) = prefix[word] + popc(bits[word] & below(bit)) is the row of that voxel
// in (b,z,y,x)-sorted order -- i.e. the order at::unique_dim(sorted=true) produces -- with no sort.
struct OccIndex {
  int B, D, H, W;
  unsigned long long ncells;
  size_t nwords;
  unsigned long long* bits;  // [nwords]
  uint32_t* prefix;          // [nwords] exclusive popcount prefix
  int* total;                // device scalar: number of set bits
};

int occ_create(Arena& a, OccIndex* occ, int B, int D, int H, int W, hipStream_t st, bool zero = true);
size_t occ_bits_bytes(const OccIndex& occ);   // bytes occ_create(zero = true) clears
// atomic-free marking through persistent byte maps (writes EVERY bitmap word: create with zero = false)
int occ_mark_coords4_bytemap(Arena& a, const OccIndex& occ, const int32_t* coors4, int n, hipStream_t st);
int scan_u32_exclusive(Arena& a, const uint32_t* in, uint32_t* out, size_t n, hipStream_t st);
int occ_scan(Arena& a, const OccIndex& occ, hipStream_t st);                           // prefix + total
int occ_mark_coords4(const OccIndex& occ, const int32_t* coors4, int n, hipStream_t st);
// coords of all set bits in rank order -> out [total,4]
int occ_compact_coords4(const OccIndex& occ, int32_t* out, hipStream_t st);
// up to four byte fills in one launch; every buffer must be an arena allocation (256-byte aligned, padded to 256 bytes:
// the fill works in 8-byte words)
int fill_many(hipStream_t st, int n, void* const* ptrs, const size_t* bytes, const unsigned char* byte_values);
int read_int(const int* dev, int* host, hipStream_t st);  // async copy + stream sync
// A device int -> the host WITHOUT a copy command: a one-thread kernel writes (value, ticket) into pinned host memory
// behind the producer on `st` and the host spins on the ticket.  hipMemcpyAsync D2H + synchronise costs a copy-engine
// command with 20-40 us of GPU idle around it every time a data-dependent count sizes the next launches
// (profiles/r04_v1_timeline_gaps.txt: every large gap of a step sits behind a copyBuffer).  post_int returns at once;
// work queued between post_int and wait_int runs while the host waits.
int post_int(Arena& a, const int* dev, hipStream_t st, unsigned* ticket);
int wait_int(Arena& a, unsigned ticket, hipStream_t st, int* value);
int side_stream(Arena& a, hipStream_t* out);               // the workspace's non-blocking helper stream
static constexpr int kZeroInts = 256;
int zero_ints(Arena& a, int** out);                        // the workspace's self-restoring zero counters (kZeroInts)
// the workspace's chunk-split buffers, grown on demand (a grow synchronises `st` first: earlier launches may still read them)
int ksplit_buffers(Arena& a, size_t scratch_bytes, int counters, float** scratch, unsigned** count, hipStream_t st);
int stream_wait_stream(Arena& a, hipStream_t waiter, hipStream_t producer);  // event from the workspace's pool
int pooled_event(Arena& a, hipEvent_t* out);               // recycled hipEventDisableTiming events

__device__ __forceinline__ int occ_lookup(const unsigned long long* __restrict__ bits,
                                          const uint32_t* __restrict__ prefix,
                                          unsigned long long cell) {
  const size_t w = (size_t)(cell >> 6);
  const unsigned long long word = bits[w];
  const unsigned long long bit = 1ull << (cell & 63);
  if (!(word & bit)) return -1;
  return (int)(prefix[w] + (uint32_t)__popcll(word & (bit - 1)));
}

// ----------------------------------------------------------------------------- attention-probability dropout
// nn.MultiheadAttention(dropout = p) in training mode zeroes attention PROBABILITIES with probability p and scales the kept
// ones by 1 / (1 - p) (fusion_encoder.py:458 F.dropout(attn_output_weights)).  The flash-style kernels never hold the
// probability matrix, so forward and backward recompute the same keep / drop decision per (sample * heads + head, query, key)
// from a counter-based hash of a per-call seed (the finaliser of MurmurHash3 over the packed indices).  thresh = p * 2^32;
// thresh == 0: no dropout (every kernel skips the hash).  fusion_ops.attention_keep_mask restates it in torch for the tests.
struct AttnDrop {
  unsigned long long seed;
  unsigned thresh;
  float inv_keep;
};
__host__ __device__ inline bool attn_keep(unsigned long long seed, unsigned bh, unsigned i, unsigned j, unsigned thresh) {
  unsigned long long x = seed ^ (((unsigned long long)bh << 48) | ((unsigned long long)(i & 0xffffffu) << 24) |
                                 (unsigned long long)(j & 0xffffffu));
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return (unsigned)(x >> 32) >= thresh;
}
static inline AttnDrop attn_drop_of(float p, unsigned long long seed) {
  AttnDrop d{seed, 0u, 1.f};
  if (p > 0.f) {
    const double t = (double)p * 4294967296.0;
    d.thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    d.inv_keep = 1.f / (1.f - p);
  }
  return d;
}

// ----------------------------------------------------------------------------- split activation format
// Row-major [N][C/32] chunks of 128 bytes; a chunk holds 32 channels as 4 x (8 f16 hi) followed by 4 x (8 f16 lo)
// (value = hi + lo).  The four 16-byte hi pieces of a chunk -- what the four k-group lanes of one MFMA row fetch
// with ONE load instruction -- are contiguous (64 B), and so are the lo pieces: the texture addresser coalesces a
// row's lanes into one request.  (Interleaving hi|lo per 8 channels made every lane a separate 16-byte request at
// a 32-byte stride: 64 requests per gather instruction instead of 16.)
// Index, in 16-byte units, of the hi piece of 8-channel unit u of `row` (c_units = C/8); the lo piece is +4.
__host__ __device__ inline size_t split_hi_index(size_t row, int c_units, int u) {
  return (row * (size_t)c_units + (size_t)(u & ~3)) * 2 + (size_t)(u & 3);
}

// ----------------------------------------------------------------------------- voxel geometry (isf_voxelize.hip, isf_runtime.hip)
__device__ __forceinline__ bool voxel_of_point(const float* __restrict__ p, float vx, float vy, float vz,
                                               float x0, float y0, float z0, int gx, int gy, int gz,
                                               int& cx, int& cy, int& cz) {
  // fp32 subtract, fp32 IEEE divide, floor -- exactly voxelization_cpu.cpp:24 / voxelization_cuda.cu:37
  const float fx = floorf(__fdiv_rn(__fsub_rn(p[0], x0), vx));
  const float fy = floorf(__fdiv_rn(__fsub_rn(p[1], y0), vy));
  const float fz = floorf(__fdiv_rn(__fsub_rn(p[2], z0), vz));
  // float -> int: everything outside [0, grid) (incl. NaN / huge) is invalid
  if (!(fx >= 0.f && fx < (float)gx && fy >= 0.f && fy < (float)gy && fz >= 0.f && fz < (float)gz))
    return false;
  cx = (int)fx; cy = (int)fy; cz = (int)fz;
  return true;
}

struct VoxGeom {
  float vx, vy, vz, x0, y0, z0;
  int gx, gy, gz;
};

static inline VoxGeom make_geom(const float vs[3], const float range[6]) {
  VoxGeom g;
  g.vx = vs[0]; g.vy = vs[1]; g.vz = vs[2];
  g.x0 = range[0]; g.y0 = range[1]; g.z0 = range[2];
  // grid = round((max-min)/vs) in fp32 (voxelization_cpu.cpp:120-123)
  g.gx = (int)roundf((range[3] - range[0]) / vs[0]);
  g.gy = (int)roundf((range[4] - range[1]) / vs[1]);
  g.gz = (int)roundf((range[5] - range[2]) / vs[2]);
  return g;
}


// B <= kVoxMaxBatch frames of one launch: frame b owns points [off[b], off[b + 1])
static constexpr int kVoxMaxBatch = 8;
struct VoxBatch {
  long long off[kVoxMaxBatch + 1];
  int B;
};
// dynamic voxelization of ALL frames + byte-map marking of the level-0 occupancy index in ONE launch (the LiDAR branch ran
// one voxelize launch per frame and re-read the coordinates in a separate mark launch); coors4 [P, 4] is written here.
int occ_voxelize_mark_bytemap(Arena& a, const OccIndex& occ, const float* points, int P, int C, const VoxGeom& g,
                              const VoxBatch& vb, int32_t* coors4, hipStream_t st);

// ----------------------------------------------------------------------------- internal ops (arena-aware)
// isf_voxelize.hip
int dynamic_voxelize_impl(const float* points, int P, int C, const float vs[3], const float range[6],
                          int32_t* coors, int coors_stride, int coors_col0, int batch_idx,
                          hipStream_t st);
// isf_vfe.hip
int dynamic_vfe_impl(Arena& a, const float* points, const int32_t* coors4, int P, int Cin, int B,
                     const float vs[3], const float range[6], const float* w1, const float* scale1,
                     const float* shift1, int c1, const float* w2, const float* scale2,
                     const float* shift2, int c2, float* voxel_feats, int32_t* voxel_coors,
                     int32_t* pt2vox, int* num_voxels_host, OccIndex* occ_out, int grid_d_alloc,
                     hipStream_t st, hipEvent_t* coords_ready = nullptr,
                     void* voxel_feats_split = nullptr /* [P, c2] rows in the split format as well (whole rows are written
                     there INSTEAD of voxel_feats; rows cut by a wave boundary in both) */,
                     const VoxBatch* voxelize = nullptr /* non-null: coors4 is an OUTPUT -- the frames are voxelized (vs,
                     range) inside the byte-map marking launch (occ_voxelize_mark_bytemap) */);
// isf_rulebook.hip
int build_perm(Arena& a, const OccIndex& occ, const int32_t* coors4, int n, int32_t** perm_out,
               hipStream_t st);
int launch_nbr(Arena& a, const int32_t* out_coors4, int n_out, const int in_shape[3], const int ks[3],
               const int st[3], const int pd[3], bool subm, const OccIndex& in_occ, const int32_t* perm,
               int32_t* nbr, int nbr_stride, unsigned long long* pair_count, hipStream_t st_);
// line-compressed table (isf_rulebook.hip): lines [ks0 * ks1][stride] + mask [stride], rank order only
int launch_nbr_lines(Arena& a, const int32_t* out_coors4, int n_out, const int in_shape[3], const int ks[3],
                     const int st[3], const int pd[3], bool subm, const OccIndex& in_occ, int32_t* lines, uint32_t* mask,
                     int stride, unsigned long long* pair_count, hipStream_t st_);
int launch_mark_out(const int32_t* in_coors4, int n_in, const int in_shape[3], const int ks[3],
                    const int st[3], const int pd[3], const OccIndex& out_occ, hipStream_t st_);
// isf_spconv.hip
bool sparse_conv_mfma_supported(int c_in, int c_out);
int pack_filters_impl(const float* w, int K, int cin, int cout, float* packed, hipStream_t st);
int sparse_conv_forward_packed_impl(const float* x, int n_in, int c_in, const float* packed, int K,
                                    int c_out, const int32_t* nbr, int nbr_stride, int n_out,
                                    const float* scale, const float* shift, const float* residual,
                                    int relu, float* y, hipStream_t st);
int sparse_conv_forward_generic_impl(const float* x, int c_in, const float* w, int K, int c_out,
                                     const int32_t* nbr, int nbr_stride, int n_out, const float* scale,
                                     const float* shift, const float* residual, int relu, float* y,
                                     hipStream_t st);
// isf_spconv16.hip
bool sparse_conv_f16x3_supported(int c_in, int c_out);
int sparse_conv_forward_f16x3_impl(const void* xs, int c_in, const void* packed16, int K, int c_out,
                                   const int32_t* nbr, int nbr_stride, int n_out, const float* scale,
                                   const float* shift, const void* residual, int relu, void* ys,
                                   int mode /* 0 | 1 single-pass f16 | timing diagnostics */, hipStream_t st,
                                   const int32_t* order = nullptr, struct Conv16LaunchInfo* query = nullptr,
                                   const int32_t* rowmap = nullptr /* sorted launch: position -> output row (conv_row_sort_impl) */);
// How a launch of that kernel is cut into tiles (query != nullptr: filled instead of launching), and the per-part tile
// order that evens out the work of the tiles sharing a CU (conv16_tile_order_impl; nullptr = slot j works on tile j).
struct Conv16LaunchInfo {
  int full, half, part_rows;   // Conv16Plan
  int TM, ncb, wgs_per_cu, cus_per_xcd;
};
static inline int conv16_order_parts(const Conv16LaunchInfo& i) { return i.ncb == 2 ? 4 : 8; }
static inline int conv16_order_tiles(const Conv16LaunchInfo& i) { return i.full + i.half; }
// a reordering only helps a launch that is resident in one round with CUs that hold more than one tile
static inline bool conv16_order_applies(const Conv16LaunchInfo& i) {
  const int t = i.full + i.half;
  return t > i.cus_per_xcd && t <= i.wgs_per_cu * i.cus_per_xcd && t <= 255 && i.cus_per_xcd <= 64;
}
// a tile table needs the whole part resident in one round: groups per part <= CUs x workgroups per CU x groups per tile
static inline bool conv16_table_applies(const Conv16LaunchInfo& i) {
  return i.TM > 0 && i.cus_per_xcd > 0 && i.part_rows / 16 <= i.cus_per_xcd * i.wgs_per_cu * (i.TM / 16);
}
static inline int conv16_table_ints(const Conv16LaunchInfo& i) {
  return conv16_order_parts(i) * 2 * i.wgs_per_cu * i.cus_per_xcd;
}
int conv16_tile_table_impl(const int32_t* group_work, int n_out, const Conv16LaunchInfo& info, int32_t* table,
                           hipStream_t st);
// isf_spconv_cu.hip: per 16-row group the taps through which one of its rows has a neighbour (masks) and their count (work)
int conv_group_masks_impl(const int32_t* nbr, int nbr_stride, int K, int n_out, int32_t* masks, int32_t* work,
                          hipStream_t st);
// isf_spconv_dma.hip: the same convolution for the narrow layers (<= 64 channels in and out) with the gathered rows
// brought in by LDS-DMA, one cache line per lane quad; bit-identical to sparse_conv_forward_f16x3_impl
bool sparse_conv_dma_supported(int c_in, int c_out);
int sparse_conv_forward_dma_impl(const void* xs, int c_in, const void* packed16, int K, int c_out, const int32_t* nbr,
                                 int nbr_stride, int n_out, const float* scale, const float* shift,
                                 const void* residual, int relu, void* ys, int mode, hipStream_t st,
                                 const int32_t* order = nullptr, Conv16LaunchInfo* query = nullptr,
                                 const uint32_t* lmask = nullptr /* line-compressed table: nbr = lines */, int nx = 0,
                                 const int32_t* rowmap = nullptr /* sorted launch: position -> output row */,
                                 long long* trace = nullptr /* mode 512: per-workgroup trace (isf_sparse_conv_dma_trace) */);
int conv16_tile_order_impl(const int32_t* nbr, int nbr_stride, int K, int n_out, const Conv16LaunchInfo& info,
                           int32_t* work /* [parts * tiles] scratch */, int32_t* order /* [parts * tiles] */,
                           hipStream_t st, const uint32_t* lmask = nullptr /* line-compressed table's masks instead of nbr */);
// isf_voxelize.hip: hand-written stable LSD radix sort (wave multi-split): idx_sorted = stable order of the keys' low bits
int stable_sort_u32_impl(Arena& a, const uint32_t* keys, int n, int key_bits, int* idx_sorted, hipStream_t st);
// ROW SORT of a deep SubM launch (round 6): the positions of a launch's part are dealt to its rows in the order of their
// TAP MASKS, so that the rows of a tile (and of a 16-row group) want the same taps -- a tile walks popcount(OR of its rows'
// masks) taps, a group multiplies through popcount(OR of its 16 masks): -15 % steps and -17 % MFMAs at level 3 on the
// benchmark geometry (CPU census: profiles/r06_row_sort.txt).  rowmap [stride]: position -> row; nbr_sorted [K][stride]:
// the neighbour table by position.  The conv kernel computes positions and reads / writes residual and output rows
// through rowmap; every output row is still computed by the same products in the same order: bit-identical.
int conv_row_sort_impl(Arena& a, const int32_t* nbr, int nbr_stride, int K, int n_out, int part_rows, int32_t* rowmap,
                       int32_t* nbr_sorted, hipStream_t st,
                       int key_mode = 1 /* 1: 6 coarse bits (which ky rows of the planes above / below hold a neighbour), one radix
                                           pass; 2: coarse | the nine in-plane taps, two passes */);
// the same for a LINE-COMPRESSED table (lines [num_lines][stride] + 27-bit masks [stride]): the keys come from the masks;
// full_key: all 27 mask bits decide (the finest level, whose rows have few and varied neighbours) instead of 16
int conv_row_sort_lines_impl(Arena& a, const int32_t* lines, const uint32_t* lmask, int nbr_stride, int num_lines, int n_out,
                             int part_rows, bool full_key, int32_t* rowmap, int32_t* lines_sorted, uint32_t* lmask_sorted,
                             hipStream_t st);
// tile order of a launch of several rounds: the tiles of an XCD band by band in y (z-neighbour rows stay in its L2)
bool conv16_band_order_applies(const Conv16LaunchInfo& info);
int conv16_band_order_impl(const int32_t* coors4, int n_out, const Conv16LaunchInfo& info, int band, int32_t* order /* [parts * tiles] */,
                           hipStream_t st);
// isf_spconv_deep.hip (round 6): the deep layers' 4-wave two-group launches with LDS-DMA gathers and one hand-scheduled
// instruction stream per step; the tile kernel's plan / order / query semantics; bit-identical to it
bool sparse_conv_deep_supported(int c_in, int c_out);
int sparse_conv_forward_deep_impl(bool balance, bool table, const uint4* xs, int c_in, const uint4* wpk, const float* winv,
                                  int K, int c_out, const int32_t* nbr, int nbr_stride, int n_out, const float* scale,
                                  const float* shift, const uint4* residual, int relu, uint4* ys, hipStream_t st,
                                  const int32_t* order, Conv16LaunchInfo* query);
// isf_spconv_cu.hip: the same convolution for the 256-column layers as one workgroup per compute unit over units of equal
// matrix work (plan built once per rulebook); bit-identical to sparse_conv_forward_f16x3_impl
struct ConvCuPlan {
  const int32_t* group_masks = nullptr;   // [ceil(n_out / 16)] taps a 16-row group multiplies through
  const int2* units = nullptr;            // [*num_units] (first group, groups)
  const int32_t* num_units = nullptr;     // device scalar
  int max_units = 0;                      // host-side bound of *num_units (sizes the grid: no host sync)
  int n_out = 0;
  int variant = 0;                        // 0 = production; timing diagnostics / pipeline depths: isf_spconv_cu.hip
  int cap = 16;                           // groups per unit the plan was cut for (16: one workgroup per CU; 8: two)
};
int conv_cu_variant_cap(int variant);     // the unit shape a kernel variant works on
bool sparse_conv_cu_supported(int c_in, int c_out);
size_t conv_cu_plan_ints(int n_out);      // enough for either shape
int conv_cu_plan_impl(const int32_t* nbr, int nbr_stride, int K, int n_out, int32_t* buf /* conv_cu_plan_ints(n_out) */,
                      ConvCuPlan* plan, hipStream_t st, int cap = 16);
int sparse_conv_forward_cu_impl(const void* xs, int c_in, const void* packed16, int K, int c_out, const int32_t* nbr,
                                int nbr_stride, int n_out, const float* scale, const float* shift, const void* residual,
                                int relu, void* ys, const ConvCuPlan& plan, hipStream_t st);
// isf_spconv_stage.hip (LDS-staged input rows; staging tables of a rulebook)
int stage_tables_impl(const int32_t* nbr, int nbr_stride, int K, uint16_t* slots, int32_t* ulist, int32_t* ucount,
                      hipStream_t st);
int sparse_conv_forward_staged_impl(const void* xs, int c_in, const void* packed16, int K, int c_out,
                                    const uint16_t* slots, int nbr_stride, const int32_t* ulist,
                                    const int32_t* ucount, int n_out, const float* scale, const float* shift,
                                    const void* residual, int relu, void* ys, int stage_rows, int mode,
                                    hipStream_t st);
int pack_filters16_impl(Arena& a, const float* w, int K, int cin, int cout, void* packed16, hipStream_t st,
                        bool transposed = false /* w = [K][cout][cin], the per-tap transpose of the packed filter */);
int f32_to_split_impl(const float* x, size_t n_elems, void* xs, hipStream_t st);
int split_to_f32_impl(const void* xs, size_t n_elems, float* x, hipStream_t st);
int f32_to_half_impl(const float* x, size_t n_elems, void* xh, hipStream_t st);   // [N][C] f16 rows (f16 storage mode)
int half_to_f32_impl(const void* xh, size_t n_elems, float* x, hipStream_t st);
// isf_encoder.hip
// fmt: 0 = fp32 rows, 1 = split rows, 2 = f16 rows
int sparse_to_dense_bev_impl(Arena& a, const void* feats, int fmt, const int32_t* indices, int n, int C,
                             int B, int D, int H, int W, float* out, const OccIndex* occ,
                             hipStream_t st);

}  // namespace isf
