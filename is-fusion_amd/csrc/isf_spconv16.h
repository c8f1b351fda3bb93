// isf_spconv16.h -- device helpers of the f16x3 sparse-conv kernel (isf_spconv16.hip): fragment types, LDS-DMA, the
// split format, the XCD-aware tile mapping and the epilogue.
#pragma once
#include "isf_common.h"

namespace isf {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

static constexpr int kMaxTaps = 27;

// narrow layers (CIN <= 64, <= 64 output columns) run all their 32-channel chunks in one step: half / the same
// number of barriers for twice the MFMAs per barrier
template <int CIN, int NT>
struct Conv16Step {
  static constexpr int KCH = (CIN <= 64 && NT <= 4) ? CIN / 32 : 1;      // 32-channel chunks per step
};

// LDS-DMA of 16 B per lane: LDS[lds_base + lane*16] = *gsrc.  Issued through inline asm on purpose: when hipcc
// sees a global_load_lds it drains vmcnt(0) before every later ds_read (it cannot prove the buffers differ),
// which would serialise the next step's weight/activation prefetch behind the current step's MFMAs.  Hidden
// here, the DMA stays in flight during the compute; the loops wait for it explicitly (s_waitcnt vmcnt(N) +
// barrier) right before the buffer is read.  M0 carries the wave-uniform LDS base and is restored.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_base_bytes /* wave-uniform */) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_base_bytes)
      : "memory");
}

// N consecutive KiB in one go: LDS[lds_base + i * 1024 + lane * 16] = gsrc[i * 64] for i < N -- ONE M0 set-up and N loads with
// immediate offsets (the offset field moves the global AND the LDS address).  The phase trace put the four separate
// glds16 of a wave's weight share at 410 cycles per step, 12 % of it (profiles/r05_att_256_v2.txt): each carried its own
// m0 save / set / nop / restore and address arithmetic.
template <int N>
__device__ __forceinline__ void glds16_run(const void* gsrc, unsigned lds_base_bytes /* wave-uniform */) {
  static_assert(N == 1 || N == 2 || N == 4, "1, 2 or 4 KiB");
  unsigned keep;
  if constexpr (N == 1)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_base_bytes) : "memory");
  else if constexpr (N == 2)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_base_bytes) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, off offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, off offset:3072\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_base_bytes) : "memory");
}

// the hi halves only (single-pass f16 modes): of the N KiB of a run, the even ones -- N / 2 loads at offsets 0, 2048
template <int N>
__device__ __forceinline__ void glds16_run_hi(const void* gsrc, unsigned lds_base_bytes /* wave-uniform */) {
  static_assert(N == 2 || N == 4, "2 or 4 KiB");
  unsigned keep;
  if constexpr (N == 2)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_base_bytes) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %1, off offset:2048\n\t"
                 "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_base_bytes) : "memory");
}

__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

__device__ __forceinline__ void split8(const f32x8 v, uint4& hi, uint4& lo) {
  const h8 h = __builtin_convertvector(v, h8);
  const f32x8 r = v - __builtin_convertvector(h, f32x8);
  const h8 l = __builtin_convertvector(r, h8);
  hi = *reinterpret_cast<const uint4*>(&h);
  lo = *reinterpret_cast<const uint4*>(&l);
}

__device__ __forceinline__ f32x8 join8(const uint4 hi, const uint4 lo) {
  const h8 h = *reinterpret_cast<const h8*>(&hi);
  const h8 l = *reinterpret_cast<const h8*>(&lo);
  return __builtin_convertvector(h, f32x8) + __builtin_convertvector(l, f32x8);
}

// XCD-aware tile mapping: workgroups are dealt round-robin to the 8 XCDs (each with its own 4 MB L2) in
// linear-id order.  XCD x works through ONE CONTIGUOUS range of rows (rows are (b,z,y,x)-sorted, so the y / z
// neighbours a tile gathers are rows of tiles the same XCD touches a little earlier or later: its L2 holds that
// sliding window instead of every XCD fetching every row), and with two column blocks only ever on column block
// x & 1, so that the weights it streams are half of the layer's.
// A row range ("part": 4 with two column blocks, 8 with one) is cut into `full` tiles of TM rows followed by `half`
// tiles of TM / 2 rows (every wave of a half tile owns ONE 16-row group instead of two) -- see conv16_plan.
// -> false when this workgroup has no tile.
struct Conv16Plan {
  int full, half;   // tiles per part.  half < 0: TABLE mode -- `full` slots per part, slot j of a part works on the tile
                    // (first 16-row group, groups) the launch's tile table names (conv16_table_part)
  int part_rows;    // rows per part (a multiple of 16; the parts split the rows evenly whatever the tile mix)
};

// tile j of part `part` -> its first row, the first row past its share (the part's end or n_out, whichever comes
// first) and whether it is a half tile; false when the tile is empty.  Host and device: tests/test_tile_plan.py walks
// the same arithmetic the kernel uses.
__host__ __device__ inline bool conv16_tile_rows(Conv16Plan plan, int TM, int n_out, int part, int j, int& row0,
                                                 int& row_end, bool& half) {
  if (j >= plan.full + plan.half) return false;
  half = j >= plan.full;
  const int off = half ? plan.full * TM + (j - plan.full) * (TM / 2) : j * TM;
  row0 = part * plan.part_rows + off;
  const int part_end = (part + 1) * plan.part_rows;
  row_end = part_end < n_out ? part_end : n_out;
  return off < plan.part_rows && row0 < n_out;
}

// order (optional): [parts][full + half] -- the tile workgroup slot j of a part works on (a permutation per part, built
// by conv16_tile_order_impl so that the tiles sharing a CU add up to about the same work).
__device__ __forceinline__ bool conv16_tile_of_block(int ncb, Conv16Plan plan, int TM, int n_out, int& cb, int& row0,
                                                     int& row_end, bool& half,
                                                     const int32_t* __restrict__ order = nullptr, int bid = -1) {
  if (bid < 0) bid = (int)blockIdx.x;     // (chunk-split launches: the workgroup's id within its half of the grid)
  const int xcd = bid & 7;
  int j = bid >> 3;
  cb = ncb == 2 ? xcd & 1 : 0;
  const int part = ncb == 2 ? xcd >> 1 : xcd;
  if (plan.half < 0) {   // tile table: (first group, groups) per slot; a tile of <= TM / 32 groups runs as a half tile
    const int g0 = order[2 * (part * plan.full + j)], ng = order[2 * (part * plan.full + j) + 1];
    row0 = g0 * 16;
    row_end = row0 + ng * 16 < n_out ? row0 + ng * 16 : n_out;
    half = ng * 32 <= TM;
    return ng > 0 && row0 < n_out;
  }
  if (order) j = order[part * (plan.full + plan.half) + j];
  return conv16_tile_rows(plan, TM, n_out, part, j, row0, row_end, half);
}
static inline int conv16_grid_blocks(Conv16Plan plan) { return 8 * (plan.half < 0 ? plan.full : plan.full + plan.half); }

// TILE TABLE of a launch that is resident in one round of workgroups: instead of cutting a part's rows into equal tiles
// (whose matrix work varies 3x with the density of the scene, so that the launch ends with its densest compute unit),
// the part's 16-row groups are dealt to the XCD's compute units as contiguous runs of about EQUAL WORK (work of a group =
// taps through which one of its rows has a neighbour, + a constant for its fixed cost), at most wgs_per_cu * gt groups
// per CU, and a CU's run is cut into full tiles of gt groups plus one tile with the remainder; tile k of CU c goes to
// workgroup slot c + cus * k, which is where the dispatcher puts it (round-robin over the XCD's CUs,
// tools/probes/wg_placement.hip).  Dense CUs get fewer rows, sparse ones more.  Host and device walk the same code
// (tests/test_tile_plan.py).  -> false when the part does not fit one round (the caller keeps the uniform plan).
static constexpr int kTableGroupBias = 2;
__host__ __device__ inline bool conv16_table_part(const int32_t* work, int G0, int G1, int cus, int wgs_per_cu, int gt,
                                                  int32_t* tiles /* [wgs_per_cu * cus][2] */) {
  const int slots = wgs_per_cu * cus, cap = wgs_per_cu * gt;
  for (int j = 0; j < 2 * slots; ++j) tiles[j] = 0;
  if (G1 - G0 > cus * cap) return false;
  long long rem = 0;
  for (int g = G0; g < G1; ++g) rem += work[g] + kTableGroupBias;
  int g = G0;
  for (int c = 0; c < cus && g < G1; ++c) {
    const int left_cus = cus - c, left_groups = G1 - g;
    const long long target = (rem + left_cus - 1) / left_cus;
    int must = left_groups - (left_cus - 1) * cap;     // what the others cannot hold
    if (must < 0) must = 0;
    int take = 0;
    long long acc = 0;
    while (g + take < G1 && take < cap) {
      const long long w = work[g + take] + kTableGroupBias;
      if (take >= must && (acc >= target || (take > 0 && acc + w - target > target - acc))) break;
      acc += w;
      ++take;
    }
    for (int k = 0, off = 0; off < take; ++k) {
      const int ng = take - off < gt ? take - off : gt;
      tiles[2 * (c + cus * k)] = g + off;
      tiles[2 * (c + cus * k) + 1] = ng;
      off += ng;
    }
    g += take;
    rem -= acc;
  }
  return g == G1;
}

// How a launch is cut into tiles.  The matrix pipe is per SIMD and a workgroup puts one wave (NW = 4) on each SIMD of
// its CU, so what bounds a launch that fits the chip in ONE round of workgroups is the largest number of 16-row groups
// any SIMD ends up with.  With uniform TM-row tiles that is wgs_per_cu * RG on the CUs that get a full set of
// workgroups while the others idle: 636 workgroups on 256 CUs x 3 slots = 6 groups per SIMD on 124 CUs, 4 on 132, for
// an average need of 4.97.  Mixing full and half tiles -- per CU f full + (wgs_per_cu - f) half, dealt in that order:
// the dispatcher places workgroups round-robin over the CUs of an XCD, every CU gets the same set
// (tools/probes/wg_placement.hip, profiles/r02_call13_tile_mix.txt) -- brings the maximum down to ceil(need).
// Launches of several rounds keep uniform tiles (the dispatcher refills slots as they drain).
// align_groups: every tile starts at a multiple of 16 * align_groups rows (the LDS-staged kernel works on 64-row units
// of the rulebook's staging tables: align_groups = 4; TM / 2 must be a multiple of it too).
static inline Conv16Plan conv16_plan(int n_out, int TM, int ncb, int wgs_per_cu, int cus_per_xcd, bool balance,
                                     int align_groups = 1) {
  const int parts = ncb == 2 ? 4 : 8;
  const int gt = TM / 16;                                       // 16-row groups per full tile
  const int groups = ceil_div(ceil_div(ceil_div(n_out, 16), parts), align_groups) * align_groups;   // per part = per XCD
  Conv16Plan plan{ceil_div(groups, gt), 0, groups * 16};
  if (!balance || (gt & 1) || plan.full > wgs_per_cu * cus_per_xcd) return plan;
  const int per_cu = ceil_div(groups, cus_per_xcd);
  int f = ceil_div(per_cu - wgs_per_cu * (gt / 2), gt / 2);
  f = f < 0 ? 0 : (f > wgs_per_cu ? wgs_per_cu : f);
  plan.full = min(f * cus_per_xcd, ceil_div(groups, gt));
  const int rest = groups - plan.full * gt;
  plan.half = rest > 0 ? ceil_div(rest, gt / 2) : 0;
  return plan;
}

// ------------------------------------------------------------------------------------------------------------------
// Unit plan of the one-workgroup-per-CU kernel (isf_spconv_cu.hip).  The rows of a launch are cut into UNITS of whole
// 16-row groups, every unit carrying about the same matrix work -- work of a group = the taps through which at least one
// of its rows has a neighbour = what it issues MFMAs for -- so that each compute unit gets one unit (or r of them) and
// they all finish together, whatever the density of the scene under them.  Host and device walk the same arithmetic
// (tests/test_tile_plan.py): cut u of U0 = first group whose inclusive work prefix exceeds u * total / U0; a unit longer
// than kCuCapGroups groups (what one workgroup's accumulators hold) is split evenly.
static constexpr int kCuCapGroups = 16;       // 256 rows x 256 columns of fp32 accumulators in 8 waves x 128 registers
static constexpr int kCuAvgGroups = 12;       // a launch gets cus * r balanced units with n_groups / (cus * r) <= this

// Two shapes (round 6): cap 16 = one 8-wave workgroup per compute unit (the unit's 256 x 256 accumulator tile is the CU's
// register file); cap 8 = two 4-wave workgroups per compute unit, each over a unit of <= 8 groups (128 rows x 256 columns,
// a wave = 8 groups x 64 columns): the two workgroups of a CU run unsynchronised, one's load-issue phase under the other's
// multiply phase.  `slots` = compute units x workgroups per compute unit.
static constexpr int kCuCapGroups8 = 8;
static constexpr int kCuAvgGroups8 = 6;
__host__ __device__ inline int conv_cu_avg_of_cap(int cap) { return cap == kCuCapGroups8 ? kCuAvgGroups8 : kCuAvgGroups; }
__host__ __device__ inline int conv_cu_balanced_units(int n_groups, int slots, int cap = kCuCapGroups) {
  const int per = slots * conv_cu_avg_of_cap(cap);
  const int r = n_groups > per ? (n_groups + per - 1) / per : 1;
  return slots * r;
}
// upper bound of the final unit count (splitting adds at most one unit per `cap` groups)
__host__ __device__ inline int conv_cu_max_units(int n_groups, int slots, int cap = kCuCapGroups) {
  return conv_cu_balanced_units(n_groups, slots, cap) + (n_groups + cap - 1) / cap;
}
// first group of balanced unit u (0 <= u <= U0): W = inclusive prefix of the per-group work
__host__ __device__ inline int conv_cu_cut(const int32_t* W, int n_groups, int U0, int u) {
  if (u <= 0) return 0;
  if (u >= U0) return n_groups;
  const long long target = (long long)u * (long long)W[n_groups - 1] / U0;
  int lo = 0, hi = n_groups;                 // first i with W[i] > target
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((long long)W[mid] > target) hi = mid; else lo = mid + 1;
  }
  return lo;
}
__host__ __device__ inline int conv_cu_pieces(int len, int cap = kCuCapGroups) { return len <= 0 ? 0 : (len + cap - 1) / cap; }
// piece p of a balanced unit of `len` groups starting at group c -> (first group, groups)
__host__ __device__ inline void conv_cu_piece(int c, int len, int p, int& g0, int& ng, int cap = kCuCapGroups) {
  const int P = conv_cu_pieces(len, cap);
  const int a = (int)((long long)p * len / P), b = (int)((long long)(p + 1) * len / P);
  g0 = c + a;
  ng = b - a;
}

// Epilogue shared by both kernels: per 16-row group, accumulator (col = lane&15, row = 4*(lane>>4)+t) -> LDS row-major
// (wave-private transpose tile) -> one lane per (row, 8-channel unit): BN fold (incl. the weight scale), residual,
// ReLU, split, store.  The 8 BN scales / shifts of an item are fetched with two 32-byte loads issued together (the
// scalar form compiled to a load-wait pair per channel: 128 dependent round trips per wave and tile; measured
// 4.49 -> 4.09 ms per step over the 21 convs, profiles/r02_call1_knockout_variants.txt).
// tile_l: this wave's 16 x (16*EPN+4) fp32 LDS tile.  All rows of the wave: row0w .. row0w + 16*RG.
template <int NT, int RG>
struct Conv16Epi {
  static constexpr int EPN = NT > 4 ? 4 : NT;            // column tiles per pass (keeps the transpose tile small)
  static constexpr int RS = 16 * EPN + 4;
  static constexpr int wave_bytes = 16 * RS * 4;
};

// F16IO: residual and output rows are plain f16 ([N][C] halves, 2 bytes per element: the "half format" of the f16
// storage mode) instead of split rows: one 16-byte piece per 8-channel unit at row * (C / 8) + unit.
template <int NT, int RG, bool F16IO = false>
__device__ __forceinline__ void conv16_epilogue(const f32x4 (&acc)[RG][NT], float* tile_l, int lane, int row0w,
                                                int col0 /* first output channel of this workgroup */, int cout,
                                                float winv, const float* __restrict__ scale,
                                                const float* __restrict__ shift, const uint4* __restrict__ residual,
                                                uint4* __restrict__ ys, int n_out, int relu, int groups = RG,
                                                const int32_t* __restrict__ rowmap = nullptr /* position -> row */) {
  using E = Conv16Epi<NT, RG>;
  constexpr int EPN = E::EPN, RS = E::RS;
  const int col = lane & 15, kg = lane >> 4;
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) {
    if (rg >= groups) continue;                        // half tiles / short units: the upper row groups do not exist
                                                       // (continue, not break: a break keeps hipcc from unrolling RG = 16
                                                       // and the accumulators would be indexed through scratch memory)
#pragma unroll
    for (int ps = 0; ps < NT / EPN; ++ps) {
      constexpr int UNITS = (16 * EPN) / 8;           // 8-channel units per row in this pass
      constexpr int ITEMS = (16 * UNITS + 63) / 64;   // (row, unit) items per lane
      // the residual rows of this pass are requested before the LDS transpose, so their latency hides behind it
      uint4 res_hi[ITEMS], res_lo[ITEMS];
      if (residual) {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
          const int i = lane + 64 * it;
          const int grow = row0w + rg * 16 + i / UNITS;
          res_hi[it] = make_uint4(0, 0, 0, 0);
          res_lo[it] = make_uint4(0, 0, 0, 0);
          if (i < 16 * UNITS && grow < n_out) {
            const int unit = (col0 + ps * (16 * EPN)) / 8 + i % UNITS;
            const int gm = rowmap ? rowmap[grow] : grow;      // the row this position of the (sorted) launch computes
            if (F16IO) {
              res_hi[it] = residual[(size_t)gm * (cout >> 3) + unit];
            } else {
              const size_t o = split_hi_index((size_t)gm, cout >> 3, unit);
              res_hi[it] = residual[o];
              res_lo[it] = residual[o + 4];
            }
          }
        }
      }
#pragma unroll
      for (int nt = 0; nt < EPN; ++nt)
#pragma unroll
        for (int t = 0; t < 4; ++t) tile_l[(4 * kg + t) * RS + nt * 16 + col] = acc[rg][ps * EPN + nt][t];
      // wave-private tile: a wave-level fence is enough (LDS ops of one wave complete in order)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      for (int it = 0; it < ITEMS; ++it) {
        const int i = lane + 64 * it;
        const int r = i / UNITS, u = i % UNITS;
        const int grow = row0w + rg * 16 + r;
        if (i < 16 * UNITS && grow < n_out) {
          const float* tp = tile_l + r * RS + u * 8;
          f32x8 v;
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = tp[j];
          const int gc = col0 + ps * (16 * EPN) + u * 8;
          f32x8 sc8, sh8;
#pragma unroll
          for (int j = 0; j < 8; ++j) { sc8[j] = 1.f; sh8[j] = 0.f; }
          if (scale) sc8 = *reinterpret_cast<const f32x8*>(scale + gc);
          if (shift) sh8 = *reinterpret_cast<const f32x8*>(shift + gc);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], scale ? sc8[j] * winv : winv, sh8[j]);
          if (residual) v += join8(res_hi[it], res_lo[it]);   // F16IO: res_lo is zero
          if (relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
          }
          uint4 hi, lo;
          split8(v, hi, lo);
          const int gm = rowmap ? rowmap[grow] : grow;
          if (F16IO) {
            ys[(size_t)gm * (cout >> 3) + (gc >> 3)] = hi;   // round-to-nearest f16 of the fp32 result
          } else {
            const size_t o = split_hi_index((size_t)gm, cout >> 3, gc >> 3);
            ys[o] = hi;
            ys[o + 4] = lo;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
  }
}

}  // namespace isf
