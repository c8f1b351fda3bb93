// isf_spconv_stage.hip -- the f16x3 sparse convolution with the tile's input rows STAGED IN LDS.
//
// isf_spconv16.hip gathers every A fragment global -> VGPR in the MFMA operand layout: a quad of lanes holds four
// different rows, the CU's address unit pays a cycle per distinct cache line of a quad, and a row is fetched once per
// (output row, tap) that uses it -- 3 to 6 times per tile (tools/tile_stats.py).  Here the DISTINCT input rows of a
// tile are listed once per rulebook ("staging tables", below), copied global -> LDS with row-coalesced LDS-DMA (eight
// lanes per 128-byte split-format chunk: one cache line per quad), and the 27 taps read their A fragments from LDS
// through a slot table; the reference's gather stage is the same idea with an HBM staging buffer per tap
// (mmdet3d/ops/bevfusion-ops/spconv/include/spconv/reordering.cu.h:21-97, spconv_ops.h:300-345).
// Arithmetic, accumulation order, weight ring and epilogue are those of spconv_f16x3_kernel: the output is
// bit-identical to it (tests/test_gpu_parity.py::test_staged_conv_*).
//
// Staging tables of a rulebook (rb_unit_tables_kernel; one "unit" = 64 consecutive output rows, kUnitRows):
//   ulist  [units][kUnitCap]  the distinct input rows the unit's 27 x 64 table entries name, ascending
//   ucount [units]            how many
//   slots  [K][nbr_stride]    uint16: position of nbr[k][o] in the list of unit o / 64, 0xFFFF = no neighbour
// A tile is 1, 2 or 4 whole units; each unit gets an equal share of the tile's LDS rows, list entries beyond the share
// are gathered from global memory as before (slot >= share), so any LDS budget is correct.  The prologue turns the
// tile's slots into one int32 LDS table -- physical LDS slot, or 0x80000000 | row for an entry beyond the share, or
// -1 -- so the main loop costs what the gather kernel's costs per entry: one table read, then one operand read.
// LDS image of slot S, chunk kc: 128 bytes at stage + (kc * cap + S) * 128; the eight 16-byte pieces (4 hi k-groups,
// 4 lo) are stored at piece position p ^ ((S >> 1) & 7): the 16 rows of an MFMA row group then fall on 16 different
// bank quads when their slots are consecutive (ds_read_b128 serves 16 lanes per LDS cycle over 64 banks), and a quad of
// DMA lanes still reads one contiguous 64-byte half of the row's chunk.
#include "isf_spconv16.h"

#include <map>
#include <mutex>

namespace isf {

static constexpr int kUnitRows = 64;
static constexpr int kUnitCap = kMaxTaps * kUnitRows;   // 1728: every table entry of a unit distinct
static constexpr int kMaxStageIters = 16;               // DMA instructions per wave and unit: 16 x 8 rows x 4 waves

template <int NT, int RG, int KCH, int NW>
struct StageSmem {
  static constexpr int TM = 16 * RG * NW;
  static constexpr int tab_bytes = kMaxTaps * TM * 4;                    // per (tap, row): LDS slot | 0x80000000 + row | -1
  static constexpr int bbuf_bytes = 2 * KCH * NT * 2048;
  static constexpr int rowid_bytes = 1280 * 4;                          // rows of the staged slots (cap_rows <= 1280)
  static constexpr int misc_bytes = 256;
  // compile-time offsets only: a run-time-offset LDS pointer makes hipcc emit an illegal aperture compare
  static constexpr int fixed_bytes = tab_bytes + bbuf_bytes + rowid_bytes + misc_bytes;   // the A stage follows
  static constexpr int EPN = NT > 4 ? 4 : NT;
  static constexpr int epi_bytes = NW * 16 * (16 * EPN + 4) * 4;
  static_assert(epi_bytes <= tab_bytes + bbuf_bytes, "the epilogue tile overlays the table and the weight ring");
  static_assert(fixed_bytes % 128 == 0, "the A stage is 128-byte aligned");
  static constexpr size_t bytes(int cap_rows) { return (size_t)fixed_bytes + (size_t)cap_rows * KCH * 128; }
};

// MODE bit 1: single-pass f16 (hi halves only), as in spconv_f16x3_kernel.
template <int CIN, int NT, int RG, int NW, int MODE = 0>
__global__ __launch_bounds__(64 * NW, (NW >= 8 ? 1 : 2)) void spconv_staged_kernel(
    const uint4* __restrict__ xs, const uint16_t* __restrict__ slots, int nbr_stride,
    const int32_t* __restrict__ ulist, const int32_t* __restrict__ ucount, const uint4* __restrict__ wpk,
    const float* __restrict__ w_inv_scale, int K, int cout, const float* __restrict__ scale,
    const float* __restrict__ shift, const uint4* __restrict__ residual, uint4* __restrict__ ys, int n_out, int relu,
    Conv16Plan plan, int cap_rows) {
  constexpr bool HALF = (MODE & 1) != 0;
  constexpr int KCH = Conv16Step<CIN, NT>::KCH;
  using S = StageSmem<NT, RG, KCH, NW>;
  constexpr int NTHR = 64 * NW;
  constexpr int TM = S::TM;
  constexpr int WR = 16 * RG;
  constexpr int NCH = CIN / 32;
  constexpr int NCG = NCH / KCH;
  constexpr int CH8 = CIN / 8;
  constexpr int BN = 16 * NT;
  constexpr int MAXU = TM / kUnitRows;   // units of a full tile
  static_assert(TM % (2 * kUnitRows) == 0, "full and half tiles are whole units");
  extern __shared__ __attribute__((aligned(128))) char smem[];
  int* tab_l = reinterpret_cast<int*>(smem);                                    // [27][TM]
  uint4* bbuf = reinterpret_cast<uint4*>(smem + S::tab_bytes);                  // [2][KCH][NT][2][64]
  int* rowid_l = reinterpret_cast<int*>(smem + S::tab_bytes + S::bbuf_bytes);   // [cap_rows] rows of the staged slots
  int* misc = reinterpret_cast<int*>(smem + S::tab_bytes + S::bbuf_bytes + S::rowid_bytes);
  char* stage = smem + S::fixed_bytes;                                          // [KCH][cap_rows][128 B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kg = lane >> 4;
  const int ncb = cout / BN;
  int cb, row0, row_end;
  bool half_tile;
  if (!conv16_tile_of_block(ncb, plan, TM, n_out, cb, row0, row_end, half_tile)) return;
  const int ntiles_total = cout >> 4;
  const int nu = half_tile ? MAXU / 2 : MAXU;                 // units of this tile
  const int ug0 = row0 / kUnitRows;                           // tiles start on unit boundaries (plan alignment)
  const int capu_lds = (cap_rows / nu) & ~7;                  // LDS rows per unit (DMA instructions cover 8 slots)
  const int capu = capu_lds < kMaxStageIters * 8 * NW ? capu_lds : kMaxStageIters * 8 * NW;
  int cnt[MAXU];                                              // staged rows per unit (wave-uniform)
#pragma unroll
  for (int u = 0; u < MAXU; ++u) {
    int c = 0;
    if (u < nu && row0 + u * kUnitRows < row_end) c = ucount[ug0 + u];   // units past the part's end are another tile's
    c = __builtin_amdgcn_readfirstlane(c);
    cnt[u] = c < capu ? c : capu;
    // rows of the staged slots -> LDS (coalesced): the staging DMA of every chunk group reads them from there
    if (u < nu)
      for (int i = tid; i < cnt[u]; i += NTHR) rowid_l[u * capu + i] = ulist[(size_t)(ug0 + u) * kUnitCap + i];
  }

  // ---- prologue: the tile's table -> LDS, per-wave tap mask.  An entry becomes the physical LDS slot of the row, or
  // (list position beyond the unit's LDS share) the row itself, flagged: the main loop never touches the lists.
  {
    constexpr int NP = kMaxTaps * TM / 2;
    constexpr int NB_IT = (NP + NTHR - 1) / NTHR;
    unsigned tmp[NB_IT];
#pragma unroll
    for (int it = 0; it < NB_IT; ++it) {   // two rows per 32-bit load; all loads first, then the dependent work
      const int i = tid + it * NTHR;
      const int k = i / (TM / 2), r = 2 * (i - k * (TM / 2));
      tmp[it] = 0xFFFFFFFFu;
      const int rw = r % WR;
      const int grow = half_tile ? (rw < WR / 2 ? row0 + (r / WR) * (WR / 2) + rw : row_end) : row0 + r;
      // grow is even and < nbr_stride - 1; rows in [n_out, nbr_stride) hold 0xFFFF
      if (i < K * (TM / 2) && grow < row_end)
        tmp[it] = *reinterpret_cast<const unsigned*>(slots + (size_t)k * nbr_stride + grow);
    }
    // rows of the entries beyond the LDS share: every load is issued (a harmless list head where none is needed)
    // before any is used -- a load under a divergent condition is waited for on the spot, 27 round trips in a row
    int ent[NB_IT][2], far[NB_IT][2];
#pragma unroll
    for (int it = 0; it < NB_IT; ++it) {
      const int i = tid + it * NTHR;
      const int k = i / (TM / 2), r = 2 * (i - k * (TM / 2));
      const int rw = r % WR;
      // unit of the row: tile offset / 64 (a half tile keeps only the first WR / 2 rows of every wave)
      const int u = half_tile ? ((r / WR) * (WR / 2) + (rw < WR / 2 ? rw : 0)) / kUnitRows : r / kUnitRows;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int sl = (int)((tmp[it] >> (16 * h)) & 0xFFFFu);
        const bool is_far = sl != 0xFFFF && sl >= capu;
        ent[it][h] = sl == 0xFFFF ? -1 : (sl < capu ? u * capu + sl : -2);
        far[it][h] = ulist[(size_t)(ug0 + u) * kUnitCap + (is_far ? sl : 0)];
      }
    }
#pragma unroll
    for (int it = 0; it < NB_IT; ++it)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (ent[it][h] == -2) ent[it][h] = (int)(0x80000000u | (unsigned)far[it][h]);
#pragma unroll
    for (int it = 0; it < NB_IT; ++it) {
      const int i = tid + it * NTHR;
      if (i < K * (TM / 2)) {
        tab_l[2 * i] = ent[it][0];
        tab_l[2 * i + 1] = ent[it][1];
      }
    }
  }
  __syncthreads();
  unsigned rgm[RG];
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) rgm[rg] = 0;
  for (int k = 0; k < K; ++k) {
#pragma unroll
    for (int rg = 0; rg < RG; rg += 4) {
      const int r = rg * 16 + lane;
      // rows at or past the end of this tile's share (they belong to the next part or do not exist) never count
      const bool has = r < WR && tab_l[k * TM + wave * WR + r] != -1;
      const unsigned long long m = __ballot(has);
#pragma unroll
      for (int j = 0; j < 4 && rg + j < RG; ++j) rgm[rg + j] |= (((m >> (16 * j)) & 0xffffull) ? 1u : 0u) << k;
    }
  }
  unsigned wmask = 0;
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) {
    rgm[rg] = __builtin_amdgcn_readfirstlane(rgm[rg]);
    wmask |= rgm[rg];
  }
  if (lane == 0) misc[wave] = (int)wmask;
  __syncthreads();
  unsigned wg_mask = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) wg_mask |= (unsigned)misc[w];
  wg_mask = __builtin_amdgcn_readfirstlane(wg_mask);
  const int ntaps = __popc(wg_mask);
  const int nsteps = ntaps * NCG;

  f32x4 acc[RG][NT];
#pragma unroll
  for (int rg = 0; rg < RG; ++rg)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[rg][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  struct Cursor {
    unsigned rem;
    int tap, ch;
  };
  auto advance = [&](Cursor& c) {
    if (c.rem == 0) {
      c.rem = wg_mask;
      ++c.ch;
    }
    c.tap = __ffs(c.rem) - 1;
    c.rem &= c.rem - 1;
  };

  const unsigned stage_addr = __builtin_amdgcn_readfirstlane(lds_addr(stage));
  auto stage_A = [&](int cg) {
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
      if (u >= nu) break;
      constexpr int BATCH = 4;
      for (int sb0 = wave; sb0 * 8 < cnt[u]; sb0 += NW * BATCH) {
        int row[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
          const int s = (sb0 + j * NW) * 8 + (lane >> 3);
          row[j] = s < cnt[u] ? rowid_l[u * capu + s] : -1;
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
          const int sb = sb0 + j * NW;
          if (sb * 8 >= cnt[u]) break;
          const int S = u * capu + sb * 8 + (lane >> 3);
          const int p = (lane & 7) ^ ((S >> 1) & 7);
          if (row[j] >= 0) {
#pragma unroll
            for (int kc = 0; kc < KCH; ++kc) {
              if (HALF && p >= 4) continue;
              const uint4* src = xs + ((size_t)row[j] * CH8 + (cg * KCH + kc) * 4) * 2 + p;
              glds16(src, stage_addr + (unsigned)((kc * cap_rows + u * capu + sb * 8) * 128));
            }
          }
        }
      }
    }
  };
  auto load_A = [&](uint4 (&dst)[RG][KCH][2], int tap, int cg) {
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      if ((rgm[rg] >> tap) & 1u) {
        const int e = tab_l[tap * TM + wave * WR + rg * 16 + col];
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) {
          dst[rg][kc][0] = make_uint4(0, 0, 0, 0);
          dst[rg][kc][1] = make_uint4(0, 0, 0, 0);
        }
        if (e >= 0) {            // staged: LDS slot e
          // explicit LDS (address space 3) loads: with generic pointers hipcc merges this branch and the gather
          // below into ONE flat_load through a selected pointer, and an LDS read through the flat path costs what
          // the gather costs
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          typedef const __attribute__((address_space(3))) u32x4 lds_u4;
          auto ld = [](unsigned a) {
            const u32x4 v = *reinterpret_cast<lds_u4*>((size_t)a);
            return make_uint4(v.x, v.y, v.z, v.w);
          };
          const unsigned off = (unsigned)e * 128u + (unsigned)((kg ^ ((e >> 1) & 7)) << 4);
#pragma unroll
          for (int kc = 0; kc < KCH; ++kc) {
            const unsigned base = stage_addr + (unsigned)(kc * cap_rows) * 128u;
            dst[rg][kc][0] = ld(base + off);
            if (!HALF) dst[rg][kc][1] = ld(base + (off ^ 64u));
          }
        } else if (e != -1) {    // beyond the unit's LDS share: the plain gather of row e & 0x7fffffff
          const int row = e & 0x7fffffff;
#pragma unroll
          for (int kc = 0; kc < KCH; ++kc) {
            const uint4* p = xs + ((size_t)row * CH8 + (cg * KCH + kc) * 4) * 2 + kg;
            dst[rg][kc][0] = p[0];
            if (!HALF) dst[rg][kc][1] = p[4];
          }
        }
      }
    }
  };
  const unsigned bbuf_addr = __builtin_amdgcn_readfirstlane(lds_addr(bbuf));
  auto stage_B = [&](int tap, int cg, int buf) {
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc) {
      const uint4* src = wpk + (((size_t)tap * NCH + cg * KCH + kc) * ntiles_total + cb * NT) * 128;
      const unsigned dst = bbuf_addr + (unsigned)((buf * KCH + kc) * (NT * 128)) * 16u;
#pragma unroll
      for (int i = 0; i < (NT * 128 + NTHR - 1) / NTHR; ++i) {
        const int base = i * NTHR + wave * 64;
        if (base < NT * 128 && !(HALF && ((base >> 6) & 1))) glds16(src + base + lane, dst + (unsigned)base * 16u);
      }
    }
  };

  uint4 a_nxt[RG][KCH][2];
  uint4 a_cur[RG][KCH][2];
  Cursor cur{0u, -1, -1};
  bool fresh = true;   // the stage of the current chunk group was requested one step ago: read A after the barrier
  if (nsteps > 0) {
    advance(cur);
    stage_A(cur.ch);
    stage_B(cur.tap, cur.ch, 0);
  }
  for (int s = 0; s < nsteps; ++s) {
    const int tap = cur.tap, ch = cur.ch;
    if (!fresh) {
#pragma unroll
      for (int rg = 0; rg < RG; ++rg)
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) { a_cur[rg][kc][0] = a_nxt[rg][kc][0]; a_cur[rg][kc][1] = a_nxt[rg][kc][1]; }
    }
    // vmcnt(0): this wave's share of B(s), of the stage and its fall-back gathers have landed; lgkmcnt(0): its LDS
    // reads of the stage are done, so after the barrier the stage may be overwritten
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();
    if (fresh) {   // the stage of this chunk group has just landed: read this step's A now, and say so to the compiler
      load_A(a_cur, tap, ch);
      __builtin_amdgcn_s_waitcnt(0x0070);   // (nothing else is in flight yet; no wait is then needed in front of the MFMAs)
    }
    bool next_fresh = false;
    if (s + 1 < nsteps) {
      advance(cur);
      if (cur.ch == ch) load_A(a_nxt, cur.tap, ch);
      else next_fresh = true;   // a new chunk group: its rows are requested after this step's MFMAs (below)
      stage_B(cur.tap, cur.ch, (s + 1) & 1);
    }
    if ((wmask >> tap) & 1u) {
      const uint4* b = bbuf + (s & 1) * (KCH * NT * 128) + lane;
      bool need[RG];
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) need[rg] = (rgm[rg] >> tap) & 1u;
      uint4 bhu_n = b[0], blu_n = make_uint4(0, 0, 0, 0);
      if (!HALF) blu_n = b[64];
#pragma unroll
      for (int i = 0; i < KCH * NT; ++i) {
        const int kc = i / NT, nt = i % NT;
        const uint4 bhu = bhu_n, blu = blu_n;
        if (i + 1 < KCH * NT) {
          bhu_n = b[((i + 1) * 2 + 0) * 64];
          if (!HALF) blu_n = b[((i + 1) * 2 + 1) * 64];
        }
        const h8 bh = *reinterpret_cast<const h8*>(&bhu);
        const h8 bl = *reinterpret_cast<const h8*>(&blu);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
          if (need[rg]) {
            const h8 ah = *reinterpret_cast<const h8*>(&a_cur[rg][kc][0]);
            const h8 al = *reinterpret_cast<const h8*>(&a_cur[rg][kc][1]);
            if (!HALF) {
              acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[rg][nt], 0, 0, 0);
              acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[rg][nt], 0, 0, 0);
            }
            acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[rg][nt], 0, 0, 0);
          }
        }
      }
    }
    if (next_fresh) {
      // hipcc drains vmcnt around a loop that issues the staging DMA; here, behind the MFMAs, that wait only covers
      // the weight stage requested before them.  After this step's barrier nobody reads the stage any more -- except
      // in a one-step chunk group, whose A was read after the barrier: everyone must be past that read.
      if (fresh) {
        __builtin_amdgcn_s_waitcnt(0x0070);
        __syncthreads();
      }
      stage_A(cur.ch);
    }
    fresh = next_fresh;
  }
  __builtin_amdgcn_s_waitcnt(0x0070);
  __syncthreads();

  float* tile_l = reinterpret_cast<float*>(smem) + wave * (Conv16Epi<NT, RG>::wave_bytes / 4);
  conv16_epilogue<NT, RG>(acc, tile_l, lane, row0 + wave * (half_tile ? WR / 2 : WR), cb * BN, cout, *w_inv_scale, scale,
                          shift, residual, ys, row_end, relu, half_tile ? RG / 2 : RG);
}

// ------------------------------------------------------------------------------------------ staging tables
// One workgroup per 64-row unit: the unit's K x 64 table entries are sorted as (input row, position) keys in LDS
// (bitonic), runs of equal rows are numbered, and every entry learns the number of its run.
__global__ __launch_bounds__(256) void rb_unit_tables_kernel(const int32_t* __restrict__ nbr, int nbr_stride, int K,
                                                             uint16_t* __restrict__ slots,
                                                             int32_t* __restrict__ ulist,
                                                             int32_t* __restrict__ ucount) {
  __shared__ unsigned long long key[2048];
  __shared__ int wsum[4];
  const int unit = blockIdx.x, t = threadIdx.x;
  const int n_real = K * kUnitRows;
  int n2 = 256;                                   // sort size: power of two >= n_real (>= 256: one element per thread)
  while (n2 < n_real) n2 <<= 1;
  for (int i = t; i < n2; i += 256) {
    unsigned long long kv = ~0ull;                // padding sorts last
    if (i < n_real) {
      const int k = i >> 6, r = i & 63;
      const uint32_t v = (uint32_t)nbr[(size_t)k * nbr_stride + unit * kUnitRows + r];   // -1 -> 0xFFFFFFFF: after every row
      kv = ((unsigned long long)v << 11) | (unsigned)i;
    }
    key[i] = kv;
  }
  __syncthreads();
  for (int kk = 2; kk <= n2; kk <<= 1) {
    for (int j = kk >> 1; j > 0; j >>= 1) {
      for (int p = t; p < n2 / 2; p += 256) {
        const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
        const int l = i | j;
        const unsigned long long a = key[i], b = key[l];
        const bool up = (i & kk) == 0;
        if ((a > b) == up) { key[i] = b; key[l] = a; }
      }
      __syncthreads();
    }
  }
  // every thread owns n2 / 256 consecutive sorted elements
  const int per = n2 / 256;
  const int i0 = t * per;
  int heads = 0;
  for (int e = 0; e < per; ++e) {
    const int i = i0 + e;
    const unsigned long long kv = key[i];
    const uint32_t v = (uint32_t)(kv >> 11);
    const bool valid = kv != ~0ull && v != 0xFFFFFFFFu;
    const bool head = valid && (i == 0 || (uint32_t)(key[i - 1] >> 11) != v);
    heads += head;
  }
  // exclusive scan of `heads` over the 256 threads
  int incl = heads;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d, 64);
    if ((t & 63) >= d) incl += o;
  }
  if ((t & 63) == 63) wsum[t >> 6] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < (t >> 6); ++w) base += wsum[w];
  int run = base + incl - heads;                  // heads before this thread's first element
  for (int e = 0; e < per; ++e) {
    const int i = i0 + e;
    const unsigned long long kv = key[i];
    if (kv == ~0ull) continue;
    const uint32_t v = (uint32_t)(kv >> 11);
    const int pos = (int)(kv & 2047u);
    const size_t dst = (size_t)(pos >> 6) * nbr_stride + unit * kUnitRows + (pos & 63);
    if (v == 0xFFFFFFFFu) {
      slots[dst] = 0xFFFFu;
      continue;
    }
    const bool head = i == 0 || (uint32_t)(key[i - 1] >> 11) != v;
    if (head) {
      ulist[(size_t)unit * kUnitCap + run] = (int32_t)v;
      ++run;
    }
    slots[dst] = (uint16_t)(run - 1);
  }
  if (t == 255) ucount[unit] = run;
}

int stage_tables_impl(const int32_t* nbr, int nbr_stride, int K, uint16_t* slots, int32_t* ulist, int32_t* ucount,
                      hipStream_t st) {
  ISF_REQUIRE(K >= 1 && K <= kMaxTaps && nbr_stride > 0 && nbr_stride % 128 == 0, ISF_ERR_ARG,
              "stage_tables: %d taps, stride %d", K, nbr_stride);
  hipLaunchKernelGGL(rb_unit_tables_kernel, dim3(nbr_stride / kUnitRows), dim3(256), 0, st, nbr, nbr_stride, K, slots,
                     ulist, ucount);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

// ------------------------------------------------------------------------------------------ launch
template <int CIN, int NT, int RG, int NW, int MODE>
static int launch_staged(bool balance, int cap_rows, const uint4* xs, const uint4* wpk, const float* winv, int K,
                         int cout, const uint16_t* slots, int nbr_stride, const int32_t* ulist, const int32_t* ucount,
                         int n_out, const float* scale, const float* shift, const uint4* residual, int relu, uint4* ys,
                         hipStream_t st) {
  using S = StageSmem<NT, RG, Conv16Step<CIN, NT>::KCH, NW>;
  constexpr int KCH = Conv16Step<CIN, NT>::KCH;
  auto kern = spconv_staged_kernel<CIN, NT, RG, NW, MODE>;
  constexpr int max_lds = 160 * 1024;
  int max_cap = (int)((max_lds - S::fixed_bytes) / (KCH * 128)) & ~31;
  if (max_cap > 1280) max_cap = 1280;
  if (cap_rows > max_cap) cap_rows = max_cap;
  cap_rows &= ~31;                                   // up to 4 units, each a multiple of 8 rows
  ISF_REQUIRE(cap_rows >= 32, ISF_ERR_ARG, "sparse_conv_staged: %d LDS rows (>= 32)", cap_rows);
  const size_t bytes = S::bytes(cap_rows);
  // occupancy of this instantiation at this LDS size (the tile plan deals workgroups per CU)
  static std::mutex mu;
  static std::map<int, int> occ_of_cap;
  static int cus_per_xcd = 0;
  int wgs_per_cu = 0;
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = occ_of_cap.find(cap_rows);
    if (it == occ_of_cap.end()) {
      ISF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      max_lds));
      int dev = 0, cus = 0, occ = 0;
      ISF_HIP_TRY(hipGetDevice(&dev));
      ISF_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      ISF_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64 * NW, bytes));
      cus_per_xcd = cus >= 8 ? cus / 8 : 1;
      it = occ_of_cap.emplace(cap_rows, occ > 0 ? occ : 1).first;
    }
    wgs_per_cu = it->second;
  }
  const int ncb = cout / (16 * NT);
  ISF_REQUIRE(ncb == 1 || ncb == 2, ISF_ERR_UNSUPPORTED, "sparse_conv_staged: %d column blocks", ncb);
  const Conv16Plan plan = conv16_plan(n_out, S::TM, ncb, wgs_per_cu, cus_per_xcd, balance, kUnitRows / 16);
  hipLaunchKernelGGL(kern, dim3(conv16_grid_blocks(plan)), dim3(64 * NW), bytes, st, xs, slots, nbr_stride, ulist,
                     ucount, wpk, winv, K, cout, scale, shift, residual, ys, n_out, relu, plan, cap_rows);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

template <int CIN, int NT>
static int launch_staged_rows(int mode, int cap_rows, const uint4* xs, const uint4* wpk, const float* winv, int K,
                              int cout, const uint16_t* slots, int nbr_stride, const int32_t* ulist,
                              const int32_t* ucount, int n_out, const float* scale, const float* shift,
                              const uint4* residual, int relu, uint4* ys, hipStream_t st) {
#define ISF_ARGS_ST (mode & 32) == 0, cap_rows, xs, wpk, winv, K, cout, slots, nbr_stride, ulist, ucount, n_out, scale, shift, residual, relu, ys, st
  const bool wide = NT == 8 && cout == 128 && n_out >= 8 * 256;   // the 8-wave shape of launch16_rows
  if ((mode & ~32) == 1) {
    if (wide) return launch_staged<CIN, (NT == 8 ? NT : 2), 2, 8, 1>(ISF_ARGS_ST);
    return launch_staged<CIN, NT, 2, 4, 1>(ISF_ARGS_ST);
  }
  if (wide) return launch_staged<CIN, (NT == 8 ? NT : 2), 2, 8, 0>(ISF_ARGS_ST);
  return launch_staged<CIN, NT, 2, 4, 0>(ISF_ARGS_ST);
#undef ISF_ARGS_ST
}

template <int CIN>
static int dispatch_staged(int mode, int cap_rows, const uint4* xs, const uint4* wpk, const float* winv, int K,
                           int cout, const uint16_t* slots, int nbr_stride, const int32_t* ulist,
                           const int32_t* ucount, int n_out, const float* scale, const float* shift,
                           const uint4* residual, int relu, uint4* ys, hipStream_t st) {
  switch (cout) {
    case 32:  return launch_staged_rows<CIN, 2>(mode, cap_rows, xs, wpk, winv, K, cout, slots, nbr_stride, ulist, ucount, n_out, scale, shift, residual, relu, ys, st);
    case 64:  return launch_staged_rows<CIN, 4>(mode, cap_rows, xs, wpk, winv, K, cout, slots, nbr_stride, ulist, ucount, n_out, scale, shift, residual, relu, ys, st);
    case 128:
    case 256: return launch_staged_rows<CIN, 8>(mode, cap_rows, xs, wpk, winv, K, cout, slots, nbr_stride, ulist, ucount, n_out, scale, shift, residual, relu, ys, st);
  }
  return ISF_ERR_UNSUPPORTED;
}

int sparse_conv_forward_staged_impl(const void* xs, int c_in, const void* packed16, int K, int c_out,
                                    const uint16_t* slots, int nbr_stride, const int32_t* ulist,
                                    const int32_t* ucount, int n_out, const float* scale, const float* shift,
                                    const void* residual, int relu, void* ys, int stage_rows, int mode,
                                    hipStream_t st) {
  if (n_out <= 0) return ISF_OK;
  ISF_REQUIRE(K >= 1 && K <= kMaxTaps, ISF_ERR_UNSUPPORTED, "sparse_conv_staged: %d taps (max 27)", K);
  ISF_REQUIRE(sparse_conv_f16x3_supported(c_in, c_out), ISF_ERR_UNSUPPORTED,
              "sparse_conv_staged: (Cin,Cout)=(%d,%d) not built", c_in, c_out);
  ISF_REQUIRE(nbr_stride % 128 == 0 && nbr_stride >= n_out, ISF_ERR_ARG, "sparse_conv_staged: bad nbr_stride");
  const int m = mode & ~32;
  ISF_REQUIRE(m == 0 || m == 1, ISF_ERR_ARG, "sparse_conv_staged: mode %d (0 split precision, 1 single-pass f16, +32)", mode);
  const uint4* w = reinterpret_cast<const uint4*>(packed16);
  const float* winv = reinterpret_cast<const float*>(reinterpret_cast<const char*>(packed16) +
                                                     (size_t)K * c_in * c_out * 4);
  const uint4* x = reinterpret_cast<const uint4*>(xs);
  const uint4* r = reinterpret_cast<const uint4*>(residual);
  uint4* y = reinterpret_cast<uint4*>(ys);
  switch (c_in) {
    case 32:  return dispatch_staged<32>(mode, stage_rows, x, w, winv, K, c_out, slots, nbr_stride, ulist, ucount, n_out, scale, shift, r, relu, y, st);
    case 64:  return dispatch_staged<64>(mode, stage_rows, x, w, winv, K, c_out, slots, nbr_stride, ulist, ucount, n_out, scale, shift, r, relu, y, st);
    case 128: return dispatch_staged<128>(mode, stage_rows, x, w, winv, K, c_out, slots, nbr_stride, ulist, ucount, n_out, scale, shift, r, relu, y, st);
    case 256: return dispatch_staged<256>(mode, stage_rows, x, w, winv, K, c_out, slots, nbr_stride, ulist, ucount, n_out, scale, shift, r, relu, y, st);
  }
  return ISF_ERR_UNSUPPORTED;
}

}  // namespace isf

extern "C" {

int isf_stage_unit_rows(void) { return isf::kUnitRows; }
int isf_stage_unit_cap(void) { return isf::kUnitCap; }

int isf_rulebook_stage_tables(const int32_t* nbr, int nbr_stride, int num_taps, uint16_t* slots, int32_t* ulist,
                              int32_t* ucount, isf_stream_t stream) {
  ISF_REQUIRE(nbr && slots && ulist && ucount, ISF_ERR_ARG, "rulebook_stage_tables: null pointer");
  return isf::stage_tables_impl(nbr, nbr_stride, num_taps, slots, ulist, ucount, isf::as_stream(stream));
}

int isf_sparse_conv_forward_staged(const void* features_split, int num_in, int c_in, const void* packed16,
                                   int num_taps, int c_out, const uint16_t* slots, int nbr_stride,
                                   const int32_t* ulist, const int32_t* ucount, int num_out, const float* scale,
                                   const float* shift, const void* residual_split, int relu, void* out_split,
                                   int stage_rows, int mode, isf_stream_t stream) {
  ISF_REQUIRE(num_in >= 0 && num_out >= 0 && c_in > 0 && c_out > 0 && num_taps > 0, ISF_ERR_ARG,
              "sparse_conv_forward_staged: bad arguments");
  if (num_out == 0) return ISF_OK;
  ISF_REQUIRE(features_split && packed16 && slots && ulist && ucount && out_split &&
                  ((scale == nullptr) == (shift == nullptr)),
              ISF_ERR_ARG, "sparse_conv_forward_staged: null pointer");
  return isf::sparse_conv_forward_staged_impl(features_split, c_in, packed16, num_taps, c_out, slots, nbr_stride, ulist,
                                              ucount, num_out, scale, shift, residual_split, relu, out_split,
                                              stage_rows, mode, isf::as_stream(stream));
}

}  // extern "C"
