// isf_spconv_dma.hip -- the f16x3 sparse convolution (isf_spconv16.hip) for the NARROW layers (<= 64 input and output
// channels: levels 0 / 1, 350 k rows each) with the gathered input rows brought in by LDS-DMA in the layout the address
// unit likes.
//
// Why.  The CU's address unit walks a 64-lane memory instruction quad by quad (4 consecutive lanes) and pays one cycle
// per distinct cache line a quad touches (tools/probes/masked_gather.hip, profiles/r02_call16_gather_probe.txt).  The
// gather of isf_spconv16.hip loads straight into the MFMA A-operand layout -- lane = (row = lane & 15, k-group =
// lane >> 4) -- so every quad is four different rows = four lines: 64 cycles per instruction against 16 for a load whose
// quads stay inside one line.  On the narrow layers a step has few MFMAs per gathered row (64 -> 64: 48 MFMAs for 8
// gather instructions per wave; 32 -> 32: 12 for 4), so the address unit, not the matrix pipe, sets the step time:
// 12 waves x 8 x 64 cycles = 3 us of address-unit time per round of steps on a CU against 1.1 us of MFMAs (measured 2.3 us
// with the neighbour sharing of isf_spconv16.hip; knock-out without gathers: -39 % / -35 % / -47 % on 64 -> 64 / 32 -> 32 /
// 64 -> 32, profiles/r02_call1_knockout_variants.txt).
// Here a gather instruction's lane l fetches (row = l >> 2, 16-byte piece (l & 3) rotated by row >> 2): one quad = the
// 64 contiguous bytes of ONE row.  It cannot land in registers that way (the MFMA wants a row's four pieces in four
// different 16-lane groups), so it goes global -> LDS with global_load_lds_dwordx4 -- no VGPRs, no VALU, asynchronous --
// into a wave-private 4 KiB transit buffer, and the wave reads its A fragments from there with four ds_read_b128 in
// the MFMA layout (the rotation makes the 16 lanes of a k-group hit 16 different bank groups).  Round 2 tried the same
// load pattern with ds_bpermute_b32 to fix the layout in registers and lost to the 16 permutes per step
// (profiles/r02_call16_gather_probe.txt); the transit costs 4 LDS reads.  The same kernel on the WIDE layers (>= 128
// output columns) lost 2-9 %: their busiest CUs are bound by the matrix pipe (profiles/r03_conv_trace.txt) -- they stay
// on isf_spconv16.hip.
// Other differences from isf_spconv16.hip:
//   * one 32-channel chunk per step, taps outer / chunks inner (the same order of products per accumulator as the
//     gather kernel's two-chunk steps: results are BIT-IDENTICAL), a lane's neighbour index lives in a register and is
//     fetched one tap ahead: no neighbour table in LDS, 32 KiB (64 output columns) / 24 KiB (32) per workgroup =
//     5 / 6 workgroups per CU instead of 3;
//   * rows without a neighbour read a zero line instead of being masked (the DMA writes every lane's 16 bytes); no
//     neighbour sharing (every row is fetched, at a quarter of the cost).
#include "isf_spconv16.h"

#include <atomic>

namespace isf {

// Probe builds only (tools/probes/build_dma_knockouts.sh compiles this file with -DISF_DMA_KNOCKOUT=<bits> into side
// libraries; the shipped library is built with 0 and contains none of it): leave a part of the step out to see what the
// step's time is made of.  1 = no row gathers, 2 = no weight staging, 4 = no MFMAs (nor their LDS reads), 8 = no
// per-step barrier.  Results are garbage.
#ifndef ISF_DMA_KNOCKOUT
#define ISF_DMA_KNOCKOUT 0
#endif
#ifndef ISF_DMA_EARLY_B
#define ISF_DMA_EARLY_B 1
#endif
__device__ uint4 g_zero_line[8];   // 128 zero bytes: what a row without a neighbour reads

constexpr int kDmaTraceWords = 16;
__device__ __forceinline__ unsigned long long dma_shader_clock() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  return t;
}

template <int NT, int NW, int RG = 2>
struct ConvDmaSmem {
  static constexpr int TM = 16 * RG * NW;
  static constexpr int bbuf_bytes = 2 * NT * 2048;        // double-buffered weight stage
  static constexpr int transit_bytes = NW * RG * 2048;    // per wave: [RG row groups][hi, lo][64 pieces of 16 B]
  static constexpr int epi_bytes = NW * Conv16Epi<NT, RG>::wave_bytes;
  static constexpr int main_bytes = bbuf_bytes + transit_bytes;
  static constexpr int bytes = main_bytes > epi_bytes ? main_bytes : epi_bytes;   // wave masks overlay the weight stage
};

// MODE: bit 1 = single-pass f16 (hi halves only), bit 256 (with 1) = f16 storage -- as in spconv_f16x3_kernel
// LINES: the neighbour table comes LINE-COMPRESSED (isf_rulebook.hip): `nbr` = lines [K / nx][nbr_stride] (row of the
// first present neighbour of a (kz, ky) line), `lmask` [nbr_stride] (bit k: tap k present); a lane keeps its rows' masks
// in registers and loads one int32 per LINE (one line ahead) instead of one per tap; the prologue reads one word per row
// instead of 27.  Same indices, same products: bit-identical.
template <int CIN, int NT, int NW, int MODE, int RG = 2, bool LINES = false>
__global__ __launch_bounds__(64 * NW) void spconv_dma_kernel(
    const uint4* __restrict__ xs, const int32_t* __restrict__ nbr, const uint32_t* __restrict__ lmask, int nx,
    int nbr_stride, const uint4* __restrict__ wpk,
    const float* __restrict__ w_inv_scale, int K, int cout, const float* __restrict__ scale,
    const float* __restrict__ shift, const uint4* __restrict__ residual, uint4* __restrict__ ys, int n_out, int relu,
    Conv16Plan plan, const int32_t* __restrict__ order,
    const int32_t* __restrict__ rowmap /* nullptr | sorted launch: position -> output row (conv_row_sort_impl) */,
    long long* __restrict__ trace /* MODE bit 512: kDmaTraceWords int64 per workgroup (isf_sparse_conv_dma_trace) */) {
  constexpr bool HALF = (MODE & 1) != 0, F16IO = (MODE & 256) != 0, TRACE = (MODE & 512) != 0;
  // TRACE (diagnostic instantiations only): constant-clock stamps at entry / after the prologue / after the loop / at
  // exit, and wave 0's shader-clock account of the loop: cycles at the per-step vmcnt(0), at the barrier, in the section
  // that reads the transit / weights and issues the next step's loads, in the multiply section
  long long t_entry = 0, t_pro = 0, t_loop = 0;
  unsigned long long c_wait = 0, c_bar = 0, c_issue = 0, c_mul = 0, c_rd = 0, c_adv = 0;
  if (TRACE) t_entry = wall_clock64();
  static_assert(!F16IO || HALF, "f16 storage implies single-pass f16 arithmetic");
  using S = ConvDmaSmem<NT, NW, RG>;
  constexpr int NTHR = 64 * NW, TM = S::TM, WR = 16 * RG;
  constexpr int NCH = CIN / 32, CH8 = CIN / 8, BN = 16 * NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* bbuf = reinterpret_cast<uint4*>(smem);                                   // [2][NT][2][64]
  int* misc = reinterpret_cast<int*>(smem);                                       // [NW] wave masks (prologue only)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kg = lane >> 4;
  const int ncb = cout / BN;
  int cb, row0, row_end;
  bool half_tile;
  if (!conv16_tile_of_block(ncb, plan, TM, n_out, cb, row0, row_end, half_tile, order)) return;
  const int ntiles_total = cout >> 4;
  const int row0w = row0 + wave * (half_tile ? WR / 2 : WR);      // first row of this wave
  const int wrows = half_tile ? WR / 2 : WR;

  // ---- prologue: which taps does each 16-row group of the wave use (bit k of rgm[rg]); no table is kept
  unsigned rgm[RG];
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) rgm[rg] = 0u;
  if constexpr (LINES) {
    const int row = row0w + lane;
    const unsigned mrow = (lane < wrows && row < row_end) ? lmask[row] : 0u;
#pragma unroll
    for (int k = 0; k < kMaxTaps; ++k) {
      const unsigned long long m = __ballot((mrow >> k) & 1u);
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) rgm[rg] |= (((m >> (16 * rg)) & 0xffffull) ? 1u : 0u) << k;
    }
  } else {
    int tmp[kMaxTaps];
    const int row = row0w + lane;
    const bool live = lane < wrows && row < row_end;
#pragma unroll
    for (int k = 0; k < kMaxTaps; ++k) {
      tmp[k] = -1;
      if (k < K && live) tmp[k] = nbr[(size_t)k * nbr_stride + row];
    }
#pragma unroll
    for (int k = 0; k < kMaxTaps; ++k) {
      const unsigned long long m = __ballot(tmp[k] >= 0);
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) rgm[rg] |= (((m >> (16 * rg)) & 0xffffull) ? 1u : 0u) << k;
    }
  }
  unsigned wmask = 0u;
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
  __syncthreads();   // the masks have been read: the weight stage may overwrite them
  const int nsteps = __popc(wg_mask) * NCH;

  f32x4 acc[RG][NT];
#pragma unroll
  for (int rg = 0; rg < RG; ++rg)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[rg][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // gather lane: row (lane >> 2) of a 16-row group, piece rotated so that the MFMA-layout read is conflict-free:
  // LDS position 4 r + t holds piece (t - (r >> 2)) & 3 of row r; the reader (row col, k-group kg) finds its piece at
  // position 4 col + ((kg + (col >> 2)) & 3)
  const int grow_l = lane >> 2;                                   // this lane's row within a group (gather side)
  const int gpiece = ((lane & 3) - (grow_l >> 2)) & 3;
  const int rpos = 4 * col + ((kg + (col >> 2)) & 3);             // read side
  const unsigned bbuf_addr = __builtin_amdgcn_readfirstlane(lds_addr(bbuf));
  const uint4* transit = reinterpret_cast<const uint4*>(smem + S::bbuf_bytes) + wave * (RG * 128);
  const unsigned transit_addr = __builtin_amdgcn_readfirstlane(lds_addr(transit));
  const uint4* zero = g_zero_line;

  // neighbour index of this lane's gather row through a tap, per row group (-1: none)
  auto load_idx = [&](int tap, int (&idx)[RG]) {
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      const int row = row0w + rg * 16 + grow_l;
      idx[rg] = -1;
      if (((rgm[rg] >> tap) & 1u) && row < row_end) idx[rg] = nbr[(size_t)tap * nbr_stride + row];
    }
  };
  // LINES: this lane's gather rows' tap masks, the first-neighbour row of the current / the next needed line
  unsigned mg[RG];
  int base_cur[RG], base_nxt[RG];
  int line_cur = -1;
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) {
    mg[rg] = 0u;
    base_cur[rg] = base_nxt[rg] = -1;
    if constexpr (LINES) {
      const int row = row0w + rg * 16 + grow_l;
      if (rgm[rg] && row < row_end) mg[rg] = lmask[row];
    }
  }
  auto line_of = [&](int tap) -> int { return nx == 3 ? (tap * 43) >> 7 : tap; };   // tap / 3 for tap < 27
  auto load_line = [&](int line, int (&base)[RG]) {
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      const int row = row0w + rg * 16 + grow_l;
      const unsigned lbits = ((nx == 3 ? 7u : 1u) << (line * nx));
      base[rg] = -1;
      if ((mg[rg] & lbits) && row < row_end) base[rg] = nbr[(size_t)line * nbr_stride + row];
    }
  };
  auto line_idx = [&](int tap, int (&idx)[RG]) {     // indices through `tap` from base_cur (its line) and the masks
    const unsigned first = (unsigned)(line_of(tap) * nx);
    const unsigned below = ((1u << tap) - 1u) & ~((1u << first) - 1u);
#pragma unroll
    for (int rg = 0; rg < RG; ++rg)
      idx[rg] = ((mg[rg] >> tap) & 1u) ? base_cur[rg] + __popc(mg[rg] & below) : -1;
  };
  auto issue_A = [&](int tap, int ch, const int (&idx)[RG]) {
    if constexpr ((ISF_DMA_KNOCKOUT & 1) != 0) return;
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      if ((rgm[rg] >> tap) & 1u) {                                  // wave-uniform
        const uint4* src = zero + gpiece;
        if (idx[rg] >= 0)
          src = F16IO ? xs + (size_t)idx[rg] * CH8 + ch * 4 + gpiece
                      : xs + ((size_t)idx[rg] * CH8 + ch * 4) * 2 + gpiece;
        glds16(src, transit_addr + (unsigned)(rg * 2) * 1024u);
        if (!HALF) glds16(src + 4, transit_addr + (unsigned)(rg * 2 + 1) * 1024u);   // zero line: 8 pieces long
      }
    }
  };
  constexpr int PW = NT * 128 / NW;     // 16-byte weight pieces per wave and step
  auto stage_B = [&](int tap, int ch, int buf) {
    if constexpr ((ISF_DMA_KNOCKOUT & 2) != 0) return;
    if constexpr (HALF ? (PW == 128 || PW == 256) : (PW == 64 || PW == 128 || PW == 256)) {   // one run per wave (glds16_run)
      const uint4* srun = wpk + (((size_t)tap * NCH + ch) * ntiles_total + cb * NT) * 128 + wave * PW + lane;
      const unsigned drun = bbuf_addr + (unsigned)(buf * (NT * 128) + wave * PW) * 16u;
      if constexpr (HALF) glds16_run_hi<PW / 64>(srun, drun);   // single-pass f16: the hi KiB of each column tile only
      else glds16_run<PW / 64>(srun, drun);
      return;
    }
    const uint4* src = wpk + (((size_t)tap * NCH + ch) * ntiles_total + cb * NT) * 128;
    const unsigned dst = bbuf_addr + (unsigned)(buf * (NT * 128)) * 16u;
#pragma unroll
    for (int i = 0; i < (NT * 128 + NTHR - 1) / NTHR; ++i) {
      const int base = i * NTHR + wave * 64;
      if (base < NT * 128 && !(HALF && ((base >> 6) & 1))) glds16(src + base + lane, dst + (unsigned)base * 16u);
    }
  };

  // step cursor: taps (set bits of wg_mask, increasing) outer, chunks inner
  unsigned rem = wg_mask;
  int tap = -1, ch = NCH - 1;
  int idx_cur[RG], idx_nxt[RG];
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) idx_cur[rg] = idx_nxt[rg] = -1;
  auto advance = [&]() {   // -> the next step's (tap, ch); on a tap change rotates the index registers
    if (++ch == NCH) {
      ch = 0;
      tap = __ffs(rem) - 1;
      rem &= rem - 1;
      if constexpr (LINES) {
        const int line = line_of(tap);
        if (line != line_cur) {              // a new line: its bases were requested one line ago
          line_cur = line;
#pragma unroll
          for (int rg = 0; rg < RG; ++rg) base_cur[rg] = base_nxt[rg];
          const unsigned later = (line + 1) * nx < 32 ? rem & ~((1u << ((line + 1) * nx)) - 1u) : 0u;
          if (later) load_line(line_of(__ffs(later) - 1), base_nxt);
        }
        line_idx(tap, idx_cur);
      } else {
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) idx_cur[rg] = idx_nxt[rg];
        if (rem) load_idx(__ffs(rem) - 1, idx_nxt);
      }
    }
  };
  if (TRACE) t_pro = wall_clock64();
  if (nsteps > 0) {
    if constexpr (LINES) load_line(line_of(__ffs(rem) - 1), base_nxt);
    else load_idx(__ffs(rem) - 1, idx_nxt);
    advance();
    issue_A(tap, ch, idx_cur);
    stage_B(tap, ch, 0);
  }
  for (int s = 0; s < nsteps; ++s) {
    const int tap_s = tap;
    unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    if (TRACE) c0 = dma_shader_clock();
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's A(s) rows and its share of B(s) have landed
    if (TRACE) c1 = dma_shader_clock();
    if constexpr ((ISF_DMA_KNOCKOUT & 8) == 0)
      __syncthreads();                    // B(s) complete for every wave; everyone is done reading buffer (s+1)&1
    if (TRACE) c2 = dma_shader_clock();
    uint4 a_cur[RG][2];
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      a_cur[rg][0] = make_uint4(0, 0, 0, 0);
      a_cur[rg][1] = make_uint4(0, 0, 0, 0);
      if ((rgm[rg] >> tap_s) & 1u) {
        a_cur[rg][0] = transit[(rg * 2) * 64 + rpos];
        if (!HALF) a_cur[rg][1] = transit[(rg * 2 + 1) * 64 + rpos];
      }
    }
    const uint4* b = bbuf + (s & 1) * (NT * 128) + lane;
    uint4 bhu_n = b[0], blu_n = make_uint4(0, 0, 0, 0);
    if (!HALF) blu_n = b[64];
    // the next step's weight run goes out while the transit reads are still in flight (it lands in the other weight
    // buffer); the rows follow once the transit has been read
    const bool more = s + 1 < nsteps;
#if ISF_DMA_EARLY_B
    unsigned long long ca = 0, cb = 0;
    if (TRACE) ca = dma_shader_clock();   // (waits for the fragment reads: their LDS round trip on its own)
    if (more) {
      advance();
      stage_B(tap, ch, (s + 1) & 1);
    }
    if (TRACE) cb = dma_shader_clock();
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the transit has been read, the next rows may overwrite it
    if (more) issue_A(tap, ch, idx_cur);
    if (TRACE) {
      c_rd += ca - c2;
      c_adv += cb - ca;
    }
#else   // probe builds (tools/probes/build_side_lib.sh isf_spconv_dma.hip ISF_DMA_EARLY_B=0): round 4's order, for A/B
    __builtin_amdgcn_s_waitcnt(0xC07F);
    if (more) {
      advance();
      issue_A(tap, ch, idx_cur);
      stage_B(tap, ch, (s + 1) & 1);
    }
#endif
    if (TRACE) {
      c3 = dma_shader_clock();
      c_wait += c1 - c0;
      c_bar += c2 - c1;
      c_issue += c3 - c2;
    }
    if (((wmask >> tap_s) & 1u) && (ISF_DMA_KNOCKOUT & 4) == 0) {
      bool need[RG];
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) need[rg] = (rgm[rg] >> tap_s) & 1u;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const uint4 bhu = bhu_n, blu = blu_n;
        if (nt + 1 < NT) {
          bhu_n = b[((nt + 1) * 2 + 0) * 64];
          if (!HALF) blu_n = b[((nt + 1) * 2 + 1) * 64];
        }
        const h8 bh = *reinterpret_cast<const h8*>(&bhu);
        const h8 bl = *reinterpret_cast<const h8*>(&blu);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
          if (need[rg]) {
            const h8 ah = *reinterpret_cast<const h8*>(&a_cur[rg][0]);
            const h8 al = *reinterpret_cast<const h8*>(&a_cur[rg][1]);
            if (!HALF) {
              acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[rg][nt], 0, 0, 0);
              acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[rg][nt], 0, 0, 0);
            }
            acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[rg][nt], 0, 0, 0);
          }
        }
      }
    }
    if (TRACE) c_mul += dma_shader_clock() - c3;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();  // all waves done with the weight buffers -> reuse as the epilogue transpose tile
  if (TRACE) t_loop = wall_clock64();

  float* tile_l = reinterpret_cast<float*>(smem) + wave * (Conv16Epi<NT, RG>::wave_bytes / 4);
  conv16_epilogue<NT, RG, F16IO>(acc, tile_l, lane, row0w, cb * BN, cout, *w_inv_scale, scale, shift, residual, ys,
                                 row_end, relu, half_tile ? RG / 2 : RG, rowmap);
  if (TRACE) {
    __syncthreads();
    if (tid == 0) {
      unsigned hw_id, xcc_id;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
      long long* t = trace + (size_t)blockIdx.x * kDmaTraceWords;
      t[0] = t_entry; t[1] = t_pro; t[2] = t_loop; t[3] = wall_clock64();
      t[4] = nsteps; t[5] = hw_id; t[6] = xcc_id; t[7] = (long long)row0 | ((long long)half_tile << 32);
      t[8] = (long long)c_wait; t[9] = (long long)c_bar; t[10] = (long long)c_issue; t[11] = (long long)c_mul;
      t[12] = (long long)c_rd; t[13] = (long long)c_adv; t[14] = 0; t[15] = 0;   // of c_issue: fragment reads, index + weight run
    }
  }
}

bool sparse_conv_dma_supported(int c_in, int c_out) {
  return (c_in == 32 || c_in == 64) && (c_out == 32 || c_out == 64);
}

// 4 waves x 32 rows per workgroup.  Measured slower (profiles/r03_dma_gather.txt): 8 waves (one weight stage per 256 rows,
// 3 workgroups per CU) 64 -> 64 0.740 -> 0.768 ms per step; 4 waves x 64 rows (RG = 4: twice the MFMAs per step and
// barrier, 3 workgroups per CU) 0.770 -> 0.865 -- the resident workgroups are what hides the loads' round trip.
template <int CIN, int NT, int MODE, bool LINES, int NW = 4, int RG = 2>
static int launch_dma(bool balance, const uint4* xs, const uint4* wpk, const float* winv, int K, int cout,
                      const int32_t* nbr, const uint32_t* lmask, int nx, int nbr_stride, int n_out, const float* scale,
                      const float* shift, const uint4* residual, int relu, uint4* ys, hipStream_t st,
                      const int32_t* order, Conv16LaunchInfo* query, const int32_t* rowmap, long long* trace) {
  using S = ConvDmaSmem<NT, NW, RG>;
  auto kern = spconv_dma_kernel<CIN, NT, NW, MODE, RG, LINES>;
  static std::atomic<int> wgs_per_cu{0}, cus_per_xcd{0};
  if (wgs_per_cu.load(std::memory_order_acquire) == 0) {
    if (S::bytes > 48 * 1024)
      ISF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      S::bytes));
    int dev = 0, cus = 0, occ = 0;
    ISF_HIP_TRY(hipGetDevice(&dev));
    ISF_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    ISF_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64 * NW, S::bytes));
    cus_per_xcd.store(cus >= 8 ? cus / 8 : 1, std::memory_order_relaxed);
    wgs_per_cu.store(occ > 0 ? occ : 1, std::memory_order_release);
  }
  const int ncb = cout / (16 * NT);
  const Conv16Plan plan = conv16_plan(n_out, S::TM, ncb, wgs_per_cu.load(std::memory_order_relaxed),
                                      cus_per_xcd.load(std::memory_order_relaxed), balance);
  if (query) {
    *query = Conv16LaunchInfo{plan.full, plan.half, plan.part_rows, S::TM, ncb, wgs_per_cu.load(std::memory_order_relaxed),
                              cus_per_xcd.load(std::memory_order_relaxed)};
    return ISF_OK;
  }
  hipLaunchKernelGGL(kern, dim3(conv16_grid_blocks(plan)), dim3(64 * NW), S::bytes, st, xs, nbr, lmask, nx, nbr_stride, wpk,
                     winv, K, cout, scale, shift, residual, ys, n_out, relu, plan, order, rowmap, trace);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

template <int CIN, int NT>
static int dispatch_dma(int mode, const uint4* xs, const uint4* wpk, const float* winv, int K, int cout,
                        const int32_t* nbr, const uint32_t* lmask, int nx, int nbr_stride, int n_out, const float* scale,
                        const float* shift, const uint4* residual, int relu, uint4* ys, hipStream_t st,
                        const int32_t* order, Conv16LaunchInfo* query, const int32_t* rowmap, long long* trace) {
  const bool balance = (mode & 32) == 0;
#define ISF_ARGS_DMA balance, xs, wpk, winv, K, cout, nbr, lmask, nx, nbr_stride, n_out, scale, shift, residual, relu, ys, st, order, query, rowmap, trace
  if (lmask) {
    switch (mode & ~32) {
      case 512: return launch_dma<CIN, NT, 512, true>(ISF_ARGS_DMA);   // trace (isf_sparse_conv_dma_trace)
      case 0: return launch_dma<CIN, NT, 0, true>(ISF_ARGS_DMA);
      case 1: return launch_dma<CIN, NT, 1, true>(ISF_ARGS_DMA);
      case 257: return launch_dma<CIN, NT, 257, true>(ISF_ARGS_DMA);
    }
  } else {
    switch (mode & ~32) {
      case 512: return launch_dma<CIN, NT, 512, false>(ISF_ARGS_DMA);
      case 0: return launch_dma<CIN, NT, 0, false>(ISF_ARGS_DMA);
      case 1: return launch_dma<CIN, NT, 1, false>(ISF_ARGS_DMA);
      case 257: return launch_dma<CIN, NT, 257, false>(ISF_ARGS_DMA);
    }
  }
#undef ISF_ARGS_DMA
  ISF_REQUIRE(false, ISF_ERR_ARG, "sparse_conv_dma: mode %d (0, 1, 257, +32)", mode);
}

int sparse_conv_forward_dma_impl(const void* xs, int c_in, const void* packed16, int K, int c_out, const int32_t* nbr,
                                 int nbr_stride, int n_out, const float* scale, const float* shift,
                                 const void* residual, int relu, void* ys, int mode, hipStream_t st,
                                 const int32_t* order, Conv16LaunchInfo* query, const uint32_t* lmask, int nx,
                                 const int32_t* rowmap, long long* trace) {
  if (n_out <= 0) {
    if (query) *query = Conv16LaunchInfo{0, 0, 0, 0, 0, 0, 0};
    return ISF_OK;
  }
  ISF_REQUIRE(K >= 1 && K <= kMaxTaps, ISF_ERR_UNSUPPORTED, "sparse_conv_dma: %d taps (max 27)", K);
  ISF_REQUIRE(sparse_conv_dma_supported(c_in, c_out), ISF_ERR_UNSUPPORTED, "sparse_conv_dma: (Cin,Cout)=(%d,%d) not built",
              c_in, c_out);
  ISF_REQUIRE(nbr_stride % 128 == 0 && nbr_stride >= n_out, ISF_ERR_ARG, "sparse_conv_dma: bad nbr_stride");
  ISF_REQUIRE(!lmask || ((nx == 1 || nx == 3) && K % nx == 0), ISF_ERR_ARG, "sparse_conv_dma: %d taps in lines of %d", K, nx);
  const uint4* w = reinterpret_cast<const uint4*>(packed16);
  const float* winv = reinterpret_cast<const float*>(reinterpret_cast<const char*>(packed16) + (size_t)K * c_in * c_out * 4);
  const uint4* x = reinterpret_cast<const uint4*>(xs);
  const uint4* r = reinterpret_cast<const uint4*>(residual);
  uint4* y = reinterpret_cast<uint4*>(ys);
#define ISF_CALL_DMA(CI, NTT) dispatch_dma<CI, NTT>(mode, x, w, winv, K, c_out, nbr, lmask, nx, nbr_stride, n_out, scale, shift, r, relu, y, st, order, query, rowmap, trace)
  if (c_in == 32) return c_out == 32 ? ISF_CALL_DMA(32, 2) : ISF_CALL_DMA(32, 4);
  return c_out == 32 ? ISF_CALL_DMA(64, 2) : ISF_CALL_DMA(64, 4);
#undef ISF_CALL_DMA
}

}  // namespace isf

extern "C" {

int isf_sparse_conv_forward_dma(const void* features_split, int num_in, int c_in, const void* packed16, int num_taps,
                                int c_out, const int32_t* nbr, int nbr_stride, int num_out, const float* scale,
                                const float* shift, const void* residual_split, int relu, void* out_split, int mode,
                                const int32_t* order, isf_stream_t stream) {
  ISF_REQUIRE(num_in >= 0 && num_out >= 0 && c_in > 0 && c_out > 0 && num_taps > 0, ISF_ERR_ARG,
              "sparse_conv_forward_dma: bad arguments");
  if (num_out == 0) return ISF_OK;
  ISF_REQUIRE(features_split && packed16 && nbr && out_split && ((scale == nullptr) == (shift == nullptr)), ISF_ERR_ARG,
              "sparse_conv_forward_dma: null pointer");
  return isf::sparse_conv_forward_dma_impl(features_split, c_in, packed16, num_taps, c_out, nbr, nbr_stride, num_out,
                                           scale, shift, residual_split, relu, out_split, mode, isf::as_stream(stream),
                                           order, nullptr);
}

int isf_sparse_conv_forward_dma_lines(const void* features_split, int num_in, int c_in, const void* packed16, int num_taps,
                                      int taps_per_line, int c_out, const int32_t* lines, const uint32_t* mask,
                                      int nbr_stride, int num_out, const float* scale, const float* shift,
                                      const void* residual_split, int relu, void* out_split, int mode,
                                      isf_stream_t stream) {
  ISF_REQUIRE(num_in >= 0 && num_out >= 0 && c_in > 0 && c_out > 0 && num_taps > 0, ISF_ERR_ARG,
              "sparse_conv_forward_dma_lines: bad arguments");
  if (num_out == 0) return ISF_OK;
  ISF_REQUIRE(features_split && packed16 && lines && mask && out_split && ((scale == nullptr) == (shift == nullptr)),
              ISF_ERR_ARG, "sparse_conv_forward_dma_lines: null pointer");
  return isf::sparse_conv_forward_dma_impl(features_split, c_in, packed16, num_taps, c_out, lines, nbr_stride, num_out,
                                           scale, shift, residual_split, relu, out_split, mode, isf::as_stream(stream),
                                           nullptr, nullptr, mask, taps_per_line);
}

int isf_sparse_conv_dma_trace(const void* features_split, int num_in, int c_in, const void* packed16, int num_taps,
                              int taps_per_line, int c_out, const int32_t* table, const uint32_t* mask, int nbr_stride,
                              int num_out, const float* scale, const float* shift, const void* residual_split, int relu,
                              void* out_split, long long* trace, int trace_capacity_blocks, int* grid_blocks,
                              isf_stream_t stream) {
  ISF_REQUIRE(num_in >= 0 && num_out > 0 && features_split && packed16 && table && out_split && trace && grid_blocks &&
                  trace_capacity_blocks > 0 && ((scale == nullptr) == (shift == nullptr)), ISF_ERR_ARG,
              "sparse_conv_dma_trace: bad arguments");
  isf::Conv16LaunchInfo info;
  ISF_TRY(isf::sparse_conv_forward_dma_impl(features_split, c_in, packed16, num_taps, c_out, table, nbr_stride, num_out,
                                            scale, shift, residual_split, relu, out_split, 512, isf::as_stream(stream),
                                            nullptr, &info, mask, taps_per_line, nullptr, nullptr));
  const int blocks = 8 * (info.full + (info.half < 0 ? 0 : info.half));   // conv16_grid_blocks
  ISF_REQUIRE(blocks <= trace_capacity_blocks, ISF_ERR_ARG,
              "sparse_conv_dma_trace: the launch has %d workgroups, the trace buffer holds %d", blocks,
              trace_capacity_blocks);
  *grid_blocks = blocks;
  return isf::sparse_conv_forward_dma_impl(features_split, c_in, packed16, num_taps, c_out, table, nbr_stride, num_out,
                                           scale, shift, residual_split, relu, out_split, 512, isf::as_stream(stream),
                                           nullptr, nullptr, mask, taps_per_line, nullptr, trace);
}

}  // extern "C"
