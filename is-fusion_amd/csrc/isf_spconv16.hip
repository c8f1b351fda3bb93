// isf_spconv16.hip -- sparse convolution forward on the f16 matrix cores with fp32-equivalent accuracy.
//
// Arithmetic ("f16x3 split"): every fp32 operand is carried as two halves, v = hi + lo with hi = f16(v),
// lo = f16(v - hi) (22 significant bits), and a product is evaluated as  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi
// with v_mfma_f32_16x16x32_f16 accumulating in fp32 (the dropped a_lo*b_lo term is 2^-22 relative).
// Measured on the CPU (tests/test_host.py::test_f16x3_numerics): same error as an fp32 matmul.  The f16
// MFMA pipe is 16x the fp32 MFMA rate on gfx950 (2.5 PFLOP/s vs 157 TFLOP/s dense), so three passes are
// still 5.3x the fp32 matrix rate.  Weights are scaled by a power of two at pack time so their low halves
// stay in the normal f16 range; activations are stored between layers already split (same 4 B/element as
// fp32; layout: isf_common.h, "split activation format"), so the inner loop has no conversions at all.  |activation| must stay below 65504 (f16 max):
// outside that range the result is inf/NaN, never silently wrong.
//
// Structure (register-stationary; differs from the fp32 kernel in isf_spconv.hip):
//   workgroup = 4 waves, 128 consecutive output rows x BN = 16*NT output channels;
//   wave w owns rows [32w, 32w+32) = two 16-row MFMA row groups, all BN columns: 2*NT accumulators of
//   4 VGPRs stay in registers over all taps and input channels -- no LDS accumulation, no atomics;
//   A (gathered input rows, split format): global -> VGPR directly in the MFMA A-fragment layout, 32 B
//   (hi8|lo8) contiguous per lane, prefetched one step ahead;
//   B (weights, pre-split, fragment order): one contiguous NT*2 KiB block per (tap, 32-channel chunk) is
//   DMA'd global -> LDS with global_load_lds_dwordx4 into a double buffer shared by the 4 waves;
//   one barrier per step; taps that no row of the tile uses are skipped by the whole workgroup, taps that a
//   16-row group does not use are skipped by that wave (wave-uniform branch);
//   epilogue: LDS transpose, y = act(acc*scale + shift + residual), rows written once in split format.
#include "isf_spconv16.h"
#include "isf_spconv16_mult.h"

#include <stdlib.h>

#include <atomic>
#include <type_traits>

namespace isf {

typedef int i32x4 __attribute__((ext_vector_type(4)));

// phase trace (MODE bit 2048): dwords per wave record = kPhaseHdr + kPhaseStep * kPhaseMaxSteps
constexpr int kPhaseHdr = 8, kPhaseMaxSteps = kMaxTaps * 8, kPhaseStep = 8;

__device__ __forceinline__ unsigned long long shader_clock64() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  return t;
}

__device__ uint4 g_zero_line16[8];   // 128 zero bytes: what a lane without a neighbour gathers in the A2 loop

template <int NT, int RG, int KCH, int NW>
struct Conv16Smem {
  static constexpr int TM = 16 * RG * NW;                                // rows per workgroup
  static constexpr int nbr_bytes = kMaxTaps * TM * 4;
  static constexpr int bbuf_bytes = 2 * KCH * NT * 2048;                 // double-buffered weight stage
  static constexpr int EPN = NT > 4 ? 4 : NT;                            // column tiles per epilogue pass
  static constexpr int epi_bytes = NW * 16 * (16 * EPN + 4) * 4;         // per-wave 16 x (16*EPN+4) fp32
  // the epilogue tile overlays the neighbour table and the weight ring (both dead by then)
  static constexpr int main_bytes = nbr_bytes + bbuf_bytes;
  static constexpr int work_bytes = main_bytes > epi_bytes ? main_bytes : epi_bytes;
  static constexpr int bytes = work_bytes + 256;
};


// NW waves per workgroup (4, 8 or 16): all of them share one weight stage per step, so the weight bytes a CU pulls
// through its vector memory path per MFMA fall with NW.
// MODE bit 1 = single-pass mode (precision 2 of isf_encoder_options, mode 1 of isf_sparse_conv_forward_f16x3): only the hi halves of activations and weights are fetched
// and multiplied -- plain f16 operands with fp32 accumulation, the accuracy of the reference under fp16 autocast
// (indice_conv_half), one MFMA per product instead of three.  Same buffers, same layouts; outputs are still written
// split.  Never the default: the headline configuration is fp32-class.
// MODE bits 2 / 4 / 8 are TIMING DIAGNOSTICS (the `diagnostic` option / `mode` argument; results are garbage): 2 = no activation gathers
// (A = 0), 4 = no weight DMA, 8 = no main loop (prologue + epilogue only) -- the knock-out decomposition of DESIGN.md
// section 5 as a permanent tool (tools/conv_knockout.sh).  MODE bit 16 (valid results) switches the neighbour sharing of
// the gathers off: the reference the sharing is checked against bit for bit.
template <int CIN, int NT, int RG, int NW, int MODE = 0>
__global__ __launch_bounds__(64 * NW, (NW >= 16 ? 1 : (NT * RG >= 16 ? 2 : 3))) void spconv_f16x3_kernel(
    const uint4* __restrict__ xs, const int32_t* __restrict__ nbr, int nbr_stride,
    const uint4* __restrict__ wpk, const float* __restrict__ w_inv_scale, int K, int cout,
    const float* __restrict__ scale, const float* __restrict__ shift, const uint4* __restrict__ residual,
    uint4* __restrict__ ys, int n_out, int relu, Conv16Plan plan, const int32_t* __restrict__ order,
    long long* __restrict__ trace, const int32_t* __restrict__ rowmap /* nullptr | position -> output row (sorted launch) */,
    float* __restrict__ ks_scratch, unsigned* __restrict__ ks_count /* chunk-split launches (MODE bit 524288) */) {
  // MODE bit 524288: CHUNK SPLIT.  A tile is computed by TWO workgroups -- blocks b and b + gridDim.x / 2 (ids 8 apart
  // share an XCD, so these do too) -- each over half of the 32-channel chunks; both store their accumulator tile, the
  // SECOND to arrive at the tile's counter adds the other's and runs the epilogue (a + b == b + a: the result does not
  // depend on who arrives first; per output element: (chunks of the lower half, taps ascending) + (chunks of the upper
  // half, taps ascending)).  Why: the 256-column layers live on the small deep levels -- 40 k rows at B = 4 x 300 k
  // points = 636 workgroups on 768 slots, ONE round, whose duration is its longest tile's (27 taps x 8 chunks = 216
  // steps of ~1.2 us) while the average tile has 136: 1 272 half-length workgroups refill the slots as they drain.
  constexpr bool KSPLIT = (MODE & 524288) != 0;
  // MODE bit 512: per-workgroup trace (isf_sparse_conv_trace): 8 x int64 per workgroup -- constant-clock time stamps at
  // entry / after the prologue / after the multiply loop / at exit, steps, HW_ID, XCC_ID, first row | half << 32
  constexpr bool TRACE = (MODE & 512) != 0;
  // MODE bit 2048 (with 512): PHASE TRACE (isf_sparse_conv_phase_trace; the image has no thread-trace decoder, so this is
  // the kernel's own instruction-level account): every wave stamps the shader clock (s_memtime, 1 tick = 1 shader cycle)
  // at the top of each step / after its s_waitcnt vmcnt(0) / after the barrier / after issuing the next step's loads;
  // the multiply section is what is left until the next top.  Per wave, behind the per-workgroup records:
  // kPhaseHdr dwords {clock at loop entry lo, hi, HW_ID, steps, rgm[0], rgm[1], wg_mask, clock at loop exit lo} +
  // 8 dwords per step {top, after wait, after barrier, after issue, after the index reads, after the gathers, 0, 0}.  The
  // stamps cost ~6 scalar-memory round trips per step (measured against the untraced launch by tools/conv_phase_trace.py).
  constexpr bool PHASE = (MODE & 2048) != 0;
  long long t_entry = 0, t_pro = 0, t_loop = 0;
  if (TRACE) t_entry = wall_clock64();
  constexpr bool HALF = (MODE & 1) != 0, NOGATHER = (MODE & 2) != 0, NODMA = (MODE & 4) != 0, NOLOOP = (MODE & 8) != 0;
  // MODE bit 256 (with bit 1): F16 STORAGE -- input, residual and output rows are plain f16 (2 bytes per element, the
  // reference's indice_conv_half data type end to end: src/all.cc:35-37) instead of 4-byte split rows; half the
  // activation bytes of every layer (BASELINE configs[4], the HBM-bound run).  isf_encoder_options.precision = 2.
  constexpr bool F16IO = (MODE & 256) != 0;
  constexpr bool STAG = (MODE & 65536) != 0;   // staggered issue phases, see the main loop
  // MODE bit 262144: the gathered rows TWO steps ahead (three register sets in rotation, loop unrolled by three, counted
  // vmcnt waits) -- the A2 loop below; 4-wave deep shapes only
  constexpr bool A2 = (MODE & 262144) != 0;
  static_assert(!F16IO || HALF, "f16 storage implies single-pass f16 arithmetic");
  // neighbour sharing of the gathers (load_A below) where it was measured to pay -- the layers whose gathers saturate
  // the vector-memory path: 64 -> 64 0.91 -> 0.76 ms, 64 -> 32 0.138 -> 0.128, 32 -> 32 0.312 -> 0.301 per step; the
  // layers with >= 128 output columns (and 32 -> 64) lose 3-5 % to its DPP / select / index work and keep plain gathers
  // (profiles/r02_call3_sharing.txt).  MODE bit 16 switches it off everywhere (bit-equality reference).
  constexpr bool SHARE = (MODE & 16) == 0 && NT <= 4 && (CIN >= 64 || NT == 2);
  constexpr int KCH = Conv16Step<CIN, NT>::KCH;
  // the multiply section in hand-scheduled assembly (isf_spconv16_mult.h) for the deep layers' 4-wave shape; MODE bit 16
  // ("no neighbour sharing": the deep layers do not share anyway) keeps hipcc's section -- the bit-equality reference
  constexpr bool ASMM = (MODE & (16 | 65536 | 262144)) == 0 && (RG == 2 || RG == 1) && NT == 8 && KCH == 1 &&
                        (NW == 4 || NW == 8) && (MODE & 1) == 0;
  using S = Conv16Smem<NT, RG, KCH, NW>;
  constexpr int NTHR = 64 * NW;
  constexpr int TM = S::TM;
  constexpr int WR = 16 * RG;         // rows per wave
  constexpr int NCH = CIN / 32;       // 32-channel chunks
  constexpr int NCG = NCH / KCH;      // chunk groups (steps per tap)
  constexpr int CH8 = CIN / 8;        // 8-channel (32-byte) units per input row
  constexpr int BN = 16 * NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* nbr_l = reinterpret_cast<int*>(smem);                                  // [27][TM]
  uint4* bbuf = reinterpret_cast<uint4*>(smem + S::nbr_bytes);                // [2][NT][2][64]
  int* misc = reinterpret_cast<int*>(smem + S::work_bytes);                   // [NW] wave masks

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kg = lane >> 4;
  const int ncb = cout / BN;
  int cb, row0, row_end;
  bool half_tile;   // every wave owns one 16-row group (rows row0 + 16 * wave ..) instead of RG
  const int ks_grid = KSPLIT ? (int)(gridDim.x >> 1) : (int)gridDim.x;
  const int ks_h = (KSPLIT && (int)blockIdx.x >= ks_grid) ? 1 : 0;       // which half of the chunks
  const int bid = (int)blockIdx.x - ks_h * ks_grid;
  if (!conv16_tile_of_block(ncb, plan, TM, n_out, cb, row0, row_end, half_tile, order, bid)) return;
  if constexpr (RG == 1) half_tile = false;   // one 16-row group per wave already: a short tile is a full tile with fewer rows
  const int ntiles_total = cout >> 4;

  // ---- prologue: neighbour tile -> LDS, per-wave tap mask
  {   // all loads first, then all LDS stores (a load -> wait -> store loop costs one L2 round trip per iteration)
    constexpr int NB_IT = (kMaxTaps * TM + NTHR - 1) / NTHR;
    int tmp[NB_IT];
#pragma unroll
    for (int it = 0; it < NB_IT; ++it) {
      const int i = tid + it * NTHR;
      const int k = i / TM, r = i - k * TM;
      tmp[it] = -1;   // stride = round_up(n, 128): rows beyond it have no neighbours
      // LDS position r = wave * WR + (row of the wave); in a half tile only the first WR / 2 rows of a wave exist
      const int rw = r % WR;
      const int grow = half_tile ? (rw < WR / 2 ? row0 + (r / WR) * (WR / 2) + rw : row_end) : row0 + r;
      if (i < K * TM && grow < row_end) tmp[it] = nbr[(size_t)k * nbr_stride + grow];   // row_end <= n_out <= stride
    }
#pragma unroll
    for (int it = 0; it < NB_IT; ++it) {
      const int i = tid + it * NTHR;
      if (i < K * TM) nbr_l[i] = tmp[it];
    }
  }
  __syncthreads();
  // per-row-group tap masks (bit k: some row of the 16-row group has a neighbour through tap k), wave-uniform
  unsigned rgm[RG];
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) rgm[rg] = 0;
  for (int k = 0; k < K; ++k) {
#pragma unroll
    for (int rg = 0; rg < RG; rg += 4) {   // 64 lanes cover 4 row groups per ballot
      const int r = rg * 16 + lane;
      const bool has = r < WR && nbr_l[k * TM + wave * WR + r] >= 0;
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
  static_assert(!KSPLIT || (NCG % 2 == 0 && (MODE & (512 | 262144)) == 0), "chunk split: even chunk-group count, no trace / A2 loop");
  const int nsteps = NOLOOP ? 0 : ntaps * (KSPLIT ? NCG / 2 : NCG);
  if (TRACE) t_pro = wall_clock64();

  f32x4 acc[RG][NT];
#pragma unroll
  for (int rg = 0; rg < RG; ++rg)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[rg][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // step s -> (chunk, tap): CHUNK-OUTER, taps (set bits of wg_mask, increasing) inner.  An input row is the
  // tap-k neighbour of up to ~15 output rows of this tile and its neighbours, so with the taps innermost the
  // same 128-byte row segment is re-gathered within a few steps (L1/L2 hits) instead of 8 chunks later.
  struct Cursor {
    unsigned rem;   // taps of the current chunk not yet visited
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

  // Prefetch pipeline: weights (LDS double buffer, DMA) and A fragments (registers) are both fetched ONE step
  // ahead, issued right after the barrier so that they fly during the MFMAs of the current step.  A 16-row
  // group that has no neighbour through the tap neither gathers nor multiplies (wave-uniform branches).
  // Neighbour sharing: consecutive taps of a line differ by one cell in x, and consecutive output rows are mostly
  // x-neighbours, so the row that lane r needs for the NEXT tap is very often the row that lane r+1 holds for the
  // CURRENT tap (nbr[next][r] == nbr[cur][r+1]).  Those lanes take their fragment from the right-hand lane's
  // registers (DPP row_shl:1 -- a DPP row is exactly the 16 rows of a row group at one k-group) instead of loading it
  // again: the gathers are what saturates the CU's vector-memory return path (64 B/clk; knock-outs in
  // profiles/r02_call1_knockout_variants.txt), and every shared row is a 64-byte request that is never made.  Purely
  // index-driven, so it needs no knowledge of the kernel geometry and never changes a value: the shared fragment IS
  // the fragment the load would have returned.
  uint4 a_nxt[RG][KCH][2];  // [row group][chunk of the step][hi, lo]
  uint4 a_cur[RG][KCH][2];
  auto shl1 = [](const uint4 v) {   // lane (k-group, col) <- lane (k-group, col + 1); col 15 gets 0
    uint4 r;
    r.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.x, 0x101, 0xf, 0xf, true);
    r.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.y, 0x101, 0xf, 0xf, true);
    r.z = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.z, 0x101, 0xf, 0xf, true);
    r.w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.w, 0x101, 0xf, 0xf, true);
    return r;
  };
  // prev_tap >= 0: a_cur holds the fragments of (prev_tap, same chunk group), landed
  auto load_A = [&](int tap, int cg, int prev_tap) {
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      if ((rgm[rg] >> tap) & 1u) {
        const int idx = nbr_l[tap * TM + wave * WR + rg * 16 + col];
        bool share = false;
        if (SHARE && prev_tap >= 0 && ((rgm[rg] >> prev_tap) & 1u)) {   // wave-uniform condition
          // the right-hand row's index through the previous tap (the entry after the last row of the group is another
          // group's or another tap's: col 15 never shares)
          const int right = nbr_l[prev_tap * TM + wave * WR + rg * 16 + (col < 15 ? col + 1 : col)];
          share = idx >= 0 && col < 15 && idx == right;
#pragma unroll
          for (int kc = 0; kc < KCH; ++kc) {
            const uint4 sh = shl1(a_cur[rg][kc][0]), sl = shl1(a_cur[rg][kc][1]);
            a_nxt[rg][kc][0] = share ? sh : make_uint4(0, 0, 0, 0);
            a_nxt[rg][kc][1] = share ? sl : make_uint4(0, 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int kc = 0; kc < KCH; ++kc) {
            a_nxt[rg][kc][0] = make_uint4(0, 0, 0, 0);
            a_nxt[rg][kc][1] = make_uint4(0, 0, 0, 0);
          }
        }
        if (idx >= 0 && !share && !NOGATHER) {
#pragma unroll
          for (int kc = 0; kc < KCH; ++kc) {
            const uint4* p = F16IO ? xs + (size_t)idx * CH8 + (cg * KCH + kc) * 4 + kg
                                   : xs + ((size_t)idx * CH8 + (cg * KCH + kc) * 4) * 2 + kg;   // chunk base + k-group
            a_nxt[rg][kc][0] = p[0];   // 4 contiguous hi pieces per row and instruction
            if (!HALF) a_nxt[rg][kc][1] = p[4];   // 4 contiguous lo pieces
          }
        }
      }
    }
  };
  const unsigned bbuf_addr = __builtin_amdgcn_readfirstlane(lds_addr(bbuf));
  constexpr int PW = NT * 128 / NW;                       // 16-byte weight pieces per wave and step (KCH = 1)
  constexpr bool RUNS = (MODE & 131072) == 0 && KCH == 1 && !NODMA && (HALF ? (PW == 128 || PW == 256)
                                                                                 : (PW == 64 || PW == 128 || PW == 256));
  auto stage_B = [&](int tap, int cg, int buf) {
    if constexpr (RUNS) {   // this wave's share as ONE contiguous run: one M0 set-up, PW / 64 loads (glds16_run)
      const uint4* src = wpk + (((size_t)tap * NCH + cg) * ntiles_total + cb * NT) * 128 + wave * PW + lane;
      const unsigned dst = bbuf_addr + (unsigned)(buf * (NT * 128) + wave * PW) * 16u;
      if constexpr (HALF) glds16_run_hi<PW / 64>(src, dst);   // single-pass f16: the hi KiB of each column tile only
      else glds16_run<PW / 64>(src, dst);
      return;
    }
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc) {
      const uint4* src = wpk + (((size_t)tap * NCH + cg * KCH + kc) * ntiles_total + cb * NT) * 128;
      const unsigned dst = bbuf_addr + (unsigned)((buf * KCH + kc) * (NT * 128)) * 16u;
#pragma unroll
      for (int i = 0; i < (NT * 128 + NTHR - 1) / NTHR; ++i) {
        const int base = i * NTHR + wave * 64;   // wave-uniform: this wave's 64 consecutive 16-byte pieces
        // pieces come in 64-lane groups, hi (even group) then lo (odd group) per column tile
        if (base < NT * 128 && !(HALF && ((base >> 6) & 1)) && !NODMA) glds16(src + base + lane, dst + (unsigned)base * 16u);
      }
    }
  };

  if constexpr (A2) {
    static_assert(KCH == 1 && NW == 4 && !SHARE && !HALF, "the A2 loop is built for the 4-wave deep shapes");
    uint4 A0[RG][2], A1[RG][2], A2s[RG][2];
    auto read_idx = [&](int tap_, int (&idx)[RG]) {
#pragma unroll
      for (int rg = 0; rg < RG; ++rg)
        idx[rg] = ((rgm[rg] >> tap_) & 1u) ? nbr_l[tap_ * TM + wave * WR + rg * 16 + col] : -1;
    };
    // The gathers are inline asm: hipcc's scoreboard must NOT know them, or it drains vmcnt(0) in front of the first MFMA
    // that reads a set -- i.e. waits for the rows of the step after next as well (seen in the ISA of the first version).
    // No exec-masked loads either (a lane without a neighbour reads a zero line): a merge of "loaded" and "not loaded"
    // lanes is where the compiler would touch the registers before our counted wait.  The sets are only read by the
    // MFMAs of their own step, behind that wait.
    auto gather = [&](uint4 (&S)[RG][2], int tap_, int ch_, const int (&idx)[RG]) -> int {
      int n = 0;
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) {
        if ((rgm[rg] >> tap_) & 1u) {                      // wave-uniform
          const uint4* p = idx[rg] >= 0 ? xs + ((size_t)idx[rg] * CH8 + ch_ * 4) * 2 + kg : g_zero_line16 + kg;
          asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:64"
                       : "=&v"(S[rg][0]), "=&v"(S[rg][1]) : "v"(p) : "memory");
          n += 2;
        }
      }
      return n;
    };
    auto mult = [&](const uint4 (&S)[RG][2], int tap_, int buf) {
      if (!((wmask >> tap_) & 1u)) return;
      const uint4* b = bbuf + buf * (NT * 128) + lane;
      uint4 bhu_n = b[0], blu_n = b[64];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const uint4 bhu = bhu_n, blu = blu_n;
        if (nt + 1 < NT) {
          bhu_n = b[((nt + 1) * 2 + 0) * 64];
          blu_n = b[((nt + 1) * 2 + 1) * 64];
        }
        const h8 bh = *reinterpret_cast<const h8*>(&bhu);
        const h8 bl = *reinterpret_cast<const h8*>(&blu);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
          if ((rgm[rg] >> tap_) & 1u) {
            const h8 ah = *reinterpret_cast<const h8*>(&S[rg][0]);
            const h8 al = *reinterpret_cast<const h8*>(&S[rg][1]);
            acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[rg][nt], 0, 0, 0);
            acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[rg][nt], 0, 0, 0);
            acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[rg][nt], 0, 0, 0);
          }
        }
      }
    };
    // cursors of steps s (multiply), s + 1 (weights), s + 2 (gathers), s + 3 (index reads)
    Cursor cs{0u, -1, -1}, cb, cg, ci;
    int idx_pre[RG];
    int pend = 0;              // gather instructions issued AFTER the newest weight run: what may stay in flight at the wait
    if (nsteps > 0) {
      advance(cs);
      cb = cs; cg = cs; ci = cs;
      int i0[RG];
      read_idx(cs.tap, i0);
      (void)gather(A0, cs.tap, cs.ch, i0);
      stage_B(cs.tap, cs.ch, 0);
      if (nsteps > 1) {
        advance(cb);
        cg = cb; ci = cb;
        int i1[RG];
        read_idx(cb.tap, i1);
        pend = gather(A1, cb.tap, cb.ch, i1);
        if (nsteps > 2) {
          advance(cg);
          ci = cg;
          read_idx(cg.tap, idx_pre);
        }
      }
    }
    auto body = [&](int s, const uint4 (&Su)[RG][2], uint4 (&Sl)[RG][2]) {
      // A(s) and this wave's share of B(s) have landed once at most `pend` (the gathers of A(s + 1)) are still in flight
      if (pend >= 4) __builtin_amdgcn_s_waitcnt(0x0F74);        // vmcnt(4)
      else if (pend >= 2) __builtin_amdgcn_s_waitcnt(0x0F72);   // vmcnt(2)
      else __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0)
      __syncthreads();
      const int tap_s = cs.tap;
      pend = 0;
      if (s + 1 < nsteps) stage_B(cb.tap, cb.ch, (s + 1) & 1);  // weights FIRST: the counted wait above relies on the order
      if (s + 2 < nsteps) {
        pend = gather(Sl, cg.tap, cg.ch, idx_pre);
        if (s + 3 < nsteps) {
          advance(ci);
          read_idx(ci.tap, idx_pre);
        }
        advance(cg);
      }
      if (s + 1 < nsteps) advance(cb);
      mult(Su, tap_s, s & 1);
      advance(cs);
    };
    for (int s = 0; s < nsteps; s += 3) {
      body(s, A0, A2s);
      if (s + 1 < nsteps) body(s + 1, A1, A0);
      if (s + 2 < nsteps) body(s + 2, A2s, A1);
    }
  }
  Cursor cur{0u, -1, KSPLIT ? ks_h * (NCG / 2) - 1 : -1};   // chunk split: this workgroup's first chunk group
  int idx_pre[RG];   // deep layers: row indices of the step after the current one (see the main loop)
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) idx_pre[rg] = -1;
  if (!A2 && nsteps > 0) {
    advance(cur);
    load_A(cur.tap, cur.ch, -1);
    stage_B(cur.tap, cur.ch, 0);
    if (nsteps > 1) {
      Cursor la = cur;
      advance(la);
#pragma unroll
      for (int rg = 0; rg < RG; ++rg)
        idx_pre[rg] = ((rgm[rg] >> la.tap) & 1u) ? nbr_l[la.tap * TM + wave * WR + rg * 16 + col] : -1;
    }
  }
  unsigned* ph = nullptr;          // this wave's phase record
  unsigned ph_top = 0, ph_wait = 0, ph_bar = 0, ph_issue = 0, ph_idx = 0, ph_gath = 0;
  if (PHASE) {
    ph = reinterpret_cast<unsigned*>(trace + (size_t)gridDim.x * 8) +
         ((size_t)blockIdx.x * NW + wave) * (kPhaseHdr + kPhaseStep * kPhaseMaxSteps);
    const unsigned long long t = shader_clock64();
    unsigned hw_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    if (lane == 0) {
      ph[0] = (unsigned)t; ph[1] = (unsigned)(t >> 32); ph[2] = hw_id; ph[3] = (unsigned)nsteps;
      ph[4] = rgm[0]; ph[5] = RG > 1 ? rgm[RG > 1 ? 1 : 0] : 0u; ph[6] = wg_mask;
    }
    ph_top = (unsigned)t;
  }
  for (int s = 0; s < (A2 ? 0 : nsteps); ++s) {
    const int tap = cur.tap, ch = cur.ch;
#pragma unroll
    for (int rg = 0; rg < RG; ++rg)
#pragma unroll
      for (int kc = 0; kc < KCH; ++kc) { a_cur[rg][kc][0] = a_nxt[rg][kc][0]; a_cur[rg][kc][1] = a_nxt[rg][kc][1]; }
    // this wave's share of B(s) (and its A(s) rows) must have landed before anyone reads the buffer
    // (the builtin, not inline asm, so that hipcc's own scoreboard knows every tracked load has landed too and
    //  does not re-wait for the A(s) registers in the middle of the next prefetch; simm16 0x0F70 = vmcnt(0))
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (PHASE) ph_wait = (unsigned)shader_clock64();
    __syncthreads();  // B(s) complete for every wave; everyone is done reading buffer (s+1)&1
    if (PHASE) ph_bar = (unsigned)shader_clock64();
    // column tiles [i0, i1) of this step's products (i = kc * NT + nt); B fragments one tile ahead
    auto multiply = [&](int i0, int i1) {
      if (!((wmask >> tap) & 1u)) return;
      if constexpr (ASMM) {
        // the hand-scheduled section (isf_spconv16_mult.h): the whole step's products, B fragments through a[0:31]
        const unsigned vb = bbuf_addr + (unsigned)((s & 1) * (NT * 128) + lane) * 16u;
        const int n0 = (int)((rgm[0] >> tap) & 1u), n1 = (int)((rgm[RG > 1 ? 1 : 0] >> tap) & 1u);
        const i32x4 a0h = *reinterpret_cast<const i32x4*>(&a_cur[0][0][0]), a0l = *reinterpret_cast<const i32x4*>(&a_cur[0][0][1]);
        const i32x4 a1h = *reinterpret_cast<const i32x4*>(&a_cur[RG > 1 ? 1 : 0][0][0]),
                    a1l = *reinterpret_cast<const i32x4*>(&a_cur[RG > 1 ? 1 : 0][0][1]);
        constexpr int R1 = RG > 1 ? 1 : 0, T7 = NT > 7 ? 7 : 0, T6 = NT > 7 ? 6 : 0, T5 = NT > 7 ? 5 : 0, T4 = NT > 7 ? 4 : 0,
                      T3 = NT > 7 ? 3 : 0, T2 = NT > 7 ? 2 : 0, T1 = NT > 7 ? 1 : 0;
        if constexpr (RG == 1) {            // one row group per wave: the pair pipeline without the case split
          asm volatile(ISF_TM_TEXT_RG1
                       : [c00] "+v"(acc[0][0]), [c01] "+v"(acc[0][T1]), [c02] "+v"(acc[0][T2]), [c03] "+v"(acc[0][T3]),
                         [c04] "+v"(acc[0][T4]), [c05] "+v"(acc[0][T5]), [c06] "+v"(acc[0][T6]), [c07] "+v"(acc[0][T7])
                       : [a0h] "v"(a0h), [a0l] "v"(a0l), [vb] "v"(vb)
                       : ISF_TM_CLOBBERS);
          return;
        }
        if constexpr (NW == 8) {            // 128 registers per wave: one column tile per fragment buffer, a[0:15]
          asm volatile(ISF_TN_TEXT
                       : [c00] "+v"(acc[0][0]), [c01] "+v"(acc[0][T1]), [c02] "+v"(acc[0][T2]), [c03] "+v"(acc[0][T3]),
                         [c04] "+v"(acc[0][T4]), [c05] "+v"(acc[0][T5]), [c06] "+v"(acc[0][T6]), [c07] "+v"(acc[0][T7]),
                         [c10] "+v"(acc[R1][0]), [c11] "+v"(acc[R1][T1]), [c12] "+v"(acc[R1][T2]), [c13] "+v"(acc[R1][T3]),
                         [c14] "+v"(acc[R1][T4]), [c15] "+v"(acc[R1][T5]), [c16] "+v"(acc[R1][T6]), [c17] "+v"(acc[R1][T7])
                       : [a0h] "v"(a0h), [a0l] "v"(a0l), [a1h] "v"(a1h), [a1l] "v"(a1l), [vb] "v"(vb), [n0] "s"(n0), [n1] "s"(n1)
                       : ISF_TN_CLOBBERS);
          return;
        }
        asm volatile(ISF_TM_TEXT
                     : [c00] "+v"(acc[0][0]), [c01] "+v"(acc[0][T1]), [c02] "+v"(acc[0][T2]), [c03] "+v"(acc[0][T3]),
                       [c04] "+v"(acc[0][T4]), [c05] "+v"(acc[0][T5]), [c06] "+v"(acc[0][T6]), [c07] "+v"(acc[0][T7]),
                       [c10] "+v"(acc[R1][0]), [c11] "+v"(acc[R1][T1]), [c12] "+v"(acc[R1][T2]), [c13] "+v"(acc[R1][T3]),
                       [c14] "+v"(acc[R1][T4]), [c15] "+v"(acc[R1][T5]), [c16] "+v"(acc[R1][T6]), [c17] "+v"(acc[R1][T7])
                     : [a0h] "v"(a0h), [a0l] "v"(a0l), [a1h] "v"(a1h), [a1l] "v"(a1l), [vb] "v"(vb), [n0] "s"(n0), [n1] "s"(n1)
                     : ISF_TM_CLOBBERS);
        return;
      }
      const uint4* b = bbuf + (s & 1) * (KCH * NT * 128) + lane;
      bool need[RG];
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) need[rg] = (rgm[rg] >> tap) & 1u;   // scalar (wave-uniform)
      uint4 bhu_n = b[(i0 * 2 + 0) * 64], blu_n = make_uint4(0, 0, 0, 0);   // the next B fragments are read from LDS while these multiply
      if (!HALF) blu_n = b[(i0 * 2 + 1) * 64];
#pragma unroll
      for (int i = i0; i < i1; ++i) {
        const int kc = i / NT, nt = i % NT;
        const uint4 bhu = bhu_n, blu = blu_n;
        if (i + 1 < i1) {
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
    };
    if (STAG) {
      // STAGGERED issue phases (MODE bit 65536, round 5): the barrier releases all waves into their issue phase at once
      // and a gather costs the CU's address path 64 cycles -- the burst, not the average load, is what a wave waits out
      // (profiles/r05_att_256.txt: 31 % of the loop).  Every wave queues its weight pieces first (the others need them
      // at the next barrier); the first half of the workgroup's waves then gathers and multiplies as before, the second
      // half multiplies the first half of the column tiles while the others own the address path, gathers, multiplies
      // the rest.  Same products in the same order per accumulator: bit-identical.
      const bool late = (wave & (NW / 2)) != 0;          // wave-uniform
      const bool more = s + 1 < nsteps;
      if (more) {
        advance(cur);
        stage_B(cur.tap, cur.ch, (s + 1) & 1);
        if (!late) load_A(cur.tap, cur.ch, cur.ch == ch ? tap : -1);
      }
      multiply(0, KCH * NT / 2);
      if (more && late) load_A(cur.tap, cur.ch, cur.ch == ch ? tap : -1);
      if (PHASE) ph_issue = (unsigned)shader_clock64();
      multiply(KCH * NT / 2, KCH * NT);
    } else if (PHASE && !SHARE && (NW != 4 || (MODE & 131072) != 0)) {
      // the same issue phase in three stamped pieces: index reads from LDS (+ their wait) | the gathers | the weight DMA
      if (s + 1 < nsteps) {
        advance(cur);
        int idxs[RG];
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
          idxs[rg] = ((rgm[rg] >> cur.tap) & 1u) ? nbr_l[cur.tap * TM + wave * WR + rg * 16 + col] : -1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ph_idx = (unsigned)shader_clock64();
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
          if ((rgm[rg] >> cur.tap) & 1u) {
#pragma unroll
            for (int kc = 0; kc < KCH; ++kc) {
              a_nxt[rg][kc][0] = make_uint4(0, 0, 0, 0);
              a_nxt[rg][kc][1] = make_uint4(0, 0, 0, 0);
              if (idxs[rg] >= 0) {
                const uint4* p = xs + ((size_t)idxs[rg] * CH8 + (cur.ch * KCH + kc) * 4) * 2 + kg;
                a_nxt[rg][kc][0] = p[0];
                a_nxt[rg][kc][1] = p[4];
              }
            }
          }
        }
        ph_gath = (unsigned)shader_clock64();
        stage_B(cur.tap, cur.ch, (s + 1) & 1);
      } else {
        ph_idx = ph_gath = (unsigned)shader_clock64();
      }
      ph_issue = (unsigned)shader_clock64();
      multiply(0, KCH * NT);
    } else if (!SHARE && (MODE & 131072) == 0 && NW == 4) {   // (8-wave workgroups: the two extra registers cost a wave)
      // deep layers (no neighbour sharing): the row indices of the NEXT step's gathers were read from the LDS table one
      // step ago (idx_pre) -- the phase trace put the index ds_reads + their wait at 230 exposed cycles per step
      // (profiles/r05_att_256_v2.txt) -- and this step ends by reading the ones of the step after next
      if (s + 1 < nsteps) {
        advance(cur);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
          if ((rgm[rg] >> cur.tap) & 1u) {
#pragma unroll
            for (int kc = 0; kc < KCH; ++kc) {
              a_nxt[rg][kc][0] = make_uint4(0, 0, 0, 0);
              a_nxt[rg][kc][1] = make_uint4(0, 0, 0, 0);
              if (idx_pre[rg] >= 0 && !NOGATHER) {
                const uint4* p = F16IO ? xs + (size_t)idx_pre[rg] * CH8 + (cur.ch * KCH + kc) * 4 + kg
                                       : xs + ((size_t)idx_pre[rg] * CH8 + (cur.ch * KCH + kc) * 4) * 2 + kg;
                a_nxt[rg][kc][0] = p[0];
                if (!HALF) a_nxt[rg][kc][1] = p[4];
              }
            }
          }
        }
        if (PHASE) ph_idx = ph_bar, ph_gath = (unsigned)shader_clock64();   // no index wait any more: idx piece = 0
        stage_B(cur.tap, cur.ch, (s + 1) & 1);   // (weights BEFORE the gathers measured 0.8 % slower: profiles/r05_issue_phase_ab.txt)
        if (s + 2 < nsteps) {
          Cursor la = cur;
          advance(la);
#pragma unroll
          for (int rg = 0; rg < RG; ++rg)
            idx_pre[rg] = ((rgm[rg] >> la.tap) & 1u) ? nbr_l[la.tap * TM + wave * WR + rg * 16 + col] : -1;
        }
      }
      else if (PHASE) ph_idx = ph_gath = ph_bar;
      if (PHASE) ph_issue = (unsigned)shader_clock64();
      multiply(0, KCH * NT);
    } else {
      if (s + 1 < nsteps) {
        advance(cur);
        load_A(cur.tap, cur.ch, cur.ch == ch ? tap : -1);   // a_cur = (tap, ch), landed (vmcnt(0) above)
        stage_B(cur.tap, cur.ch, (s + 1) & 1);
      }
      if (PHASE) ph_issue = (unsigned)shader_clock64();
      multiply(0, KCH * NT);
    }
    if (PHASE) {
      const unsigned t_end = (unsigned)shader_clock64();
      if (lane == 0 && s < kPhaseMaxSteps) {
        *reinterpret_cast<uint4*>(ph + kPhaseHdr + kPhaseStep * s) = make_uint4(ph_top, ph_wait, ph_bar, ph_issue);
        *reinterpret_cast<uint4*>(ph + kPhaseHdr + kPhaseStep * s + 4) = make_uint4(ph_idx, ph_gath, 0u, 0u);
      }
      ph_top = t_end;
    }
  }
  if (PHASE && lane == 0) ph[7] = ph_top;
  __builtin_amdgcn_s_waitcnt(0x0F70);
  // the last MFMAs were issued from assembly: hipcc's hazard recogniser has not seen them (XDL write -> VALU / LDS read)
  if constexpr (ASMM) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  __syncthreads();  // all waves done with the weight buffers -> reuse as the epilogue transpose tile
  if (TRACE) t_loop = wall_clock64();

  if constexpr (KSPLIT) {
    // accumulator tiles in the C/D layout, one KiB per (row group, column tile) and wave: [tile][half][wave][rg][nt][lane].
    // The exchange goes THROUGH the caches (system-scope stores / loads, sc0 sc1): a device-scope release fence here is a
    // write-back of the XCD's whole L2 per workgroup (buffer_wbl2) -- measured: the launch 0.225 -> 0.49 ms.  Stores acked
    // (s_waitcnt) before the arrival is counted; the counter is a device-scope atomic; the loads are issued behind it.
    f32x4* mine = reinterpret_cast<f32x4*>(ks_scratch) + (((size_t)bid * 2 + ks_h) * NW + wave) * (RG * NT * 64) + lane;
#pragma unroll
    for (int rg = 0; rg < RG; ++rg)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(mine + (rg * NT + nt) * 64), "v"(acc[rg][nt]) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) misc[0] = (int)atomicAdd(ks_count + bid, 1u);
    __syncthreads();
    const int arrived = misc[0];
    if (arrived == 0) return;              // first: the other workgroup of this tile finishes it
    const f32x4* other = reinterpret_cast<const f32x4*>(ks_scratch) + (((size_t)bid * 2 + (ks_h ^ 1)) * NW + wave) * (RG * NT * 64) + lane;
    f32x4 o[RG * NT];
#pragma unroll
    for (int k = 0; k < RG * NT; ++k)
      asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(o[k]) : "v"(other + k * 64) : "memory");
    static_assert(RG * NT == 8 || RG * NT == 16, "chunk split: 8 or 16 accumulator tiles per wave");
    // the wait carries the loaded registers as operands, so that nothing that reads them is scheduled in front of it
    if constexpr (RG * NT == 16)
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]), "+v"(o[8]),
                     "+v"(o[9]), "+v"(o[10]), "+v"(o[11]), "+v"(o[12]), "+v"(o[13]), "+v"(o[14]), "+v"(o[RG * NT - 1])
                   : : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7])
                   : : "memory");
#pragma unroll
    for (int rg = 0; rg < RG; ++rg)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[rg][nt] = acc[rg][nt] + o[rg * NT + nt];
    if (tid == 0) ks_count[bid] = 0u;      // the zero back for the next launch
  }

  float* tile_l = reinterpret_cast<float*>(smem) + wave * (Conv16Epi<NT, RG>::wave_bytes / 4);
  conv16_epilogue<NT, RG, F16IO>(acc, tile_l, lane, row0 + wave * (half_tile ? WR / 2 : WR), cb * BN, cout, *w_inv_scale,
                                 scale, shift, residual, ys, row_end, relu, half_tile ? RG / 2 : RG, rowmap);
  if (TRACE) {
    __syncthreads();
    if (tid == 0) {
      unsigned hw_id, xcc_id;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
      long long* t = trace + (size_t)blockIdx.x * 8;
      t[0] = t_entry; t[1] = t_pro; t[2] = t_loop; t[3] = wall_clock64();
      t[4] = nsteps; t[5] = hw_id; t[6] = xcc_id; t[7] = (long long)row0 | ((long long)half_tile << 32);
    }
  }
}

// ------------------------------------------------------------------------------------------ format kernels
__global__ void f32_to_split_kernel(const float* __restrict__ x, size_t n8, uint4* __restrict__ xs) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const f32x8 v = *reinterpret_cast<const f32x8*>(x + i * 8);
  uint4 hi, lo;
  split8(v, hi, lo);
  const size_t o = (i >> 2) * 8 + (i & 3);   // element counts are multiples of 32: chunks never straddle rows
  xs[o] = hi;
  xs[o + 4] = lo;
}

__global__ void split_to_f32_kernel(const uint4* __restrict__ xs, size_t n8, float* __restrict__ x) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const size_t o = (i >> 2) * 8 + (i & 3);
  *reinterpret_cast<f32x8*>(x + i * 8) = join8(xs[o], xs[o + 4]);
}

// half format: [N][C] f16 row-major (C a multiple of 8), 16 bytes per 8-channel unit
__global__ void f32_to_half_kernel(const float* __restrict__ x, size_t n8, uint4* __restrict__ xh) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const f32x8 v = *reinterpret_cast<const f32x8*>(x + i * 8);
  const h8 h = __builtin_convertvector(v, h8);
  xh[i] = *reinterpret_cast<const uint4*>(&h);
}

__global__ void half_to_f32_kernel(const uint4* __restrict__ xh, size_t n8, float* __restrict__ x) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const uint4 u = xh[i];
  const h8 h = *reinterpret_cast<const h8*>(&u);
  *reinterpret_cast<f32x8*>(x + i * 8) = __builtin_convertvector(h, f32x8);
}

__global__ void absmax_kernel(const float* __restrict__ w, size_t n, unsigned* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float m = 0.f;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// packed16[tap][chunk][ntile][hi|lo][lane][8 halves], lane = (col j = lane&15, k-group = lane>>4):
//   element jj = W[tap][32*chunk + 8*(lane>>4) + jj][16*ntile + (lane&15)] * 2^sw
// header (after the data): float inv_scale = 2^-sw
// transposed: `w` is the per-tap TRANSPOSE of the filter that is packed -- [K][cout][cin], i.e. the forward filter when the
// packed one is the data gradient's (dX = dY W_k^T): no transposed copy of the weights per layer and step
__global__ void pack_filters16_kernel(const float* __restrict__ w, int K, int cin, int cout,
                                      const unsigned* __restrict__ absmax_bits, uint4* __restrict__ packed,
                                      float* __restrict__ header, int transposed) {
  const float amax = __uint_as_float(*absmax_bits);
  int e = 0;
  if (amax > 0.f) (void)frexpf(amax, &e);
  const int sw = amax > 0.f ? 13 - e : 0;  // max|w|*2^sw in [2^12, 2^13)
  const float s = ldexpf(1.f, sw);
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) header[0] = ldexpf(1.f, -sw);
  const int nch = cin >> 5, ntiles = cout >> 4;
  const long long total = (long long)K * nch * ntiles * 64;
  if (t >= total) return;
  const int lane = (int)(t & 63);
  long long r = t >> 6;
  const int nt = (int)(r % ntiles); r /= ntiles;
  const int ch = (int)(r % nch);
  const int k = (int)(r / nch);
  f32x8 v;
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    const int ci = 32 * ch + 8 * (lane >> 4) + jj, co = 16 * nt + (lane & 15);
    v[jj] = (transposed ? w[((size_t)k * cout + co) * cin + ci] : w[((size_t)k * cin + ci) * cout + co]) * s;
  }
  uint4 hi, lo;
  split8(v, hi, lo);
  const size_t base = (((size_t)k * nch + ch) * ntiles + nt) * 128;
  packed[base + lane] = hi;
  packed[base + 64 + lane] = lo;
}

bool sparse_conv_f16x3_supported(int c_in, int c_out) {
  return (c_in == 32 || c_in == 64 || c_in == 128 || c_in == 256) &&
         (c_out == 32 || c_out == 64 || c_out == 128 || c_out == 256);
}

template <int CIN, int NT, int RG, int NW, int MODE = 0>
static int launch16(bool balance, bool table /* `order` is a tile table (conv16_table_part), not a permutation */, const uint4* xs, const uint4* wpk, const float* winv, int K, int cout,
                    const int32_t* nbr, int nbr_stride, int n_out, const float* scale, const float* shift,
                    const uint4* residual, int relu, uint4* ys, hipStream_t st, const int32_t* order,
                    Conv16LaunchInfo* query, long long* trace = nullptr, const int32_t* rowmap = nullptr) {
  using S = Conv16Smem<NT, RG, Conv16Step<CIN, NT>::KCH, NW>;
  auto kern = spconv_f16x3_kernel<CIN, NT, RG, NW, MODE>;
  static std::atomic<int> wgs_per_cu{0}, cus_per_xcd{0};   // of this instantiation on this device family
  if (wgs_per_cu.load(std::memory_order_acquire) == 0) {
    if (S::bytes > 48 * 1024)
      ISF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, S::bytes));
    int dev = 0, cus = 0, occ = 0;
    ISF_HIP_TRY(hipGetDevice(&dev));
    ISF_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    ISF_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64 * NW, S::bytes));
    cus_per_xcd.store(cus >= 8 ? cus / 8 : 1, std::memory_order_relaxed);
    wgs_per_cu.store(occ > 0 ? occ : 1, std::memory_order_release);
  }
  const int ncb = cout / (16 * NT);
  ISF_REQUIRE(ncb == 1 || ncb == 2, ISF_ERR_UNSUPPORTED, "sparse_conv16: %d column blocks", ncb);
  Conv16Plan plan = conv16_plan(n_out, S::TM, ncb, wgs_per_cu.load(std::memory_order_relaxed),
                                cus_per_xcd.load(std::memory_order_relaxed), balance && RG > 1);
  if (table && !query) plan = Conv16Plan{wgs_per_cu.load(std::memory_order_relaxed) * cus_per_xcd.load(std::memory_order_relaxed),
                                         -1, plan.part_rows};
  if (query) {   // what this launch would look like (conv16_tile_order_impl works on exactly these tiles)
    *query = Conv16LaunchInfo{plan.full, plan.half, plan.part_rows, S::TM, ncb, wgs_per_cu.load(std::memory_order_relaxed),
                              cus_per_xcd.load(std::memory_order_relaxed)};
    return ISF_OK;
  }
  float* ks_scratch = nullptr;
  unsigned* ks_count = nullptr;
  int grid = conv16_grid_blocks(plan);
  if constexpr ((MODE & 524288) != 0) {   // chunk split: two workgroups per tile, their exchange buffers from the workspace
    ISF_TRY(ksplit_buffers(arena_for_stream(st), (size_t)grid * 2 * NW * RG * NT * 1024, grid, &ks_scratch, &ks_count, st));
    grid *= 2;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), S::bytes, st, xs, nbr, nbr_stride, wpk, winv, K,
                     cout, scale, shift, residual, ys, n_out, relu, plan, order, trace, rowmap, ks_scratch, ks_count);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

// Workgroup shape per layer, from the measured sweep (profiles/r02_call1_knockout_variants.txt, B=4 x 300 k points):
// 8 waves x 32 rows (256-row tiles, two workgroups per CU) for the layers with exactly 128 output columns
// (128 -> 128: 1.13 (4 waves) / 1.02 (16 waves) -> 0.94 ms per 4 launches; 64 -> 128: 0.142 -> 0.130 ms); everything
// else on 4 waves x 32 rows: the narrow layers lose with wider workgroups (fewer independent workgroups to hide the
// gather latency) and the 256-column layers of the small deep levels do not have enough tiles (8 waves: 1.41 -> 1.67 ms;
// 12 waves = one 384-row workgroup per CU: 1.42 -> 1.55 ms, profiles/r02_call3_sharing.txt).
// SMALL LAUNCHES (round 5, profiles/r05_small_launch_ab.txt).  Below these row counts conv16_plan cuts a launch into HALF
// tiles only (every compute unit still gets a workgroup): on the two-group kernel a half tile leaves half of each wave's
// accumulators and of its prefetch registers idle.  The one-group instantiation (RG = 1: the same 16 rows per wave, the
// same tiles, the same order of operations per output element -- bit-identical) runs them 16-19 % faster: 256 -> 256 at
// 20 k rows 0.908 -> 0.764 ms per five launches, 128 -> 256 0.106 -> 0.088, 128 -> 128 at 30 k rows 0.434 -> 0.382 ms per four;
// single-sweep line 454 -> 507 frames/s, two sweeps 706 -> 765.  Above the bound full tiles win by as much (40 k rows:
// 1.22 -> 1.48 ms; 128 -> 128 at 119 k rows 0.835 -> 0.98 ms), so the bound is exactly where the plan stops being all-half:
// rows <= 3 workgroups x 4 groups x 16 rows x CUs / 2 column blocks (24 576 on 256 CUs) for the 256-column layers,
// rows <= 2 x 8 x 16 x CUs (65 536) for the 8-wave 128-column shape.
static int conv16_device_cus() {
  static std::atomic<int> cus{0};
  int c = cus.load(std::memory_order_acquire);
  if (c == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0)
      c = 256;
    cus.store(c, std::memory_order_release);
  }
  return c;
}

template <int CIN, int NT>
static int launch16_rows(int mode, const uint4* xs, const uint4* wpk, const float* winv, int K, int cout,
                         const int32_t* nbr, int nbr_stride, int n_out, const float* scale, const float* shift,
                         const uint4* residual, int relu, uint4* ys, hipStream_t st, const int32_t* order,
                         Conv16LaunchInfo* query, const int32_t* rowmap = nullptr) {
  // mode bit 64 (A/B): the 128-column layers of the large levels on the 4-wave 128-row tile instead of the 8-wave 256-row one
  const bool narrow_tiles = (mode & 64) != 0;
  mode &= ~64;
  // mode bit 32768 (opt-in, bit-identical, measured 8 % slower: profiles/r06_deep.txt): the deep layers' 4-wave launches on
  // isf_spconv_deep.hip (LDS-DMA gathers + one instruction stream per step)
  const bool use_deep = (mode & 32768) != 0;
  mode &= ~32768;
#define ISF_ARGS16 (mode & 32) == 0, (mode & 1024) != 0 && order != nullptr, xs, wpk, winv, K, cout, nbr, nbr_stride, n_out, scale, shift, residual, relu, ys, st, order, query, nullptr, rowmap
  // mode bits 4096 / 8192 (round 5 experiment, valid results, bit-identical): the 256-COLUMN layers as ONE column block
  // -- a workgroup owns all 256 output columns of its rows, so a row is gathered ONCE per tap and chunk instead of once per
  // column block.  The phase trace (profiles/r05_att_256.txt) shows the step bound by the vector-memory issue path (a
  // gather in the MFMA operand layout costs 64 address cycles, 12 resident waves x (3 gathers + 4 weight DMA pieces) =
  // the step time): per MFMA this halves the gather instructions.  4096: 4 waves x 32 rows (two workgroups per CU:
  // 64 KiB weight stage); 8192: 8 waves x 16 rows.  Tile-order tables belong to the two-block launch plan: ignored.
  // mode bit 524288: CHUNK SPLIT for the 256-column layers (see the kernel); the workgroup shape follows the small-launch
  // rule below (the choice does not change the arithmetic); other shapes / modes ignore the bit
  if constexpr (NT == 8 && CIN >= 128) {
    if ((mode & 524288) && cout == 256 && (mode & ~(32 | 1024 | 524288)) == 0) {
      if (n_out <= 96 * conv16_device_cus()) return launch16<CIN, NT, 1, 4, 524288>(ISF_ARGS16);
      return launch16<CIN, NT, 2, 4, 524288>(ISF_ARGS16);
    }
  }
  mode &= ~524288;
  if constexpr (NT == 8 && CIN >= 128) {
    if ((mode & (4096 | 8192)) && cout == 256 && (mode & ~(32 | 1024 | 4096 | 8192)) == 0 && !rowmap) {
      if (mode & 8192)
        return launch16<CIN, 16, 1, 8>(false, false, xs, wpk, winv, K, cout, nbr, nbr_stride, n_out, scale, shift, residual,
                                       relu, ys, st, nullptr, query);
      return launch16<CIN, 16, 2, 4>((mode & 32) == 0, false, xs, wpk, winv, K, cout, nbr, nbr_stride, n_out, scale, shift,
                                     residual, relu, ys, st, nullptr, query);
    }
  }
  mode &= ~(4096 | 8192);
  if constexpr (NT == 8 && CIN >= 128) {   // mode bit 262144: gathered rows two steps ahead (A2 loop; 4-wave shapes)
    if ((mode & ~(32 | 1024)) == 262144 && !(cout == 128 && n_out >= 8 * 256)) return launch16<CIN, NT, 2, 4, 262144>(ISF_ARGS16);
  }
  mode &= ~262144;
  if constexpr (NT == 8) {   // mode bit 131072: round 4's issue phase (index reads in the step, four separate DMA pieces): A/B
    if ((mode & ~(32 | 1024)) == 131072) {
      if (cout == 128 && n_out >= 8 * 256) return launch16<CIN, NT, 2, 8, 131072>(ISF_ARGS16);
      return launch16<CIN, NT, 2, 4, 131072>(ISF_ARGS16);
    }
  }
  mode &= ~131072;
  if constexpr (NT == 8) {   // mode bit 65536: staggered issue phases (deep layers; valid results, bit-identical)
    if ((mode & ~(32 | 1024)) == 65536) {
      if (cout == 128 && n_out >= 8 * 256) return launch16<CIN, NT, 2, 8, 65536>(ISF_ARGS16);
      return launch16<CIN, NT, 2, 4, 65536>(ISF_ARGS16);
    }
  }
  mode &= ~65536;
  switch (mode & ~(32 | 1024)) {   // single-pass f16 (opt-in) and the timing diagnostics run on the 4-wave shape
    case 0: break;
    case 1: return launch16<CIN, NT, 2, 4, 1>(ISF_ARGS16);
    case 2: return launch16<CIN, NT, 2, 4, 2>(ISF_ARGS16);
    case 4: return launch16<CIN, NT, 2, 4, 4>(ISF_ARGS16);
    case 6: return launch16<CIN, NT, 2, 4, 6>(ISF_ARGS16);
    case 8: return launch16<CIN, NT, 2, 4, 8>(ISF_ARGS16);
    case 257:   // f16 storage (+ single-pass f16 arithmetic), production workgroup shapes
      if (NT == 8 && cout == 128 && n_out >= 8 * 256) return launch16<CIN, (NT == 8 ? NT : 2), 2, 8, 257>(ISF_ARGS16);
      return launch16<CIN, NT, 2, 4, 257>(ISF_ARGS16);
    case 16:   // no neighbour sharing, production workgroup shapes
      if (NT == 8 && cout == 128 && n_out >= 8 * 256) return launch16<CIN, (NT == 8 ? NT : 2), 2, 8, 16>(ISF_ARGS16);
      return launch16<CIN, NT, 2, 4, 16>(ISF_ARGS16);
    default:
      ISF_REQUIRE(false, ISF_ERR_ARG, "sparse_conv16: mode %d (single-pass precision and the diagnostics {2,4,6,8} "
                  "are not combinable)", mode);
  }
  if constexpr (NT == 8) {   // small launches: one 16-row group per wave (see conv16_device_cus above)
    if (cout == 128 && n_out >= 8 * 256 && n_out <= 256 * conv16_device_cus()) return launch16<CIN, NT, 1, 8>(ISF_ARGS16);
  }
  if (NT == 8 && cout == 128 && n_out >= 8 * 256 && !narrow_tiles)
    return launch16<CIN, (NT == 8 ? NT : 2), 2, 8>(ISF_ARGS16);
  if constexpr (NT == 8 && CIN >= 128) {
    if (cout == 256 && n_out <= 96 * conv16_device_cus()) return launch16<CIN, NT, 1, 4>(ISF_ARGS16);
  }
  if constexpr (NT == 8 && CIN >= 128) {
    if (use_deep && !rowmap && (mode & ~(32 | 1024)) == 0 && sparse_conv_deep_supported(CIN, cout))
      return sparse_conv_forward_deep_impl((mode & 32) == 0, (mode & 1024) != 0 && order != nullptr, xs, CIN, wpk, winv, K, cout,
                                           nbr, nbr_stride, n_out, scale, shift, residual, relu, ys, st, order, query);
  }
  return launch16<CIN, NT, 2, 4>(ISF_ARGS16);
#undef ISF_ARGS16
}

template <int CIN>
static int dispatch16(int mode, const uint4* xs, const uint4* wpk, const float* winv, int K, int cout,
                      const int32_t* nbr, int nbr_stride, int n_out, const float* scale, const float* shift,
                      const uint4* residual, int relu, uint4* ys, hipStream_t st, const int32_t* order,
                      Conv16LaunchInfo* query, const int32_t* rowmap) {
  switch (cout) {
    case 32:  return launch16_rows<CIN, 2>(mode, xs, wpk, winv, K, cout, nbr, nbr_stride, n_out, scale, shift, residual, relu, ys, st, order, query, rowmap);
    case 64:  return launch16_rows<CIN, 4>(mode, xs, wpk, winv, K, cout, nbr, nbr_stride, n_out, scale, shift, residual, relu, ys, st, order, query, rowmap);
    case 128:
    case 256: return launch16_rows<CIN, 8>(mode, xs, wpk, winv, K, cout, nbr, nbr_stride, n_out, scale, shift, residual, relu, ys, st, order, query, rowmap);
  }
  return ISF_ERR_UNSUPPORTED;
}

// packed16 = K*Cin*Cout*4 bytes of fragments followed by a 64-byte header
int sparse_conv_forward_f16x3_impl(const void* xs, int c_in, const void* packed16, int K, int c_out,
                                   const int32_t* nbr, int nbr_stride, int n_out, const float* scale,
                                   const float* shift, const void* residual, int relu, void* ys, int mode,
                                   hipStream_t st, const int32_t* order, Conv16LaunchInfo* query, const int32_t* rowmap) {
  if (n_out <= 0) {
    if (query) *query = Conv16LaunchInfo{0, 0, 0, 0, 0, 0, 0};
    return ISF_OK;
  }
  ISF_REQUIRE(K >= 1 && K <= kMaxTaps, ISF_ERR_UNSUPPORTED, "sparse_conv16: %d taps (max 27)", K);
  ISF_REQUIRE(sparse_conv_f16x3_supported(c_in, c_out), ISF_ERR_UNSUPPORTED,
              "sparse_conv16: (Cin,Cout)=(%d,%d) not built", c_in, c_out);
  ISF_REQUIRE(nbr_stride % 128 == 0 && nbr_stride >= n_out, ISF_ERR_ARG, "sparse_conv16: bad nbr_stride");
  const uint4* w = reinterpret_cast<const uint4*>(packed16);
  const float* winv = reinterpret_cast<const float*>(reinterpret_cast<const char*>(packed16) +
                                                     (size_t)K * c_in * c_out * 4);
  const uint4* x = reinterpret_cast<const uint4*>(xs);
  const uint4* r = reinterpret_cast<const uint4*>(residual);
  uint4* y = reinterpret_cast<uint4*>(ys);
  switch (c_in) {
    case 32:  return dispatch16<32>(mode, x, w, winv, K, c_out, nbr, nbr_stride, n_out, scale, shift, r, relu, y, st, order, query, rowmap);
    case 64:  return dispatch16<64>(mode, x, w, winv, K, c_out, nbr, nbr_stride, n_out, scale, shift, r, relu, y, st, order, query, rowmap);
    case 128: return dispatch16<128>(mode, x, w, winv, K, c_out, nbr, nbr_stride, n_out, scale, shift, r, relu, y, st, order, query, rowmap);
    case 256: return dispatch16<256>(mode, x, w, winv, K, c_out, nbr, nbr_stride, n_out, scale, shift, r, relu, y, st, order, query, rowmap);
  }
  return ISF_ERR_UNSUPPORTED;
}

// per-workgroup trace of one launch (production workgroup shape of the two deep channel shapes): where the time of a
// launch goes -- tools/conv_trace.py
int sparse_conv_trace_impl(const void* xs, int c_in, const void* packed16, int K, int c_out, const int32_t* nbr,
                           int nbr_stride, int n_out, const float* scale, const float* shift, const void* residual,
                           int relu, void* ys, const int32_t* order, long long* trace, int trace_capacity_blocks,
                           int* grid_blocks, hipStream_t st, bool phase = false, size_t phase_capacity_bytes = 0,
                           int* waves_per_block = nullptr) {
  ISF_REQUIRE((c_in == 256 && c_out == 256) || (c_in == 128 && c_out == 128), ISF_ERR_UNSUPPORTED,
              "sparse_conv_trace: built for 128 -> 128 and 256 -> 256, got %d -> %d", c_in, c_out);
  ISF_REQUIRE(K >= 1 && K <= kMaxTaps && n_out > 0 && nbr_stride % 128 == 0 && nbr_stride >= n_out, ISF_ERR_ARG,
              "sparse_conv_trace: bad arguments");
  const uint4* w = reinterpret_cast<const uint4*>(packed16);
  const float* winv = reinterpret_cast<const float*>(reinterpret_cast<const char*>(packed16) + (size_t)K * c_in * c_out * 4);
  const uint4* x = reinterpret_cast<const uint4*>(xs);
  const uint4* r = reinterpret_cast<const uint4*>(residual);
  uint4* y = reinterpret_cast<uint4*>(ys);
  Conv16LaunchInfo info;
  const bool wide = c_out == 128 && n_out >= 8 * 256;
  for (int pass = 0; pass < 2; ++pass) {   // pass 0: the launch shape (grid size), pass 1: launch
    Conv16LaunchInfo* q = pass == 0 ? &info : nullptr;
    if (pass == 1)   // the kernel writes trace + 8 * blockIdx.x for every workgroup of the grid (unchecked before)
      ISF_REQUIRE(8 * (info.full + info.half) <= trace_capacity_blocks, ISF_ERR_ARG,
                  "sparse_conv_trace: the launch has %d workgroups, the trace buffer holds %d", 8 * (info.full + info.half),
                  trace_capacity_blocks);
    int rc;
    if (phase) {   // the per-wave phase records sit behind the per-workgroup records: [grid][8] int64, then dwords
      const int nw = (c_in == 256 || !wide) ? 4 : 8;
      if (waves_per_block) *waves_per_block = nw;
      if (pass == 1)
        ISF_REQUIRE((size_t)8 * (info.full + info.half) * (64 + (size_t)nw * (kPhaseHdr + kPhaseStep * kPhaseMaxSteps) * 4) <=
                        phase_capacity_bytes, ISF_ERR_ARG, "sparse_conv_phase_trace: trace buffer too small");
      if (c_in == 256) rc = launch16<256, 8, 2, 4, 512 | 2048>(true, false, x, w, winv, K, c_out, nbr, nbr_stride, n_out, scale, shift, r, relu, y, st, order, q, trace);
      else if (wide) rc = launch16<128, 8, 2, 8, 512 | 2048>(true, false, x, w, winv, K, c_out, nbr, nbr_stride, n_out, scale, shift, r, relu, y, st, order, q, trace);
      else rc = launch16<128, 8, 2, 4, 512 | 2048>(true, false, x, w, winv, K, c_out, nbr, nbr_stride, n_out, scale, shift, r, relu, y, st, order, q, trace);
    } else if (c_in == 256) rc = launch16<256, 8, 2, 4, 512>(true, false, x, w, winv, K, c_out, nbr, nbr_stride, n_out, scale, shift, r, relu, y, st, order, q, trace);
    else if (wide) rc = launch16<128, 8, 2, 8, 512>(true, false, x, w, winv, K, c_out, nbr, nbr_stride, n_out, scale, shift, r, relu, y, st, order, q, trace);
    else rc = launch16<128, 8, 2, 4, 512>(true, false, x, w, winv, K, c_out, nbr, nbr_stride, n_out, scale, shift, r, relu, y, st, order, q, trace);
    ISF_TRY(rc);
  }
  *grid_blocks = 8 * (info.full + info.half);
  return ISF_OK;
}

// ------------------------------------------------------------------------------------------------- tile order
// A launch that is resident in one round is as slow as its busiest CU: the matrix pipe of a CU is shared by the tiles
// the dispatcher placed on it (slots j, j + cus, j + 2 cus of the XCD: round-robin, tools/probes/wg_placement.hip),
// and the work of a tile -- the (16-row group, tap) pairs that have a neighbour, which is what it issues MFMAs for --
// varies 3x between the sparse rim and the dense middle of a LiDAR sweep.  In launch order the tiles of one CU are
// ~4000 rows apart and add up unevenly: busiest CU 1.23x (256 -> 256) / 1.29x (128 -> 128) the mean on the benchmark
// geometry.  conv16_tile_order_impl hands out the SAME tiles longest-first to the least-loaded CU with a free slot
// (LPT): 1.06x / 1.15x in the same model (profiles/r03_tile_order_lpt.txt).  The result of a layer does not change (the
// same tiles compute the same rows in the same order of operations); only which workgroup slot runs which tile.
__global__ __launch_bounds__(256) void conv16_tile_work_kernel(const int32_t* __restrict__ nbr, int nbr_stride, int K,
                                                               int n_out, Conv16Plan plan, int TM,
                                                               int32_t* __restrict__ work) {
  const int tiles = plan.full + plan.half;
  const int part = blockIdx.x / tiles, t = blockIdx.x - part * tiles;
  __shared__ int total;
  if (threadIdx.x == 0) total = 0;
  __syncthreads();
  int row0 = 0, row_end = 0;
  bool half = false;
  int mine = 0;
  if (conv16_tile_rows(plan, TM, n_out, part, t, row0, row_end, half)) {
    const int rows = half ? TM / 2 : TM;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int seg = 0; seg * 64 < rows; ++seg) {          // 64 rows = four 16-row groups per ballot
      const int row = row0 + seg * 64 + lane;
      const bool in = seg * 64 + lane < rows && row < row_end;
      for (int k = wave; k < K; k += 4) {
        const bool has = in && nbr[(size_t)k * nbr_stride + row] >= 0;
        const unsigned long long m = __ballot(has);
        mine += ((m & 0xffffull) != 0) + ((m & 0xffff0000ull) != 0) + ((m & 0xffff00000000ull) != 0) +
                ((m & 0xffff000000000000ull) != 0);
      }
    }
    if (lane == 0 && mine) atomicAdd(&total, mine);
  }
  __syncthreads();
  if (threadIdx.x == 0) work[blockIdx.x] = total;
}

// the same count from a line-compressed table's tap masks (isf_rulebook.hip): a group's taps = the OR of its rows' masks
__global__ __launch_bounds__(256) void conv16_tile_work_mask_kernel(const uint32_t* __restrict__ lmask, int n_out,
                                                                    Conv16Plan plan, int TM, int32_t* __restrict__ work) {
  const int tiles = plan.full + plan.half;
  const int part = blockIdx.x / tiles, t = blockIdx.x - part * tiles;
  __shared__ int total;
  if (threadIdx.x == 0) total = 0;
  __syncthreads();
  int row0 = 0, row_end = 0;
  bool half = false;
  if (conv16_tile_rows(plan, TM, n_out, part, t, row0, row_end, half)) {
    const int rows = half ? TM / 2 : TM;
    int mine = 0;
    for (int r = threadIdx.x; r < rows; r += 256) {      // 16 consecutive lanes = one 16-row group
      const int row = row0 + r;
      unsigned m = row < row_end ? lmask[row] : 0u;
#pragma unroll
      for (int d = 8; d >= 1; d >>= 1) m |= (unsigned)__shfl_xor((int)m, d, 64);
      if ((threadIdx.x & 15) == 0) mine += __popc(m);
    }
    if (mine) atomicAdd(&total, mine);
  }
  __syncthreads();
  if (threadIdx.x == 0) work[blockIdx.x] = total;
}

// one wave per part: longest tile first, to the CU with the least work among those with a free slot
__global__ __launch_bounds__(64) void conv16_tile_order_kernel(const int32_t* __restrict__ work, int tiles, int cus,
                                                               int32_t* __restrict__ order) {
  const int part = blockIdx.x, lane = threadIdx.x;
  // four candidate tiles per lane (tiles <= 255); key = work << 8 | (255 - tile): ties go to the lower tile index and
  // no live key is 0 (= taken)
  unsigned key[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int t = lane + 64 * q;
    key[q] = t < tiles ? ((unsigned)work[part * tiles + t] << 8) | (unsigned)(255 - t) : 0u;
  }
  const int cap = lane < cus ? (tiles - lane + cus - 1) / cus : 0;   // lane c < cus is CU c: slots c, c + cus, ...
  int used = 0;
  unsigned load = 0;
  for (int it = 0; it < tiles; ++it) {
    unsigned best = key[0];
#pragma unroll
    for (int q = 1; q < 4; ++q) best = key[q] > best ? key[q] : best;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const unsigned other = (unsigned)__shfl_xor((int)best, o);
      best = other > best ? other : best;
    }
    const int tile = 255 - (int)(best & 255u);
    const unsigned w = best >> 8;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (key[q] == best) key[q] = 0u;   // keys are unique (the tile index is part of them)
    unsigned ck = used < cap ? (load << 6) | (unsigned)lane : 0xffffffffu;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const unsigned other = (unsigned)__shfl_xor((int)ck, o);
      ck = other < ck ? other : ck;
    }
    const int cu = (int)(ck & 63u);
    if (lane == cu) {
      order[part * tiles + cu + cus * used] = tile;
      ++used;
      load += w;
    }
  }
}

// one thread per part walks conv16_table_part (<= a few thousand groups: ~20 us, on the geometry stream)
__global__ void conv16_tile_table_kernel(const int32_t* __restrict__ work, int n_groups, int part_groups, int cus,
                                         int wgs_per_cu, int gt, int32_t* __restrict__ tiles) {
  const int part = blockIdx.x;
  if (threadIdx.x != 0) return;
  const int G0 = part * part_groups < n_groups ? part * part_groups : n_groups;
  const int G1 = (part + 1) * part_groups < n_groups ? (part + 1) * part_groups : n_groups;
  (void)conv16_table_part(work, G0, G1, cus, wgs_per_cu, gt, tiles + (size_t)part * 2 * wgs_per_cu * cus);   // fits: checked
}                                                                                                               // on the host

// tile table of a launch (info from the launch query); group_work [ceil(n_out / 16)] from conv_group_masks_impl;
// table: conv16_table_ints(info) int32.  Only for launches conv16_table_applies() accepts.
int conv16_tile_table_impl(const int32_t* group_work, int n_out, const Conv16LaunchInfo& info, int32_t* table,
                           hipStream_t st) {
  ISF_REQUIRE(conv16_table_applies(info), ISF_ERR_ARG, "tile table: the launch is not one resident round");
  const int parts = conv16_order_parts(info);
  hipLaunchKernelGGL(conv16_tile_table_kernel, dim3(parts), dim3(64), 0, st, group_work, ceil_div(n_out, 16),
                     info.part_rows / 16, info.cus_per_xcd, info.wgs_per_cu, info.TM / 16, table);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int conv16_tile_order_impl(const int32_t* nbr, int nbr_stride, int K, int n_out, const Conv16LaunchInfo& info,
                           int32_t* work, int32_t* order, hipStream_t st, const uint32_t* lmask) {
  ISF_REQUIRE(conv16_order_applies(info), ISF_ERR_ARG, "tile order: launch of %d + %d tiles per part on %d x %d slots",
              info.full, info.half, info.wgs_per_cu, info.cus_per_xcd);
  const int parts = conv16_order_parts(info), tiles = conv16_order_tiles(info);
  const Conv16Plan plan{info.full, info.half, info.part_rows};
  if (lmask)
    hipLaunchKernelGGL(conv16_tile_work_mask_kernel, dim3(parts * tiles), dim3(256), 0, st, lmask, n_out, plan, info.TM, work);
  else
    hipLaunchKernelGGL(conv16_tile_work_kernel, dim3(parts * tiles), dim3(256), 0, st, nbr, nbr_stride, K, n_out, plan,
                       info.TM, work);
  hipLaunchKernelGGL(conv16_tile_order_kernel, dim3(parts), dim3(64), 0, st, work, tiles, info.cus_per_xcd, order);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

// ------------------------------------------------------------------------------------------- row sort (round 6)
__global__ __launch_bounds__(256) void conv_row_key_kernel(const int32_t* __restrict__ nbr, int nbr_stride, int K, int n_out,
                                                           int part_rows, int key_mode, uint32_t* __restrict__ keys) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n_out) return;
  uint32_t m = 0u;
  for (int k = 0; k < K; ++k) m |= (nbr[(size_t)k * nbr_stride + r] >= 0 ? 1u : 0u) << k;
  // What decides the order (CPU census of the benchmark geometry, profiles/r06_row_sort.txt): for each of the two
  // neighbouring planes (kz = 0 / kz = 2), WHICH OF ITS THREE ky ROWS holds a neighbour at all -- 6 bits that say which
  // tap lines a tile will have to walk -- then (key_mode 2) the nine in-plane taps.  Tile-taps at level 3: 7 298 unsorted,
  // 6 190 by all 27 bits in numeric order, 5 836 by the 6 coarse bits alone (ONE radix pass), 5 748 by coarse | in-plane.
  // Rows stay inside their part (= their XCD's row range).
  const uint32_t top = (m >> 18) & 0x1ffu, bot = m & 0x1ffu;
  const uint32_t coarse = ((top & 7u) ? 8u : 0u) | ((top & 0x38u) ? 16u : 0u) | ((top & 0x1c0u) ? 32u : 0u) |
                          ((bot & 7u) ? 1u : 0u) | ((bot & 0x38u) ? 2u : 0u) | ((bot & 0x1c0u) ? 4u : 0u);
  const uint32_t part = (uint32_t)(r / part_rows);
  const uint32_t m16 = (((m >> 19) & 0xffu) << 8) | ((m >> 1) & 0xffu);   // key_mode 3: eight taps above, eight below
  keys[r] = key_mode == 2 ? ((part << 15) | (coarse << 9) | ((m >> 9) & 0x1ffu))
                          : (key_mode == 3 ? ((part << 16) | m16) : ((part << 6) | coarse));
}

__global__ __launch_bounds__(256) void conv_row_permute_kernel(const int32_t* __restrict__ nbr, int nbr_stride, int K, int n_out,
                                                               int32_t* __restrict__ rowmap /* in: [n_out]; padded to the stride */,
                                                               int32_t* __restrict__ nbr_sorted) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nbr_stride) return;
  const int r = i < n_out ? rowmap[i] : -1;
  if (i >= n_out) rowmap[i] = i;                         // positions behind the rows: never read (kept in range)
  for (int k = 0; k < K; ++k) nbr_sorted[(size_t)k * nbr_stride + i] = r >= 0 ? nbr[(size_t)k * nbr_stride + r] : -1;
}

int conv_row_sort_impl(Arena& a, const int32_t* nbr, int nbr_stride, int K, int n_out, int part_rows, int32_t* rowmap,
                       int32_t* nbr_sorted, hipStream_t st, int key_mode) {
  ISF_REQUIRE(nbr && rowmap && nbr_sorted && n_out > 0 && K >= 1 && K <= kMaxTaps && part_rows > 0 && nbr_stride >= n_out,
              ISF_ERR_ARG, "conv_row_sort: bad arguments");
  const int parts = ceil_div(n_out, part_rows);
  ISF_REQUIRE(parts <= 32, ISF_ERR_ARG, "conv_row_sort: %d parts", parts);
  int part_bits = 0;
  while ((1 << part_bits) < parts) ++part_bits;
  uint32_t* keys = nullptr;
  ISF_TRY(a.alloc_n(&keys, (size_t)n_out));
  ISF_REQUIRE(K == 27 && key_mode >= 1 && key_mode <= 3, ISF_ERR_ARG, "conv_row_sort: %d taps, key mode %d", K, key_mode);
  hipLaunchKernelGGL(conv_row_key_kernel, dim3(ceil_div(n_out, 256)), dim3(256), 0, st, nbr, nbr_stride, K, n_out, part_rows,
                     key_mode, keys);
  ISF_LAUNCH_CHECK();
  ISF_TRY(stable_sort_u32_impl(a, keys, n_out, (key_mode == 2 ? 15 : key_mode == 3 ? 16 : 6) + part_bits, rowmap, st));
  hipLaunchKernelGGL(conv_row_permute_kernel, dim3(ceil_div(nbr_stride, 256)), dim3(256), 0, st, nbr, nbr_stride, K, n_out,
                     rowmap, nbr_sorted);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

__global__ __launch_bounds__(256) void conv_row_key_lines_kernel(const uint32_t* __restrict__ lmask, int n_out, int part_rows,
                                                                 int key_bits /* 16 | 27 */, uint32_t* __restrict__ keys) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n_out) return;
  const uint32_t m = lmask[r] & 0x7ffffffu;
  const uint32_t mk = key_bits == 27 ? m : ((((m >> 19) & 0xffu) << 8) | ((m >> 1) & 0xffu));
  keys[r] = ((uint32_t)(r / part_rows) << key_bits) | mk;
}

__global__ __launch_bounds__(256) void conv_row_permute_lines_kernel(const int32_t* __restrict__ lines,
                                                                     const uint32_t* __restrict__ lmask, int nbr_stride,
                                                                     int num_lines, int n_out, int32_t* __restrict__ rowmap,
                                                                     int32_t* __restrict__ lines_sorted,
                                                                     uint32_t* __restrict__ lmask_sorted) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nbr_stride) return;
  const int r = i < n_out ? rowmap[i] : -1;
  if (i >= n_out) rowmap[i] = i;
  lmask_sorted[i] = r >= 0 ? lmask[r] : 0u;
  for (int l = 0; l < num_lines; ++l) lines_sorted[(size_t)l * nbr_stride + i] = r >= 0 ? lines[(size_t)l * nbr_stride + r] : -1;
}

int conv_row_sort_lines_impl(Arena& a, const int32_t* lines, const uint32_t* lmask, int nbr_stride, int num_lines, int n_out,
                             int part_rows, bool full_key, int32_t* rowmap, int32_t* lines_sorted, uint32_t* lmask_sorted,
                             hipStream_t st) {
  ISF_REQUIRE(lines && lmask && rowmap && lines_sorted && lmask_sorted && n_out > 0 && num_lines >= 1 && num_lines <= 9 &&
                  part_rows > 0 && nbr_stride >= n_out, ISF_ERR_ARG, "conv_row_sort_lines: bad arguments");
  const int parts = ceil_div(n_out, part_rows);
  ISF_REQUIRE(parts <= 32, ISF_ERR_ARG, "conv_row_sort_lines: %d parts", parts);
  int part_bits = 0;
  while ((1 << part_bits) < parts) ++part_bits;
  const int key_bits = full_key ? 27 : 16;
  uint32_t* keys = nullptr;
  ISF_TRY(a.alloc_n(&keys, (size_t)n_out));
  hipLaunchKernelGGL(conv_row_key_lines_kernel, dim3(ceil_div(n_out, 256)), dim3(256), 0, st, lmask, n_out, part_rows, key_bits, keys);
  ISF_LAUNCH_CHECK();
  ISF_TRY(stable_sort_u32_impl(a, keys, n_out, key_bits + part_bits, rowmap, st));
  hipLaunchKernelGGL(conv_row_permute_lines_kernel, dim3(ceil_div(nbr_stride, 256)), dim3(256), 0, st, lines, lmask, nbr_stride,
                     num_lines, n_out, rowmap, lines_sorted, lmask_sorted);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

// ------------------------------------------------------------------------------------------- band order (round 6)
// A launch of SEVERAL rounds (levels 0 / 1: 2 700 tiles on 768 slots) hands its tiles out in row order = (b, z, y, x)
// order: the ~96 tiles an XCD has in flight are one stretch of ONE z-plane, and the rows their dz = +-1 taps gather --
// the same (y, x) region of the neighbouring planes, ~10 k rows away -- were fetched a few hundred tiles ago or will be a
// few hundred tiles later: the dense planes of a level-1 sweep are 2.5 MB each, three of them do not stay in an XCD's 4-MB
// L2, and every input row comes in from the fabric three times (measured FETCH per 64 -> 64 launch 394 MB against 196
// algorithmic, profiles/r05_v2_traffic.json).  This order deals an XCD's tiles BAND BY BAND instead: all tiles (of every
// plane of the part) whose first row lies in y-band 0, then band 1, ...; the z-neighbours of a tile are then a few tiles
// away in time, its y-neighbours one band later -- both inside the in-flight window.  A permutation of the launch's own
// tiles (read through `order` like the LPT tables): results are bit-identical.
__global__ __launch_bounds__(256) void conv16_band_order_kernel(const int32_t* __restrict__ coors4, int n_out, Conv16Plan plan,
                                                                int TM, int band, int32_t* __restrict__ order) {
  extern __shared__ unsigned long long bkey[];
  const int tiles = plan.full + plan.half, part = blockIdx.x;
  for (int t = threadIdx.x; t < tiles; t += 256) {
    int row0 = 0, row_end = 0;
    bool half = false;
    unsigned long long k = ~0ull;                       // empty tiles go last
    if (conv16_tile_rows(plan, TM, n_out, part, t, row0, row_end, half)) {
      const int4 c = reinterpret_cast<const int4*>(coors4)[row0];     // (b, z, y, x)
      k = ((unsigned long long)(unsigned)(c.z / band) << 40) | ((unsigned long long)(unsigned)c.x << 32) |
          ((unsigned long long)(unsigned)c.y << 20) | (unsigned long long)(unsigned)t;
    }
    bkey[t] = k;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < tiles; t += 256) {
    const unsigned long long k = bkey[t];
    int rank = 0;
    for (int u = 0; u < tiles; ++u) rank += bkey[u] < k || (bkey[u] == k && u < t);
    order[(size_t)part * tiles + rank] = t;
  }
}

bool conv16_band_order_applies(const Conv16LaunchInfo& i) {
  const int t = i.full + i.half;
  return i.half == 0 && t > 2 * i.wgs_per_cu * i.cus_per_xcd && t <= 6000;
}

int conv16_band_order_impl(const int32_t* coors4, int n_out, const Conv16LaunchInfo& info, int band, int32_t* order,
                           hipStream_t st) {
  ISF_REQUIRE(conv16_band_order_applies(info) && band > 0 && coors4 && order, ISF_ERR_ARG, "band order: bad arguments");
  const int parts = conv16_order_parts(info), tiles = conv16_order_tiles(info);
  const Conv16Plan plan{info.full, info.half, info.part_rows};
  hipLaunchKernelGGL(conv16_band_order_kernel, dim3(parts), dim3(256), (size_t)tiles * 8, st, coors4, n_out, plan, info.TM, band,
                     order);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int pack_filters16_impl(Arena& a, const float* w, int K, int cin, int cout, void* packed16, hipStream_t st, bool transposed) {
  unsigned* amax = nullptr;
  ISF_TRY(a.alloc_n(&amax, 64));
  ISF_HIP_TRY(hipMemsetAsync(amax, 0, sizeof(unsigned), st));
  const size_t n = (size_t)K * cin * cout;
  // at most 64 workgroups: every wave ends in an atomicMax on ONE word (~88 memory-side updates per microsecond), and
  // 1024 workgroups x 4 waves made this 19 us per call -- 42 calls per training step (profiles/r05_train_step_after.txt)
  hipLaunchKernelGGL(absmax_kernel, dim3(ceil_div((long long)n, 4096) < 64 ? ceil_div((long long)n, 4096) : 64),
                     dim3(256), 0, st, w, n, amax);
  const long long total = (long long)K * (cin >> 5) * (cout >> 4) * 64;
  hipLaunchKernelGGL(pack_filters16_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, w, K, cin, cout, amax,
                     reinterpret_cast<uint4*>(packed16),
                     reinterpret_cast<float*>(reinterpret_cast<char*>(packed16) + n * 4), transposed ? 1 : 0);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int f32_to_split_impl(const float* x, size_t n_elems, void* xs, hipStream_t st) {
  if (n_elems == 0) return ISF_OK;
  ISF_REQUIRE(n_elems % 32 == 0, ISF_ERR_ARG, "f32_to_split: element count must be a multiple of 32");
  hipLaunchKernelGGL(f32_to_split_kernel, dim3(ceil_div((long long)(n_elems / 8), 256)), dim3(256), 0, st, x,
                     n_elems / 8, reinterpret_cast<uint4*>(xs));
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int f32_to_half_impl(const float* x, size_t n_elems, void* xh, hipStream_t st) {
  if (n_elems == 0) return ISF_OK;
  ISF_REQUIRE(n_elems % 8 == 0, ISF_ERR_ARG, "f32_to_half: element count must be a multiple of 8");
  hipLaunchKernelGGL(f32_to_half_kernel, dim3(ceil_div((long long)(n_elems / 8), 256)), dim3(256), 0, st, x,
                     n_elems / 8, reinterpret_cast<uint4*>(xh));
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int half_to_f32_impl(const void* xh, size_t n_elems, float* x, hipStream_t st) {
  if (n_elems == 0) return ISF_OK;
  ISF_REQUIRE(n_elems % 8 == 0, ISF_ERR_ARG, "half_to_f32: element count must be a multiple of 8");
  hipLaunchKernelGGL(half_to_f32_kernel, dim3(ceil_div((long long)(n_elems / 8), 256)), dim3(256), 0, st,
                     reinterpret_cast<const uint4*>(xh), n_elems / 8, x);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int split_to_f32_impl(const void* xs, size_t n_elems, float* x, hipStream_t st) {
  if (n_elems == 0) return ISF_OK;
  ISF_REQUIRE(n_elems % 32 == 0, ISF_ERR_ARG, "split_to_f32: element count must be a multiple of 32");
  hipLaunchKernelGGL(split_to_f32_kernel, dim3(ceil_div((long long)(n_elems / 8), 256)), dim3(256), 0, st,
                     reinterpret_cast<const uint4*>(xs), n_elems / 8, x);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

}  // namespace isf

extern "C" {

size_t isf_packed_filter16_bytes(int num_taps, int c_in, int c_out) {
  return (size_t)num_taps * (size_t)c_in * (size_t)c_out * 4 + 64;
}

int isf_pack_filters_f16x3(const float* filters, int num_taps, int c_in, int c_out, void* packed16,
                           isf_stream_t stream) {
  ISF_REQUIRE(filters && packed16 && num_taps > 0, ISF_ERR_ARG, "pack_filters_f16x3: bad arguments");
  ISF_REQUIRE(isf::sparse_conv_f16x3_supported(c_in, c_out), ISF_ERR_UNSUPPORTED,
              "pack_filters_f16x3: (Cin,Cout)=(%d,%d) not built", c_in, c_out);
  isf::Arena& a = isf::arena_for_stream(isf::as_stream(stream));
  ISF_TRY(a.reset());
  return isf::pack_filters16_impl(a, filters, num_taps, c_in, c_out, packed16, isf::as_stream(stream));
}

int isf_pack_filters_f16x3_transposed(const float* filters_t, int num_taps, int c_in, int c_out, void* packed16,
                                      isf_stream_t stream) {
  ISF_REQUIRE(filters_t && packed16 && num_taps > 0, ISF_ERR_ARG, "pack_filters_f16x3_transposed: bad arguments");
  ISF_REQUIRE(isf::sparse_conv_f16x3_supported(c_in, c_out), ISF_ERR_UNSUPPORTED,
              "pack_filters_f16x3_transposed: (Cin,Cout)=(%d,%d) not built", c_in, c_out);
  isf::Arena& a = isf::arena_for_stream(isf::as_stream(stream));
  ISF_TRY(a.reset());
  return isf::pack_filters16_impl(a, filters_t, num_taps, c_in, c_out, packed16, isf::as_stream(stream), true);
}

int isf_f32_to_split(const float* x, size_t num_elems, void* xs, isf_stream_t stream) {
  ISF_REQUIRE(num_elems == 0 || (x && xs), ISF_ERR_ARG, "f32_to_split: null pointer");
  return isf::f32_to_split_impl(x, num_elems, xs, isf::as_stream(stream));
}

int isf_f32_to_half(const float* x, size_t num_elems, void* xh, isf_stream_t stream) {
  ISF_REQUIRE(num_elems == 0 || (x && xh), ISF_ERR_ARG, "f32_to_half: null pointer");
  return isf::f32_to_half_impl(x, num_elems, xh, isf::as_stream(stream));
}

int isf_half_to_f32(const void* xh, size_t num_elems, float* x, isf_stream_t stream) {
  ISF_REQUIRE(num_elems == 0 || (x && xh), ISF_ERR_ARG, "half_to_f32: null pointer");
  return isf::half_to_f32_impl(xh, num_elems, x, isf::as_stream(stream));
}

int isf_split_to_f32(const void* xs, size_t num_elems, float* x, isf_stream_t stream) {
  ISF_REQUIRE(num_elems == 0 || (x && xs), ISF_ERR_ARG, "split_to_f32: null pointer");
  return isf::split_to_f32_impl(xs, num_elems, x, isf::as_stream(stream));
}

int isf_sparse_conv_forward_f16x3(const void* features_split, int num_in, int c_in, const void* packed16,
                                  int num_taps, int c_out, const int32_t* nbr, int nbr_stride, int num_out,
                                  const float* scale, const float* shift, const void* residual_split, int relu,
                                  void* out_split, int mode, isf_stream_t stream) {
  ISF_REQUIRE(num_in >= 0 && num_out >= 0 && c_in > 0 && c_out > 0 && num_taps > 0, ISF_ERR_ARG,
              "sparse_conv_forward_f16x3: bad arguments");
  if (num_out == 0) return ISF_OK;
  ISF_REQUIRE(features_split && packed16 && nbr && out_split && ((scale == nullptr) == (shift == nullptr)),
              ISF_ERR_ARG, "sparse_conv_forward_f16x3: null pointer");
  const int m = mode & ~(32 | 64 | 4096 | 8192 | 32768 | 65536 | 131072 | 262144 | 524288);   // bit 32 = uniform tiles (no full / half mix), combinable; 524288 = chunk split (256-column layers); 4096 / 8192 = one
                                               // column block for the 256-column layers (4 x 32-row / 8 x 16-row waves)
  ISF_REQUIRE(mode >= 0 && (m == 0 || m == 1 || m == 2 || m == 4 || m == 6 || m == 8 || m == 16 || m == 257), ISF_ERR_ARG,
              "sparse_conv_forward_f16x3: mode %d (0 default, 1 single-pass f16, 257 f16 storage, diagnostics 2 / 4 / 6 / "
              "8 / 16, +32)", mode);
  return isf::sparse_conv_forward_f16x3_impl(features_split, c_in, packed16, num_taps, c_out, nbr, nbr_stride,
                                             num_out, scale, shift, residual_split, relu, out_split, mode,
                                             isf::as_stream(stream));
}

int isf_sparse_conv_tile_order(const int32_t* nbr, int nbr_stride, int num_taps, int num_out, int c_in, int c_out,
                               int mode, int32_t* work, int32_t* order, int* num_entries, isf_stream_t stream) {
  ISF_REQUIRE(nbr && work && order && num_entries && num_out >= 0, ISF_ERR_ARG, "sparse_conv_tile_order: bad arguments");
  *num_entries = 0;
  if (num_out == 0) return ISF_OK;
  ISF_REQUIRE(isf::sparse_conv_f16x3_supported(c_in, c_out), ISF_ERR_UNSUPPORTED,
              "sparse_conv_tile_order: (Cin,Cout)=(%d,%d) not built", c_in, c_out);
  isf::Conv16LaunchInfo info;
  if (mode & 2048) {   // the table is for isf_sparse_conv_forward_dma: its launch plan differs (workgroups per CU)
    ISF_REQUIRE(isf::sparse_conv_dma_supported(c_in, c_out), ISF_ERR_UNSUPPORTED,
                "sparse_conv_tile_order: (Cin,Cout)=(%d,%d) has no LDS-DMA kernel", c_in, c_out);
    ISF_TRY(isf::sparse_conv_forward_dma_impl(nullptr, c_in, nullptr, num_taps, c_out, nbr, nbr_stride, num_out, nullptr,
                                              nullptr, nullptr, 0, nullptr, mode & ~2048, isf::as_stream(stream), nullptr,
                                              &info));
  } else {
    ISF_TRY(isf::sparse_conv_forward_f16x3_impl(nullptr, c_in, nullptr, num_taps, c_out, nbr, nbr_stride, num_out, nullptr,
                                                nullptr, nullptr, 0, nullptr, mode, isf::as_stream(stream), nullptr, &info));
  }
  if (!isf::conv16_order_applies(info)) return ISF_OK;
  *num_entries = isf::conv16_order_parts(info) * isf::conv16_order_tiles(info);
  return isf::conv16_tile_order_impl(nbr, nbr_stride, num_taps, num_out, info, work, order, isf::as_stream(stream));
}

int isf_sparse_conv_forward_f16x3_ordered(const void* features_split, int num_in, int c_in, const void* packed16,
                                          int num_taps, int c_out, const int32_t* nbr, int nbr_stride, int num_out,
                                          const float* scale, const float* shift, const void* residual_split, int relu,
                                          void* out_split, int mode, const int32_t* order, isf_stream_t stream) {
  ISF_REQUIRE(num_in >= 0 && num_out >= 0 && c_in > 0 && c_out > 0 && num_taps > 0, ISF_ERR_ARG,
              "sparse_conv_forward_f16x3_ordered: bad arguments");
  if (num_out == 0) return ISF_OK;
  ISF_REQUIRE(features_split && packed16 && nbr && out_split && ((scale == nullptr) == (shift == nullptr)),
              ISF_ERR_ARG, "sparse_conv_forward_f16x3_ordered: null pointer");
  const int m = mode & ~(32 | 65536 | 131072 | 262144 | 524288);
  ISF_REQUIRE(mode >= 0 && (m == 0 || m == 1 || m == 16 || m == 257), ISF_ERR_ARG,
              "sparse_conv_forward_f16x3_ordered: mode %d (0, 1, 16, 257, +32, +65536, +131072)", mode);
  return isf::sparse_conv_forward_f16x3_impl(features_split, c_in, packed16, num_taps, c_out, nbr, nbr_stride,
                                             num_out, scale, shift, residual_split, relu, out_split, mode,
                                             isf::as_stream(stream), order);
}

int isf_sparse_conv_tile_table(const int32_t* nbr, int nbr_stride, int num_taps, int num_out, int c_in, int c_out,
                               int mode, int32_t* scratch, int32_t* table, int* num_ints, isf_stream_t stream) {
  ISF_REQUIRE(nbr && scratch && table && num_ints && num_out >= 0, ISF_ERR_ARG, "sparse_conv_tile_table: bad arguments");
  *num_ints = 0;
  if (num_out == 0) return ISF_OK;
  ISF_REQUIRE(isf::sparse_conv_f16x3_supported(c_in, c_out), ISF_ERR_UNSUPPORTED,
              "sparse_conv_tile_table: (Cin,Cout)=(%d,%d) not built", c_in, c_out);
  ISF_REQUIRE(mode == 0 || mode == 1 || mode == 16 || mode == 257, ISF_ERR_ARG, "sparse_conv_tile_table: mode %d", mode);
  isf::Conv16LaunchInfo info;   // the workgroup shape, and with it the table, depends on the mode
  ISF_TRY(isf::sparse_conv_forward_f16x3_impl(nullptr, c_in, nullptr, num_taps, c_out, nbr, nbr_stride, num_out, nullptr,
                                              nullptr, nullptr, 0, nullptr, mode, isf::as_stream(stream), nullptr, &info));
  if (!isf::conv16_table_applies(info)) return ISF_OK;
  // `table` holds 3072 ints (include/isf_hip.h): 8 parts x 2 x (workgroups per CU x CUs per XCD <= 192); a device with
  // more slots per XCD is refused instead of written past the buffer
  ISF_REQUIRE(isf::conv16_table_ints(info) <= 3072, ISF_ERR_UNSUPPORTED,
              "sparse_conv_tile_table: the launch needs %d table ints, the interface holds 3072", isf::conv16_table_ints(info));
  const int ng = isf::ceil_div(num_out, 16);
  ISF_TRY(isf::conv_group_masks_impl(nbr, nbr_stride, num_taps, num_out, scratch, scratch + ng, isf::as_stream(stream)));
  ISF_TRY(isf::conv16_tile_table_impl(scratch + ng, num_out, info, table, isf::as_stream(stream)));
  *num_ints = isf::conv16_table_ints(info);
  return ISF_OK;
}

int isf_sparse_conv_forward_f16x3_tiled(const void* features_split, int num_in, int c_in, const void* packed16,
                                        int num_taps, int c_out, const int32_t* nbr, int nbr_stride, int num_out,
                                        const float* scale, const float* shift, const void* residual_split, int relu,
                                        void* out_split, int mode, const int32_t* table, isf_stream_t stream) {
  ISF_REQUIRE(num_in >= 0 && num_out >= 0 && c_in > 0 && c_out > 0 && num_taps > 0 && table, ISF_ERR_ARG,
              "sparse_conv_forward_f16x3_tiled: bad arguments");
  if (num_out == 0) return ISF_OK;
  ISF_REQUIRE(features_split && packed16 && nbr && out_split && ((scale == nullptr) == (shift == nullptr)),
              ISF_ERR_ARG, "sparse_conv_forward_f16x3_tiled: null pointer");
  ISF_REQUIRE(mode == 0 || mode == 1 || mode == 16 || mode == 257, ISF_ERR_ARG,
              "sparse_conv_forward_f16x3_tiled: mode %d (0, 1, 16, 257)", mode);
  return isf::sparse_conv_forward_f16x3_impl(features_split, c_in, packed16, num_taps, c_out, nbr, nbr_stride, num_out,
                                             scale, shift, residual_split, relu, out_split, mode | 1024,
                                             isf::as_stream(stream), table);
}

// conv16_table_part walked on the host (tests, tools; no device work): work [num_groups] -> tiles [parts][wgs * cus][2]
int isf_sparse_conv_tile_table_host(const int32_t* work, int num_groups, int part_groups, int parts, int cus, int wgs_per_cu,
                                    int groups_per_tile, int32_t* tiles, int* fits) {
  ISF_REQUIRE(work && tiles && fits && num_groups > 0 && part_groups > 0 && parts > 0 && cus > 0 && wgs_per_cu > 0 &&
                  groups_per_tile > 0, ISF_ERR_ARG, "sparse_conv_tile_table_host: bad arguments");
  *fits = 1;
  for (int p = 0; p < parts; ++p) {
    const int G0 = p * part_groups < num_groups ? p * part_groups : num_groups;
    const int G1 = (p + 1) * part_groups < num_groups ? (p + 1) * part_groups : num_groups;
    if (!isf::conv16_table_part(work, G0, G1, cus, wgs_per_cu, groups_per_tile, tiles + (size_t)p * 2 * wgs_per_cu * cus))
      *fits = 0;
  }
  return ISF_OK;
}

int isf_sparse_conv_trace(const void* features_split, int num_in, int c_in, const void* packed16, int num_taps,
                          int c_out, const int32_t* nbr, int nbr_stride, int num_out, const float* scale,
                          const float* shift, const void* residual_split, int relu, void* out_split,
                          const int32_t* order, long long* trace, int trace_capacity_blocks, int* grid_blocks,
                          isf_stream_t stream) {
  ISF_REQUIRE(num_in >= 0 && features_split && packed16 && nbr && out_split && trace && grid_blocks &&
                  trace_capacity_blocks > 0 &&
                  ((scale == nullptr) == (shift == nullptr)), ISF_ERR_ARG, "sparse_conv_trace: bad arguments");
  return isf::sparse_conv_trace_impl(features_split, c_in, packed16, num_taps, c_out, nbr, nbr_stride, num_out, scale,
                                     shift, residual_split, relu, out_split, order, trace, trace_capacity_blocks,
                                     grid_blocks, isf::as_stream(stream));
}

/* DIAGNOSTIC: isf_sparse_conv_trace plus per-WAVE phase stamps (shader clock, s_memtime) of every step of the multiply
 * loop -- top of the step / after s_waitcnt vmcnt(0) / after the barrier / after issuing the next step's loads.  The
 * image ships rocprofv3 without the thread-trace decoder library (rocprofv3 --att: "rocprof-trace-decoder library path
 * not found"), so this is the instruction-level account of the dominant kernel: tools/conv_phase_trace.py. */
int isf_sparse_conv_phase_trace(const void* features_split, int num_in, int c_in, const void* packed16, int num_taps,
                                int c_out, const int32_t* nbr, int nbr_stride, int num_out, const float* scale,
                                const float* shift, const void* residual_split, int relu, void* out_split,
                                const int32_t* order, long long* trace, size_t trace_bytes, int* grid_blocks,
                                int* waves_per_block, int* dwords_per_wave, isf_stream_t stream) {
  ISF_REQUIRE(num_in >= 0 && features_split && packed16 && nbr && out_split && trace && grid_blocks && waves_per_block &&
                  dwords_per_wave && ((scale == nullptr) == (shift == nullptr)), ISF_ERR_ARG,
              "sparse_conv_phase_trace: bad arguments");
  *dwords_per_wave = isf::kPhaseHdr + isf::kPhaseStep * isf::kPhaseMaxSteps;
  return isf::sparse_conv_trace_impl(features_split, c_in, packed16, num_taps, c_out, nbr, nbr_stride, num_out, scale,
                                     shift, residual_split, relu, out_split, order, trace, 1 << 30, grid_blocks,
                                     isf::as_stream(stream), true, trace_bytes, waves_per_block);
}

}  // extern "C"

