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
#include "isf_common.h"

#include <stdlib.h>

#include <type_traits>

namespace isf {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

static constexpr int kMaxTaps = 27;

extern int g_conv_precision;   // isf_encoder.hip: 0 f16x3 split (default), 1 fp32 MFMA, 2 single-pass f16 (opt-in)

// LDS-DMA of 16 B per lane: LDS[lds_base + lane*16] = *gsrc.  Issued through inline asm on purpose: when hipcc
// sees a global_load_lds it drains vmcnt(0) before every later ds_read (it cannot prove the buffers differ),
// which would serialise the next step's weight/activation prefetch behind the current step's MFMAs.  Hidden
// here, the DMA stays in flight during the compute; the loop waits for it explicitly (s_waitcnt vmcnt(0) +
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

// one all-zero 64-byte line: the gather address of rows that have no neighbour through a tap
__device__ uint4 g_zero_line[4];

// narrow layers (CIN <= 64, <= 64 output columns) run all their 32-channel chunks in one step: half / the same
// number of barriers for twice the MFMAs per barrier
template <int CIN, int NT>
struct Conv16Step {
  static constexpr int KCH = (CIN <= 64 && NT <= 4) ? CIN / 32 : 1;      // 32-channel chunks per step
};

// MODE bit 64 (ISF_CONV16_TPS=1, experiment): several TAPS per step for the narrow layers whose step already holds all
// chunks of a tap (CIN <= 64, <= 64 output columns).  Measured (profiles/r01_v8_bench.json): conv<32,32> takes 85 us
// per launch for 2800 workgroups on 768 slots, i.e. ~23 us per 128-row tile whose MFMA work is < 2 us -- every step
// (12-24 MFMAs, ~300 cycles) waits a full memory round trip for gathers issued one step earlier.  Four or two taps
// per step issue that many gathers / weight DMAs back to back and expose the latency once.
template <int CIN, int NT, int MODE>
constexpr int conv16_tps() {
  // register budget (two A sets of TPS * RG * KCH * 2 uint4 next to the accumulators, 168 VGPRs for 3 waves / SIMD):
  // 4 taps only for 32 -> 32; 2 taps for 32 -> 64 and 64 -> 32; 64 -> 64 would spill and stays at one tap
  return ((MODE & 64) != 0 && Conv16Step<CIN, NT>::KCH == CIN / 32 && CIN * NT <= 128) ? (CIN * NT <= 64 ? 4 : 2) : 1;
}

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

// NW waves per workgroup (4 or 16): all of them share one weight stage per step, so the weight bytes a CU pulls
// through its vector memory path per MFMA fall with NW (measured: a modest win for the 128-column layers of the
// large levels only; what bounds the kernel is analysed in DESIGN.md section 5).
// HALF = single-pass mode (isf_set_conv_precision(2)): only the hi halves of activations and weights are fetched and
// multiplied -- plain f16 operands with fp32 accumulation, the accuracy of the reference under fp16 autocast
// (indice_conv_half), one MFMA per product instead of three.  Same buffers, same layouts; outputs are still written
// split.  Never the default: the headline configuration is fp32-class.
// MODE bits 2 / 4 / 8 are TIMING DIAGNOSTICS (ISF_CONV16_DIAG, results are garbage): 2 = no activation gathers
// (A = 0), 4 = no weight DMA, 8 = no main loop (prologue + epilogue only) -- the knock-out decomposition of DESIGN.md
// section 5 as a permanent tool (tools/conv_knockout.sh).  MODE 0 and 1 compile to exactly what they did without them.
template <int CIN, int NT, int RG, int NW, int MODE = 0>
__global__ __launch_bounds__(64 * NW, (NW >= 16 ? 1 : (NT * RG >= 16 ? 2 : 3))) void spconv_f16x3_kernel(
    const uint4* __restrict__ xs, const int32_t* __restrict__ nbr, int nbr_stride,
    const uint4* __restrict__ wpk, const float* __restrict__ w_inv_scale, int K, int cout,
    const float* __restrict__ scale, const float* __restrict__ shift, const uint4* __restrict__ residual,
    uint4* __restrict__ ys, int n_out, int relu, int row_tiles) {
  constexpr bool HALF = (MODE & 1) != 0, NOGATHER = (MODE & 2) != 0, NODMA = (MODE & 4) != 0, NOLOOP = (MODE & 8) != 0;
  // MODE bit 16 (ISF_CONV16_PRIO=1, experiment): raise the wave's issue priority while it is in its MFMA block, so that
  // a wave that has its operands is not starved by waves still issuing loads / address arithmetic on the same SIMD
  constexpr bool PRIO = (MODE & 16) != 0;
  // MODE bit 32 (ISF_CONV16_TEPI=1, experiment): operands swapped in the MFMAs (weights as A, activations as B), so
  // the accumulators hold Y^T: lane (r = lane&15, g = lane>>4) owns FOUR CONSECUTIVE CHANNELS 16nt + 4g .. +3 of output
  // row r.  The epilogue then needs no LDS transpose, no fences and no cross-lane traffic: every lane folds BN, adds
  // its 8-byte halves of the residual, and stores its 4 hi and 4 lo halves (8 B each; the four lanes of a row cover
  // one 32-byte run).  Same products, same summation order as the default.
  constexpr bool TEPI = (MODE & 32) != 0;
  // MODE bit 256 (ISF_CONV16_VEPI=1, experiment): the default epilogue fetches the 8 BN scales / shifts of an item
  // with two 32-byte loads issued together.  In the generated code of the scalar form every channel is a
  // `global_load_dword ; s_waitcnt vmcnt(0)` pair -- 16 dependent round trips per (row, unit) item, 8 items per wave
  // and tile -- which alone accounts for several microseconds of every tile's lifetime.
  constexpr bool VEPI = (MODE & 256) != 0;
  constexpr int KCH = Conv16Step<CIN, NT>::KCH;
  constexpr int TPS = conv16_tps<CIN, NT, MODE>();   // taps per step (1 unless MODE bit 64)
  using S = Conv16Smem<NT, RG, KCH * TPS, NW>;       // the weight ring holds TPS taps per stage
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
  // XCD-aware tile mapping: workgroups are dealt round-robin to the 8 XCDs (each with its own 4 MB L2) in
  // linear-id order.  XCD x works through ONE CONTIGUOUS range of row tiles (rows are (b,z,y,x)-sorted, so the
  // y / z neighbours a tile gathers are rows of tiles the same XCD touches a little earlier or later: its L2
  // holds that sliding window instead of every XCD fetching every row), and with two column blocks only ever
  // on column block x & 1, so that the weights it streams are half of the layer's.
  const int ncb = cout / BN;
  int cb, tile;
  {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    if (ncb == 2) {
      cb = xcd & 1;
      tile = (xcd >> 1) * ((row_tiles + 3) >> 2) + j;
      if (j >= ((row_tiles + 3) >> 2)) return;
    } else {
      cb = 0;
      tile = xcd * ((row_tiles + 7) >> 3) + j;
      if (j >= ((row_tiles + 7) >> 3)) return;
    }
  }
  if (tile >= row_tiles) return;
  const int row0 = tile * TM;
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
      if (i < K * TM && row0 + r < nbr_stride) tmp[it] = nbr[(size_t)k * nbr_stride + row0 + r];
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
  const int nsteps = NOLOOP ? 0 : ntaps * NCG;

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
  uint4 a_nxt[RG][KCH][2];  // [row group][chunk of the step][hi, lo]
  auto load_A = [&](int tap, int cg) {
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      if ((rgm[rg] >> tap) & 1u) {
        const int idx = nbr_l[tap * TM + wave * WR + rg * 16 + col];
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) {
          a_nxt[rg][kc][0] = make_uint4(0, 0, 0, 0);
          a_nxt[rg][kc][1] = make_uint4(0, 0, 0, 0);
        }
        if (idx >= 0 && !NOGATHER) {
#pragma unroll
          for (int kc = 0; kc < KCH; ++kc) {
            const uint4* p = xs + ((size_t)idx * CH8 + (cg * KCH + kc) * 4) * 2 + kg;   // chunk base + k-group
            a_nxt[rg][kc][0] = p[0];   // 4 contiguous hi pieces per row and instruction
            if (!HALF) a_nxt[rg][kc][1] = p[4];   // 4 contiguous lo pieces
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
        const int base = i * NTHR + wave * 64;   // wave-uniform: this wave's 64 consecutive 16-byte pieces
        // pieces come in 64-lane groups, hi (even group) then lo (odd group) per column tile
        if (base < NT * 128 && !(HALF && ((base >> 6) & 1)) && !NODMA) glds16(src + base + lane, dst + (unsigned)base * 16u);
      }
    }
  };

  // MODE bit 128 (ISF_CONV16_WIND=1, experiment) for the <= 64-column layers: WAVE-INDEPENDENT main loop.  Their
  // weights are tiny (<= 8 KiB per tap and chunk, L2 / L1 resident), so each wave reads its B fragments straight
  // from global memory into registers instead of sharing them through the LDS ring: no DMA in inline asm, hence no
  // hand-placed vmcnt(0), no barrier per step and no lock-step between the waves -- every load is visible to the
  // compiler, which emits counted waits for a software pipeline of (tap, chunk) items two deep; a wave walks ITS OWN
  // tap mask.  Same products, same summation order (tap ascending, chunk ascending) as the default.
  constexpr bool WIND = (MODE & 128) != 0 && NT <= 4 && KCH == NCH;
  if constexpr (WIND) {
    struct Item {
      uint4 a[RG][2];   // [row group][hi, lo]
      uint4 b[NT][2];   // [column tile][hi, lo]
    };
    // Every item issues the SAME number of loads in the same order -- weights first, gathers last, rows without a
    // neighbour read the all-zero line -- so the compiler's wait before an item's first MFMA is vmcnt(loads of one
    // item): exactly the next item's loads, gathers included, stay in flight.  (Conditional gathers would make it
    // assume none were issued and wait them out.)
    auto load_item = [&](Item& it, int tap, int kc) {
      const uint4* src = wpk + (((size_t)tap * NCH + kc) * ntiles_total + cb * NT) * 128 + lane;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        it.b[nt][0] = src[nt * 128];
        it.b[nt][1] = src[nt * 128 + 64];
      }
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) {
        int idx = -1;
        if ((rgm[rg] >> tap) & 1u) idx = nbr_l[tap * TM + wave * WR + rg * 16 + col];
        const uint4* row = xs + ((size_t)(idx >= 0 ? idx : 0) * CH8 + kc * 4) * 2 + kg;
        const uint4* hi = idx >= 0 ? row : g_zero_line + kg;
        const uint4* lo = idx >= 0 ? row + 4 : g_zero_line + kg;
        it.a[rg][0] = *hi;
        it.a[rg][1] = *lo;
      }
    };
    auto mma_item = [&](const Item& it, int tap) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const h8 bh = *reinterpret_cast<const h8*>(&it.b[nt][0]);
        const h8 bl = *reinterpret_cast<const h8*>(&it.b[nt][1]);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
          if ((rgm[rg] >> tap) & 1u) {
            const h8 ah = *reinterpret_cast<const h8*>(&it.a[rg][0]);
            const h8 al = *reinterpret_cast<const h8*>(&it.a[rg][1]);
            if (TEPI) {
              acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al, acc[rg][nt], 0, 0, 0);
              acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah, acc[rg][nt], 0, 0, 0);
              acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah, acc[rg][nt], 0, 0, 0);
            } else {
              acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[rg][nt], 0, 0, 0);
              acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[rg][nt], 0, 0, 0);
              acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[rg][nt], 0, 0, 0);
            }
          }
        }
      }
    };
    // item cursor: taps of this wave's mask ascending, chunks ascending inside a tap (all wave-uniform scalars)
    unsigned rem = wmask;
    int cur_tap = -1, cur_kc = NCH;
    auto next = [&](int& tap, int& kc) -> bool {
      if (cur_kc + 1 < NCH) {
        ++cur_kc;
      } else {
        if (rem == 0) return false;
        cur_tap = __ffs(rem) - 1;
        rem &= rem - 1;
        cur_kc = 0;
      }
      tap = cur_tap;
      kc = cur_kc;
      return true;
    };
    // Two slots.  The loads of the NEXT item are issued, unconditionally and in straight-line code, right before the
    // MFMAs of the current one (past the last item a slot re-reads item (tap 0, chunk 0): valid addresses, never
    // multiplied), so the compiler's wait before those MFMAs is exactly vmcnt(loads of one item) -- conditional or
    // loop-carried refills make its bookkeeping assume the worst and drain the prefetch.
    // (An item of a 32-column layer is 8 loads and 12 MFMAs: three slots there -- two items in flight per wave.)
    constexpr int WD = NT <= 2 ? 3 : 2;
    Item i0, i1;
    int t0 = 0, k0 = 0, t1 = 0, k1 = 0;
    bool v0 = !NOLOOP && next(t0, k0);
    if (!v0) t0 = k0 = 0;
    load_item(i0, t0, k0);
    if constexpr (WD == 2) {
      while (v0) {
        bool v1 = next(t1, k1);
        if (!v1) t1 = k1 = 0;
        load_item(i1, t1, k1);
        mma_item(i0, t0);
        if (!v1) break;
        v0 = next(t0, k0);
        if (!v0) t0 = k0 = 0;
        load_item(i0, t0, k0);
        mma_item(i1, t1);
      }
    } else {
      Item i2;
      int t2 = 0, k2 = 0;
      bool v1 = v0 && next(t1, k1);
      if (!v1) t1 = k1 = 0;
      load_item(i1, t1, k1);
      while (v0) {
        bool v2 = next(t2, k2);
        if (!v2) t2 = k2 = 0;
        load_item(i2, t2, k2);
        mma_item(i0, t0);
        if (!v1) break;
        v0 = next(t0, k0);
        if (!v0) t0 = k0 = 0;
        load_item(i0, t0, k0);
        mma_item(i1, t1);
        if (!v2) break;
        v1 = next(t1, k1);
        if (!v1) t1 = k1 = 0;
        load_item(i1, t1, k1);
        mma_item(i2, t2);
      }
    }
  } else if constexpr (TPS == 1) {
    Cursor cur{0u, -1, -1};
    if (nsteps > 0) {
      advance(cur);
      load_A(cur.tap, cur.ch);
      stage_B(cur.tap, cur.ch, 0);
    }
    for (int s = 0; s < nsteps; ++s) {
      const int tap = cur.tap;
      uint4 a_cur[RG][KCH][2];
  #pragma unroll
      for (int rg = 0; rg < RG; ++rg)
  #pragma unroll
        for (int kc = 0; kc < KCH; ++kc) { a_cur[rg][kc][0] = a_nxt[rg][kc][0]; a_cur[rg][kc][1] = a_nxt[rg][kc][1]; }
      // this wave's share of B(s) (and its A(s) rows) must have landed before anyone reads the buffer
      // (the builtin, not inline asm, so that hipcc's own scoreboard knows every tracked load has landed too and
      //  does not re-wait for the A(s) registers in the middle of the next prefetch; simm16 0x0F70 = vmcnt(0))
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();  // B(s) complete for every wave; everyone is done reading buffer (s+1)&1
      if (s + 1 < nsteps) {
        advance(cur);
        load_A(cur.tap, cur.ch);
        stage_B(cur.tap, cur.ch, (s + 1) & 1);
      }
      if ((wmask >> tap) & 1u) {
        if (PRIO) __builtin_amdgcn_s_setprio(2);
        const uint4* b = bbuf + (s & 1) * (KCH * NT * 128) + lane;
        bool need[RG];
  #pragma unroll
        for (int rg = 0; rg < RG; ++rg) need[rg] = (rgm[rg] >> tap) & 1u;   // scalar (wave-uniform)
        uint4 bhu_n = b[0], blu_n = make_uint4(0, 0, 0, 0);   // the next B fragments are read from LDS while these multiply
        if (!HALF) blu_n = b[64];
  #pragma unroll
        for (int i = 0; i < KCH * NT; ++i) {   // i = kc * NT + nt
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
              if (TEPI) {
                if (!HALF) {
                  acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al, acc[rg][nt], 0, 0, 0);
                  acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah, acc[rg][nt], 0, 0, 0);
                }
                acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah, acc[rg][nt], 0, 0, 0);
              } else {
                if (!HALF) {
                  acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[rg][nt], 0, 0, 0);
                  acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[rg][nt], 0, 0, 0);
                }
                acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[rg][nt], 0, 0, 0);
              }
            }
          }
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
      }
    }
  } else {
    // ---- multi-tap steps: a step covers the next TPS taps of wg_mask (increasing tap order, so every accumulator
    // sees its products in the same order as with one tap per step)
    unsigned rem = wg_mask;
    int nxt_t[TPS];
    uint4 a_n[TPS][RG][KCH][2];
    auto next_group = [&]() {
#pragma unroll
      for (int tp = 0; tp < TPS; ++tp) {
        nxt_t[tp] = -1;
        if (rem) {
          nxt_t[tp] = __ffs(rem) - 1;
          rem &= rem - 1;
        }
      }
    };
    auto load_group = [&](int buf) {
#pragma unroll
      for (int tp = 0; tp < TPS; ++tp) {
        const int tap = nxt_t[tp];
        if (tap < 0) continue;   // wave-uniform (wg_mask is)
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
          if ((rgm[rg] >> tap) & 1u) {
            const int idx = nbr_l[tap * TM + wave * WR + rg * 16 + col];
#pragma unroll
            for (int kc = 0; kc < KCH; ++kc) {
              a_n[tp][rg][kc][0] = make_uint4(0, 0, 0, 0);
              a_n[tp][rg][kc][1] = make_uint4(0, 0, 0, 0);
            }
            if (idx >= 0 && !NOGATHER) {
#pragma unroll
              for (int kc = 0; kc < KCH; ++kc) {
                const uint4* p = xs + ((size_t)idx * CH8 + kc * 4) * 2 + kg;
                a_n[tp][rg][kc][0] = p[0];
                if (!HALF) a_n[tp][rg][kc][1] = p[4];
              }
            }
          }
        }
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) {
          const uint4* src = wpk + (((size_t)tap * NCH + kc) * ntiles_total + cb * NT) * 128;
          const unsigned dst = bbuf_addr + (unsigned)(((buf * TPS + tp) * KCH + kc) * (NT * 128)) * 16u;
#pragma unroll
          for (int i = 0; i < (NT * 128 + NTHR - 1) / NTHR; ++i) {
            const int base = i * NTHR + wave * 64;
            if (base < NT * 128 && !(HALF && ((base >> 6) & 1)) && !NODMA)
              glds16(src + base + lane, dst + (unsigned)base * 16u);
          }
        }
      }
    };
    const int ngroups = NOLOOP ? 0 : (ntaps + TPS - 1) / TPS;
    if (ngroups > 0) {
      next_group();
      load_group(0);
    }
    for (int g = 0; g < ngroups; ++g) {
      int cur_t[TPS];
      uint4 a_c[TPS][RG][KCH][2];
#pragma unroll
      for (int tp = 0; tp < TPS; ++tp) {
        cur_t[tp] = nxt_t[tp];
#pragma unroll
        for (int rg = 0; rg < RG; ++rg)
#pragma unroll
          for (int kc = 0; kc < KCH; ++kc) {
            a_c[tp][rg][kc][0] = a_n[tp][rg][kc][0];
            a_c[tp][rg][kc][1] = a_n[tp][rg][kc][1];
          }
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();   // stage g complete for every wave; everyone is done reading stage (g+1)&1
      if (g + 1 < ngroups) {
        next_group();
        load_group((g + 1) & 1);
      }
      if (PRIO) __builtin_amdgcn_s_setprio(2);
#pragma unroll
      for (int tp = 0; tp < TPS; ++tp) {
        const int tap = cur_t[tp];
        if (tap < 0 || !((wmask >> tap) & 1u)) continue;   // wave-uniform
        const uint4* b = bbuf + ((g & 1) * TPS + tp) * (KCH * NT * 128) + lane;
#pragma unroll
        for (int i = 0; i < KCH * NT; ++i) {   // i = kc * NT + nt
          const int kc = i / NT, nt = i % NT;
          const uint4 bhu = b[(i * 2 + 0) * 64];
          uint4 blu = make_uint4(0, 0, 0, 0);
          if (!HALF) blu = b[(i * 2 + 1) * 64];
          const h8 bh = *reinterpret_cast<const h8*>(&bhu);
          const h8 bl = *reinterpret_cast<const h8*>(&blu);
#pragma unroll
          for (int rg = 0; rg < RG; ++rg) {
            if ((rgm[rg] >> tap) & 1u) {
              const h8 ah = *reinterpret_cast<const h8*>(&a_c[tp][rg][kc][0]);
              const h8 al = *reinterpret_cast<const h8*>(&a_c[tp][rg][kc][1]);
              if (TEPI) {
                if (!HALF) {
                  acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al, acc[rg][nt], 0, 0, 0);
                  acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah, acc[rg][nt], 0, 0, 0);
                }
                acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah, acc[rg][nt], 0, 0, 0);
              } else {
                if (!HALF) {
                  acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[rg][nt], 0, 0, 0);
                  acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[rg][nt], 0, 0, 0);
                }
                acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[rg][nt], 0, 0, 0);
              }
            }
          }
        }
      }
      if (PRIO) __builtin_amdgcn_s_setprio(0);
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();  // all waves done with the weight buffers -> reuse as the epilogue transpose tile

  // ---- epilogue: per row group, accumulator (col = lane&15, row = 4*(lane>>4)+t) -> LDS row-major ->
  //      one lane per (row, 8-channel unit): BN fold (incl. the weight scale), residual, ReLU, split, store
  constexpr int EPN = S::EPN;
  constexpr int RS = 16 * EPN + 4;
  float* tile_l = reinterpret_cast<float*>(smem) + wave * 16 * RS;
  const float winv = *w_inv_scale;
  if (TEPI) {
    // split format addressed in 8-byte halves of the 16-byte pieces: piece index p -> uint2 index 2p (+1 for the
    // upper 4 channels of the unit)
    const uint2* res2 = reinterpret_cast<const uint2*>(residual);
    uint2* ys2 = reinterpret_cast<uint2*>(ys);
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      const int grow = row0 + wave * WR + rg * 16 + col;
      if (grow >= n_out) continue;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int gc = cb * BN + nt * 16 + 4 * kg;                       // this lane's 4 channels
        const size_t piece = split_hi_index((size_t)grow, cout >> 3, gc >> 3);
        const size_t h2 = piece * 2 + ((gc >> 2) & 1), l2 = (piece + 4) * 2 + ((gc >> 2) & 1);
        // one 16-byte load each for the four scales / shifts (the scalar form compiles to a load-wait pair per channel)
        f32x4 sc4 = f32x4{1.f, 1.f, 1.f, 1.f}, sh4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (scale) sc4 = *reinterpret_cast<const f32x4*>(scale + gc);
        if (shift) sh4 = *reinterpret_cast<const f32x4*>(shift + gc);
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float sc = scale ? sc4[t] * winv : winv;
          v[t] = fmaf(acc[rg][nt][t], sc, sh4[t]);
        }
        if (residual) {
          const uint2 rh = res2[h2], rl = res2[l2];
          const _Float16* ph = reinterpret_cast<const _Float16*>(&rh);
          const _Float16* pl = reinterpret_cast<const _Float16*>(&rl);
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] += (float)ph[t] + (float)pl[t];
        }
        uint2 oh, ol;
        _Float16* qh = reinterpret_cast<_Float16*>(&oh);
        _Float16* ql = reinterpret_cast<_Float16*>(&ol);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float x = relu ? fmaxf(v[t], 0.f) : v[t];
          const _Float16 hi = (_Float16)x;
          qh[t] = hi;
          ql[t] = (_Float16)(x - (float)hi);
        }
        ys2[h2] = oh;
        ys2[l2] = ol;
      }
    }
    return;
  }
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) {
#pragma unroll
    for (int ps = 0; ps < NT / EPN; ++ps) {  // passes of EPN column tiles (keeps the transpose tile small)
      constexpr int UNITS = (16 * EPN) / 8;           // 8-channel units per row in this pass
      constexpr int ITEMS = (16 * UNITS + 63) / 64;   // (row, unit) items per lane
      // the residual rows of this pass are requested before the LDS transpose, so their latency hides behind it
      uint4 res_hi[ITEMS], res_lo[ITEMS];
      if (residual) {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
          const int i = lane + 64 * it;
          const int grow = row0 + wave * WR + rg * 16 + i / UNITS;
          res_hi[it] = make_uint4(0, 0, 0, 0);
          res_lo[it] = make_uint4(0, 0, 0, 0);
          if (i < 16 * UNITS && grow < n_out) {
            const size_t o = split_hi_index((size_t)grow, cout >> 3, (cb * BN + ps * (16 * EPN)) / 8 + i % UNITS);
            res_hi[it] = residual[o];
            res_lo[it] = residual[o + 4];
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
        const int grow = row0 + wave * WR + rg * 16 + r;
        if (i < 16 * UNITS && grow < n_out) {
          const float* tp = tile_l + r * RS + u * 8;
          f32x8 v;
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = tp[j];
          const int gc = cb * BN + ps * (16 * EPN) + u * 8;
          if (VEPI) {
            f32x8 sc8, sh8;
#pragma unroll
            for (int j = 0; j < 8; ++j) { sc8[j] = 1.f; sh8[j] = 0.f; }
            if (scale) sc8 = *reinterpret_cast<const f32x8*>(scale + gc);
            if (shift) sh8 = *reinterpret_cast<const f32x8*>(shift + gc);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], scale ? sc8[j] * winv : winv, sh8[j]);
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float sc = scale ? scale[gc + j] * winv : winv;
              const float sh = shift ? shift[gc + j] : 0.f;
              v[j] = fmaf(v[j], sc, sh);
            }
          }
          const size_t o = split_hi_index((size_t)grow, cout >> 3, gc >> 3);
          if (residual) v += join8(res_hi[it], res_lo[it]);
          if (relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
          }
          uint4 hi, lo;
          split8(v, hi, lo);
          ys[o] = hi;
          ys[o + 4] = lo;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
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
__global__ void pack_filters16_kernel(const float* __restrict__ w, int K, int cin, int cout,
                                      const unsigned* __restrict__ absmax_bits, uint4* __restrict__ packed,
                                      float* __restrict__ header) {
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
  for (int jj = 0; jj < 8; ++jj)
    v[jj] = w[((size_t)k * cin + 32 * ch + 8 * (lane >> 4) + jj) * cout + 16 * nt + (lane & 15)] * s;
  uint4 hi, lo;
  split8(v, hi, lo);
  const size_t base = (((size_t)k * nch + ch) * ntiles + nt) * 128;
  packed[base + lane] = hi;
  packed[base + 64 + lane] = lo;
}

// Tuning switches, read once: ISF_CONV16_WIDE=1 -> one 256-column workgroup for Cout = 256 (NT = 16);
// ISF_CONV16_NW / ISF_CONV16_RG: see launch16_rows.
static const bool g_conv16_wide = [] {
  const char* e = getenv("ISF_CONV16_WIDE");
  return e ? (e[0] != '0') : false;
}();
static const int g_conv16_nw = [] {
  const char* e = getenv("ISF_CONV16_NW");
  return e ? atoi(e) : 0;
}();
static const int g_conv16_diag = [] {   // timing diagnostics, see spconv_f16x3_kernel
  const char* e = getenv("ISF_CONV16_DIAG");
  return e ? atoi(e) : 0;
}();
static const bool g_conv16_vepi = [] {   // experiment: vector loads of the BN scale / shift in the epilogue
  const char* e = getenv("ISF_CONV16_VEPI");
  return e ? (e[0] != '0') : false;
}();
static const bool g_conv16_deep = [] {   // experiment: 8 waves x 16 rows for the 128-column layers of the small levels
  const char* e = getenv("ISF_CONV16_DEEP");
  return e ? (e[0] != '0') : false;
}();
static const bool g_conv16_wind = [] {   // experiment: wave-independent main loop for the <= 64-column layers
  const char* e = getenv("ISF_CONV16_WIND");
  return e ? (e[0] != '0') : false;
}();
static const bool g_conv16_tps = [] {   // experiment: several taps per step for the narrow layers (default shape only)
  const char* e = getenv("ISF_CONV16_TPS");
  return e ? (e[0] != '0') : false;
}();
static const bool g_conv16_tepi = [] {   // experiment: transposed accumulators, LDS-free epilogue (default shape only)
  const char* e = getenv("ISF_CONV16_TEPI");
  return e ? (e[0] != '0') : false;
}();
static const bool g_conv16_prio = [] {   // experiment: s_setprio around the MFMA block (default shape only)
  const char* e = getenv("ISF_CONV16_PRIO");
  return e ? (e[0] != '0') : false;
}();
static const int g_conv16_rg = [] {
  const char* e = getenv("ISF_CONV16_RG");
  return e ? atoi(e) : 0;
}();

bool sparse_conv_f16x3_supported(int c_in, int c_out) {
  return (c_in == 32 || c_in == 64 || c_in == 128 || c_in == 256) &&
         (c_out == 32 || c_out == 64 || c_out == 128 || c_out == 256);
}

template <int CIN, int NT, int RG, int NW, int MODE = 0>
static int launch16(const uint4* xs, const uint4* wpk, const float* winv, int K, int cout, const int32_t* nbr,
                    int nbr_stride, int n_out, const float* scale, const float* shift, const uint4* residual,
                    int relu, uint4* ys, hipStream_t st) {
  using S = Conv16Smem<NT, RG, Conv16Step<CIN, NT>::KCH * conv16_tps<CIN, NT, MODE>(), NW>;
  auto kern = spconv_f16x3_kernel<CIN, NT, RG, NW, MODE>;
  static bool attr_set = false;
  if (!attr_set && S::bytes > 48 * 1024) {
    ISF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, S::bytes));
    attr_set = true;
  }
  const int row_tiles = ceil_div(n_out, S::TM);
  const int ncb = cout / (16 * NT);
  ISF_REQUIRE(ncb == 1 || ncb == 2, ISF_ERR_UNSUPPORTED, "sparse_conv16: %d column blocks", ncb);
  const int blocks = 8 * (ncb == 2 ? ceil_div(row_tiles, 4) : ceil_div(row_tiles, 8));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * NW), S::bytes, st, xs, nbr, nbr_stride, wpk, winv, K, cout, scale,
                     shift, residual, ys, n_out, relu, row_tiles);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

// Workgroup shape per layer.  16 waves x 32 rows (512-row tiles, one workgroup per CU) share each weight stage
// 4x wider than 4 waves do.  Measured (B=4 x 300 k points): a win only for the 128-column layers of the large
// levels (128 -> 128: 1.13 -> 1.03 ms per 4 launches, 64 -> 128: 0.173 -> 0.147 ms); the narrow layers lose
// (fewer independent workgroups to hide the gather latency) and the small deep levels do not have enough tiles
// for 256 CUs.  ISF_CONV16_NW=4|8|16 and ISF_CONV16_RG=1|2|4 override (tuning; RG=1 applies to <= 64-column layers only).
template <int CIN, int NT>
static int launch16_rows(const uint4* xs, const uint4* wpk, const float* winv, int K, int cout, const int32_t* nbr,
                         int nbr_stride, int n_out, const float* scale, const float* shift, const uint4* residual,
                         int relu, uint4* ys, hipStream_t st) {
  const int ncb = cout / (16 * NT);
  // the narrow-layer experiments (TPS, WIND) leave the other layers on the production heuristic below
  const int mode = (g_conv_precision == 2 ? 1 : 0) | g_conv16_diag | (g_conv16_prio ? 16 : 0) | (g_conv16_tepi ? 32 : 0) |
                   (NT <= 4 ? (g_conv16_tps ? 64 : 0) | (g_conv16_wind ? 128 : 0) : 0) | (g_conv16_vepi ? 256 : 0);
  if (mode != 0) {   // single-pass f16 (opt-in) and the timing diagnostics: the default workgroup shape only
#define ISF_MODE16(M)                                                                                                 \
  case M:                                                                                                             \
    return launch16<CIN, NT, 2, 4, M>(xs, wpk, winv, K, cout, nbr, nbr_stride, n_out, scale, shift, residual, relu, ys, \
                                      st)
    switch (mode) {
      ISF_MODE16(1);
      ISF_MODE16(2);
      ISF_MODE16(4);
      ISF_MODE16(6);
      ISF_MODE16(8);
      ISF_MODE16(16);
      ISF_MODE16(32);
      ISF_MODE16(48);
      ISF_MODE16(64);
      ISF_MODE16(96);
      ISF_MODE16(128);
      ISF_MODE16(160);
      ISF_MODE16(256);
      ISF_MODE16(384);
      default:
        ISF_REQUIRE(false, ISF_ERR_ARG, "sparse_conv16: mode %d (precision 2, ISF_CONV16_DIAG in {2,4,6,8} and ISF_CONV16_PRIO are not combinable)", mode);
    }
#undef ISF_MODE16
  }
  bool wide_wg = NT == 8 && CIN >= 64 && (long long)ceil_div(n_out, 512) * ncb >= 200;
  if (g_conv16_nw == 4) wide_wg = false;
  if (g_conv16_nw == 16) wide_wg = NT <= 8;
  if (g_conv16_nw == 8 && NT == 8)   // experiment: 8-wave (256-row) workgroups, two per CU, for the 128-column layers
    return launch16<CIN, (NT == 8 ? NT : 2), 2, 8>(xs, wpk, winv, K, cout, nbr, nbr_stride, n_out, scale, shift, residual,
                                                  relu, ys, st);
  if (wide_wg)
    return launch16<CIN, (NT <= 8 ? NT : 2), 2, 16>(xs, wpk, winv, K, cout, nbr, nbr_stride, n_out, scale, shift,
                                                     residual, relu, ys, st);
  if (g_conv16_deep && NT == 8)   // experiment: the same 128-row tile on 8 waves x 16 rows (twice the waves per SIMD)
    return launch16<CIN, (NT == 8 ? NT : 2), 1, 8>(xs, wpk, winv, K, cout, nbr, nbr_stride, n_out, scale, shift, residual,
                                                  relu, ys, st);
  if (g_conv16_rg == 1 && NT <= 4)   // experiment: 64-row workgroups for the narrow layers (twice the waves in flight)
    return launch16<CIN, (NT <= 4 ? NT : 2), 1, 4>(xs, wpk, winv, K, cout, nbr, nbr_stride, n_out, scale, shift, residual,
                                                  relu, ys, st);
  if (g_conv16_rg == 4 && NT * 4 <= 32)
    return launch16<CIN, (NT * 4 <= 32 ? NT : 2), 4, 4>(xs, wpk, winv, K, cout, nbr, nbr_stride, n_out, scale, shift,
                                                        residual, relu, ys, st);
  return launch16<CIN, NT, 2, 4>(xs, wpk, winv, K, cout, nbr, nbr_stride, n_out, scale, shift, residual, relu, ys, st);
}

template <int CIN>
static int dispatch16(const uint4* xs, const uint4* wpk, const float* winv, int K, int cout, const int32_t* nbr,
                      int nbr_stride, int n_out, const float* scale, const float* shift, const uint4* residual,
                      int relu, uint4* ys, hipStream_t st) {
  switch (cout) {
    case 32:  return launch16_rows<CIN, 2>(xs, wpk, winv, K, cout, nbr, nbr_stride, n_out, scale, shift, residual, relu, ys, st);
    case 64:  return launch16_rows<CIN, 4>(xs, wpk, winv, K, cout, nbr, nbr_stride, n_out, scale, shift, residual, relu, ys, st);
    case 128: return launch16_rows<CIN, 8>(xs, wpk, winv, K, cout, nbr, nbr_stride, n_out, scale, shift, residual, relu, ys, st);
    case 256:
      if (g_conv16_wide && g_conv_precision != 2 && g_conv16_diag == 0 && !g_conv16_prio && !g_conv16_tepi &&
          !g_conv16_tps && !g_conv16_wind && !g_conv16_vepi)
        return launch16<CIN, 16, 2, 4>(xs, wpk, winv, K, cout, nbr, nbr_stride, n_out, scale, shift, residual, relu, ys, st);
      return launch16_rows<CIN, 8>(xs, wpk, winv, K, cout, nbr, nbr_stride, n_out, scale, shift, residual, relu, ys, st);
  }
  return ISF_ERR_UNSUPPORTED;
}

// packed16 = K*Cin*Cout*4 bytes of fragments followed by a 64-byte header
int sparse_conv_forward_f16x3_impl(const void* xs, int c_in, const void* packed16, int K, int c_out,
                                   const int32_t* nbr, int nbr_stride, int n_out, const float* scale,
                                   const float* shift, const void* residual, int relu, void* ys,
                                   hipStream_t st) {
  if (n_out <= 0) return ISF_OK;
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
    case 32:  return dispatch16<32>(x, w, winv, K, c_out, nbr, nbr_stride, n_out, scale, shift, r, relu, y, st);
    case 64:  return dispatch16<64>(x, w, winv, K, c_out, nbr, nbr_stride, n_out, scale, shift, r, relu, y, st);
    case 128: return dispatch16<128>(x, w, winv, K, c_out, nbr, nbr_stride, n_out, scale, shift, r, relu, y, st);
    case 256: return dispatch16<256>(x, w, winv, K, c_out, nbr, nbr_stride, n_out, scale, shift, r, relu, y, st);
  }
  return ISF_ERR_UNSUPPORTED;
}

int pack_filters16_impl(Arena& a, const float* w, int K, int cin, int cout, void* packed16, hipStream_t st) {
  unsigned* amax = nullptr;
  ISF_TRY(a.alloc_n(&amax, 64));
  ISF_HIP_TRY(hipMemsetAsync(amax, 0, sizeof(unsigned), st));
  const size_t n = (size_t)K * cin * cout;
  hipLaunchKernelGGL(absmax_kernel, dim3(ceil_div((long long)n, 1024) < 1024 ? ceil_div((long long)n, 1024) : 1024),
                     dim3(256), 0, st, w, n, amax);
  const long long total = (long long)K * (cin >> 5) * (cout >> 4) * 64;
  hipLaunchKernelGGL(pack_filters16_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, w, K, cin, cout, amax,
                     reinterpret_cast<uint4*>(packed16),
                     reinterpret_cast<float*>(reinterpret_cast<char*>(packed16) + n * 4));
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
  isf::Arena& a = isf::arena_for_current_device();
  ISF_TRY(a.reset());
  return isf::pack_filters16_impl(a, filters, num_taps, c_in, c_out, packed16, isf::as_stream(stream));
}

int isf_f32_to_split(const float* x, size_t num_elems, void* xs, isf_stream_t stream) {
  ISF_REQUIRE(num_elems == 0 || (x && xs), ISF_ERR_ARG, "f32_to_split: null pointer");
  return isf::f32_to_split_impl(x, num_elems, xs, isf::as_stream(stream));
}

int isf_split_to_f32(const void* xs, size_t num_elems, float* x, isf_stream_t stream) {
  ISF_REQUIRE(num_elems == 0 || (x && xs), ISF_ERR_ARG, "split_to_f32: null pointer");
  return isf::split_to_f32_impl(xs, num_elems, x, isf::as_stream(stream));
}

int isf_sparse_conv_forward_f16x3(const void* features_split, int num_in, int c_in, const void* packed16,
                                  int num_taps, int c_out, const int32_t* nbr, int nbr_stride, int num_out,
                                  const float* scale, const float* shift, const void* residual_split, int relu,
                                  void* out_split, isf_stream_t stream) {
  ISF_REQUIRE(num_in >= 0 && num_out >= 0 && c_in > 0 && c_out > 0 && num_taps > 0, ISF_ERR_ARG,
              "sparse_conv_forward_f16x3: bad arguments");
  if (num_out == 0) return ISF_OK;
  ISF_REQUIRE(features_split && packed16 && nbr && out_split && ((scale == nullptr) == (shift == nullptr)),
              ISF_ERR_ARG, "sparse_conv_forward_f16x3: null pointer");
  return isf::sparse_conv_forward_f16x3_impl(features_split, c_in, packed16, num_taps, c_out, nbr, nbr_stride,
                                             num_out, scale, shift, residual_split, relu, out_split,
                                             isf::as_stream(stream));
}

}  // extern "C"
