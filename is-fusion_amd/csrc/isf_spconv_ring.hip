// isf_spconv_ring.hip -- f16x3 sparse convolution, multi-stage ring kernel (round 2).
//
// Same arithmetic, same tiling and the SAME summation order as spconv_f16x3_kernel (isf_spconv16.hip): stages are
// (32-channel chunk group, tap) in chunk-outer / tap-ascending order, products a_lo*b_hi + a_hi*b_lo + a_hi*b_hi on
// v_mfma_f32_16x16x32_f16 -- results are bit-identical, which is how this kernel is validated.  What changes is how the
// operands reach the matrix pipe.  Measured on the one-step-prefetch kernel (profiles/r02_call1_knockout_variants.txt):
// its loop without any global traffic already runs close to what the tile count allows, but gathers + weight DMA ADD
// 60-120 us per launch on top instead of hiding (every step ends in `s_waitcnt vmcnt(0)` + `__syncthreads()`, so a load
// has one step -- about one L2 round trip under load -- to land), and prologue + epilogue are 25 % of the conv time.
//
//  * every global load of the main loop is invisible to the compiler: weights and neighbour indices go global -> LDS
//    by LDS-DMA, gathered activation fragments global -> VGPR by `global_load_dwordx4` in inline asm.  hipcc therefore
//    inserts no vmcnt waits of its own in the loop; the kernel waits with ONE counted `s_waitcnt vmcnt(N)` per step,
//    N = the operations issued for later stages.  Every wave issues the same operations in the same order every step
//    (rows without a neighbour read an all-zero line; past the last stage the cursors wrap to valid stages whose data
//    is never used), which is what makes N a compile-time constant:
//        per step:   [ B(s+PA): nB LDS-DMA ] [ A(s+PA): nA hidden loads ] [ I(s+2PA+1): 1 LDS-DMA of 4 B / lane ]
//        wait at s:  N = 1 + (PA-1) * (nB + nA + 1)      (A(s) and everything older -- B(s), idx(s+PA) -- has landed)
//    PA = prefetch distance in stages (2 or 3; 1 reproduces the old kernel's timing), weight ring of PA+1 stages,
//    PA+1 statically named register sets (the step loop is unrolled by PA+1), index ring of PA+2 slots;
//  * bare `s_barrier` (LDS-DMA stays in flight across it; `__syncthreads()` is a fence that drains vmcnt);
//  * no neighbour tile in LDS and no ballot loop in the prologue: the per-16-row-group tap masks come precomputed with
//    the rulebook (rb_group_masks_kernel), the indices of a stage arrive through the 4-byte LDS-DMA ring.  LDS per
//    workgroup = weight ring + 1 KiB of indices per wave: 52 KiB for the 128-column layers at PA = 2 (3 workgroups/CU);
//  * workgroup shape (NW waves x RG row groups) chosen per launch from the tile count (sparse_conv_forward_ring_impl).
//
// The hidden loads are safe only if the compiler never copies, spills or reuses a destination register between the
// load and the `s_waitcnt` that covers it; tools/check_hidden_loads.py proves that on the generated ISA of every
// instantiation (run by __graft_entry__.build()).
#include "isf_spconv16.h"

#include <type_traits>

namespace isf {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// one all-zero 64-byte line: the gather address of rows that have no neighbour through a tap
__device__ uint4 g_ring_zero_line[4];

// destination tagged for tools/check_hidden_loads.py
__device__ __forceinline__ void hload16(u32x4& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off ; HIDDEN_LOAD" : "=v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void hwait() {
  asm volatile("s_waitcnt vmcnt(%0) ; HIDDEN_WAIT" ::"n"(N) : "memory");
}
// after the wait: the register now holds the loaded value (an empty asm that redefines it, ordered behind the wait)
__device__ __forceinline__ void hlanded(u32x4& v) { asm volatile("; HIDDEN_LANDED %0" : "+v"(v)); }

template <int CIN, int NT, int RG, int NW, int PA, bool HALF>
struct RingCfg {
  static constexpr int KCH = Conv16Step<CIN, NT>::KCH;       // 32-channel chunks per stage
  static constexpr int D = PA + 1;                           // weight ring stages = register sets
  static constexpr int NI = PA + 2;                          // index ring slots
  static constexpr int TM = 16 * RG * NW;                    // rows per workgroup
  static constexpr int stage_bytes = KCH * NT * 2048;
  static constexpr int bring_bytes = D * stage_bytes;
  static constexpr int iring_bytes = NI * NW * 256;          // 64 lanes x 4 B per wave and slot
  static constexpr int epi_bytes = NW * Conv16Epi<NT, RG>::wave_bytes;
  static constexpr int main_bytes = bring_bytes + iring_bytes;
  static constexpr int bytes = main_bytes > epi_bytes ? main_bytes : epi_bytes;
  static constexpr int nB = (KCH * NT * 2) / NW;             // LDS-DMA instructions (1 KiB each) per wave and stage
  static constexpr int nA = RG * KCH * (HALF ? 1 : 2);       // hidden 16-byte gathers per lane and stage
  static constexpr int T = nB + nA + 1;
  static constexpr int NWAIT = 1 + (PA - 1) * T;
  static_assert((KCH * NT * 2) % NW == 0, "the weight stage must split evenly over the waves");
  static_assert(NWAIT < 64, "vmcnt is a 6-bit counter");
  static_assert(16 * RG <= 64, "a wave's rows must fit one 4-byte LDS-DMA");
};

template <int CIN, int NT, int RG, int NW, int PA, bool HALF, int MINW>
__global__ __launch_bounds__(64 * NW, MINW) void spconv_ring_kernel(
    const uint4* __restrict__ xs, const int32_t* __restrict__ nbr, int nbr_stride, const uint32_t* __restrict__ gmask,
    const uint4* __restrict__ wpk, const float* __restrict__ w_inv_scale, int cout, const float* __restrict__ scale,
    const float* __restrict__ shift, const uint4* __restrict__ residual, uint4* __restrict__ ys, int n_out, int relu,
    int row_tiles) {
  using C = RingCfg<CIN, NT, RG, NW, PA, HALF>;
  constexpr int KCH = C::KCH, D = C::D, NI = C::NI, TM = C::TM;
  constexpr int WR = 16 * RG;         // rows per wave
  constexpr int NCH = CIN / 32;       // 32-channel chunks
  constexpr int NCG = NCH / KCH;      // chunk groups (stages per tap)
  constexpr int CH8 = CIN / 8;        // 8-channel (32-byte) units per input row
  constexpr int BN = 16 * NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const uint4* bring = reinterpret_cast<const uint4*>(smem);                       // [D][KCH][NT][hi|lo][64]
  const int* iring = reinterpret_cast<const int*>(smem + C::bring_bytes);          // [NI][NW][64]
  const unsigned bring_addr = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  const unsigned iring_addr = bring_addr + C::bring_bytes;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kg = lane >> 4;
  const int ncb = cout / BN;
  int cb, tile;
  if (!conv16_tile_of_block(ncb, row_tiles, cb, tile)) return;
  const int row0 = tile * TM;
  const int roww = row0 + wave * WR;
  const int ntiles_total = cout >> 4;

  // ---- tap masks of this workgroup's row groups (bit k: some row of the group has a neighbour through tap k)
  unsigned gm = 0;
  {
    const int g = (row0 >> 4) + lane;
    if (lane < NW * RG && g < (nbr_stride >> 4)) gm = gmask[g];
  }
  unsigned wg_mask = gm;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) wg_mask |= (unsigned)__shfl_xor((int)wg_mask, d, 64);
  wg_mask = __builtin_amdgcn_readfirstlane(wg_mask);
  unsigned rgm[RG];
  unsigned wmask = 0;
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) {
    rgm[rg] = (unsigned)__builtin_amdgcn_readlane((int)gm, wave * RG + rg);
    wmask |= rgm[rg];
  }
  const int nsteps = __popc(wg_mask) * NCG;

  f32x4 acc[RG][NT];
#pragma unroll
  for (int rg = 0; rg < RG; ++rg)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[rg][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // stage j -> (chunk group, tap): chunk-outer, taps = set bits of wg_mask ascending; past the end the cursor wraps
  // to the first stage again (valid addresses, data never used)
  struct Cursor {
    unsigned rem;   // taps of the current chunk group not yet visited
    int tap, cg;
  };
  auto advance = [&](Cursor& c) {
    if (c.rem == 0) {
      c.rem = wg_mask;
      c.cg = (c.cg + 1 >= NCG) ? 0 : c.cg + 1;
    }
    c.tap = __ffs(c.rem) - 1;
    c.rem &= c.rem - 1;
  };

  // I: the wave's WR neighbour indices of a tap -> its 256-byte slice of index slot `slot` (lanes >= WR re-read the last row)
  const int irow = min(roww + min(lane, WR - 1), nbr_stride - 1);
  auto issue_I = [&](const Cursor& c, int slot) {
    glds4(nbr + (size_t)c.tap * nbr_stride + irow, iring_addr + (unsigned)((slot * NW + wave) * 256));
  };
  // B: this wave's share of the stage's KCH * NT * 2 KiB of weight fragments -> ring slot `slot`
  auto issue_B = [&](const Cursor& c, int slot) {
#pragma unroll
    for (int t = 0; t < C::nB; ++t) {
      const int q = wave + NW * t;                 // 1-KiB piece of the stage (wave-uniform)
      const int kc = q / (NT * 2), jj = q - kc * (NT * 2);
      const uint4* src = wpk + (((size_t)c.tap * NCH + c.cg * KCH + kc) * ntiles_total + cb * NT) * 128 + jj * 64 + lane;
      glds16(src, bring_addr + (unsigned)(slot * C::stage_bytes + q * 1024));
    }
  };
  // A: gathered rows of the stage, straight into MFMA A-fragment registers (hidden loads)
  auto issue_A = [&](const Cursor& c, int islot, u32x4 (&a)[RG][KCH][2]) {
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      int idx = iring[(islot * NW + wave) * 64 + rg * 16 + col];
      if (roww + rg * 16 + col >= nbr_stride) idx = -1;   // rows beyond the table (the tile overhangs it)
#pragma unroll
      for (int kc = 0; kc < KCH; ++kc) {
        const uint4* row = xs + ((size_t)(idx >= 0 ? idx : 0) * CH8 + (c.cg * KCH + kc) * 4) * 2 + kg;
        const uint4* hi = idx >= 0 ? row : g_ring_zero_line + kg;
        hload16(a[rg][kc][0], hi);                         // 4 contiguous hi pieces per row and instruction
        if (!HALF) hload16(a[rg][kc][1], idx >= 0 ? row + 4 : hi);
      }
    }
  };
  auto landed = [&](u32x4 (&a)[RG][KCH][2]) {
#pragma unroll
    for (int rg = 0; rg < RG; ++rg)
#pragma unroll
      for (int kc = 0; kc < KCH; ++kc) {
        hlanded(a[rg][kc][0]);
        if (!HALF) hlanded(a[rg][kc][1]);
      }
  };
  auto multiply = [&](int tap, int slot, const u32x4 (&a)[RG][KCH][2]) {
    if (!((wmask >> tap) & 1u)) return;
    const uint4* b = bring + slot * (C::stage_bytes / 16) + lane;
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
          const h8 ah = __builtin_bit_cast(h8, a[rg][kc][0]);
          if (!HALF) {
            const h8 al = __builtin_bit_cast(h8, a[rg][kc][1]);
            acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[rg][nt], 0, 0, 0);
            acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[rg][nt], 0, 0, 0);
          }
          acc[rg][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[rg][nt], 0, 0, 0);
        }
      }
    }
  };

  if (nsteps > 0) {
    u32x4 aset[D][RG][KCH][2];
    Cursor ci{0u, -1, -1}, ca{0u, -1, -1}, cm{0u, -1, -1};   // index stream, operand stream, multiply stream
    int islot_i = 0;                                         // next index slot to fill
    // ---- warm-up: indices of stages 0..PA, then the pseudo-steps -PA..-1 in the steady-state order [B, A, I]
#pragma unroll
    for (int j = 0; j <= PA; ++j) {
      advance(ci);
      issue_I(ci, islot_i);
      islot_i = islot_i + 1 == NI ? 0 : islot_i + 1;
    }
    hwait<0>();
#pragma unroll
    for (int j = 0; j < PA; ++j) {
      advance(ca);
      issue_B(ca, j);
      issue_A(ca, j, aset[j]);
      advance(ci);
      issue_I(ci, islot_i);
      islot_i = islot_i + 1 == NI ? 0 : islot_i + 1;
    }
    int islot_a = PA;                                        // index slot of the next stage to gather (stage s + PA)
    // ---- steady state: stage s multiplies out of register set / ring slot s % D.  (One loop with an exit behind
    // every step: with a `break` inside a nested unrolled loop the exits are routed through the outer loop's latch, a
    // path on which tools/check_hidden_loads.py cannot see that the loop is left.)
    // One step (a macro, not a lambda: the register sets must stay scalarised).  The loop runs whole rounds of D
    // steps -- ONE exit, at the end of a round: with an exit behind every step hipcc's loop-exit unification routes the
    // exits through shared in-loop blocks with run-time flags (more registers, and paths tools/check_hidden_loads.py
    // cannot tell from staying in the loop).  The up to D-1 padding steps of the last round fetch wrapped stages and
    // multiply nothing.
#define ISF_RING_STEP(J)                                                                                             \
  if constexpr (J < D) {                                                                                             \
    advance(cm);                                                                                                     \
    hwait<C::NWAIT>();           /* A(s), B(s) (this wave's share), idx(s + PA) have landed */                       \
    landed(aset[J]);                                                                                                 \
    __builtin_amdgcn_s_barrier(); /* B(s) complete for every wave; slot (s-1) % D is free */                         \
    asm volatile("" ::: "memory");                                                                                   \
    advance(ca);                                                                                                     \
    issue_B(ca, (J + PA) % D);                                                                                       \
    issue_A(ca, islot_a, aset[(J + PA) % D]);                                                                        \
    islot_a = islot_a + 1 == NI ? 0 : islot_a + 1;                                                                   \
    advance(ci);                                                                                                     \
    issue_I(ci, islot_i);                                                                                            \
    islot_i = islot_i + 1 == NI ? 0 : islot_i + 1;                                                                   \
    if (s0 + J < nsteps) multiply(cm.tap, J, aset[J]);   /* wave-uniform */                                          \
  }
    for (int s0 = 0; s0 < nsteps; s0 += D) {
      ISF_RING_STEP(0)
      ISF_RING_STEP(1)
      ISF_RING_STEP(2)
      ISF_RING_STEP(3)
    }
#undef ISF_RING_STEP
    hwait<0>();            // the wrapped tail stages still in flight must land before the ring is reused
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  float* tile_l = reinterpret_cast<float*>(smem) + wave * (Conv16Epi<NT, RG>::wave_bytes / 4);
  conv16_epilogue<NT, RG>(acc, tile_l, lane, roww, cb * BN, cout, *w_inv_scale, scale, shift, residual, ys, n_out, relu);
}

// ------------------------------------------------------------------------------------------------ launch
bool sparse_conv_ring_supported(int c_in, int c_out) {
  return (c_in == 32 || c_in == 64 || c_in == 128 || c_in == 256) && (c_out == 32 || c_out == 64 || c_out == 128 || c_out == 256);
}

// tuning override (isf_tune_conv_ring; tools/conv_sweep.py): 0 = heuristic
extern int g_conv_ring;   // isf_spconv16.hip
int g_ring_nw = 0, g_ring_rg = 0, g_ring_pa = 0;

struct RingArgs {
  const uint4* xs;
  const int32_t* nbr;
  int nbr_stride;
  const uint32_t* gmask;
  const uint4* wpk;
  const float* winv;
  int cout;
  const float *scale, *shift;
  const uint4* residual;
  uint4* ys;
  int n_out, relu;
  hipStream_t st;
};

template <int CIN, int NT, int RG, int NW, int PA, bool HALF>
static int launch_ring(const RingArgs& a) {
  using C = RingCfg<CIN, NT, RG, NW, PA, HALF>;
  // waves per SIMD to plan registers for: what the LDS admits (160 KiB per CU), capped by a register estimate
  // (accumulators + PA+1 operand sets + ~56 for weight fragments and addressing; 512 registers per SIMD lane)
  constexpr int lds_wgs = (160 * 1024) / C::bytes < 1 ? 1 : (160 * 1024) / C::bytes;
  constexpr int lds_waves = (NW * lds_wgs + 3) / 4 > 8 ? 8 : (NW * lds_wgs + 3) / 4;
  constexpr int est = RG * NT * 4 + C::D * C::nA * 4 + 56;
  constexpr int reg_waves = est <= 96 ? 5 : est <= 128 ? 4 : est <= 168 ? 3 : est <= 256 ? 2 : 1;
  constexpr int need_waves = (NW + 3) / 4;   // one workgroup must fit
  constexpr int MINW0 = lds_waves < reg_waves ? lds_waves : reg_waves;
  constexpr int MINW = MINW0 < need_waves ? need_waves : MINW0;
  auto kern = spconv_ring_kernel<CIN, NT, RG, NW, PA, HALF, MINW>;
  static bool attr_set = false;
  if (!attr_set && C::bytes > 48 * 1024) {
    ISF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::bytes));
    attr_set = true;
  }
  const int row_tiles = ceil_div(a.n_out, C::TM);
  const int ncb = a.cout / (16 * NT);
  ISF_REQUIRE(ncb == 1 || ncb == 2, ISF_ERR_UNSUPPORTED, "sparse_conv_ring: %d column blocks", ncb);
  hipLaunchKernelGGL(kern, dim3(conv16_grid_blocks(ncb, row_tiles)), dim3(64 * NW), C::bytes, a.st, a.xs, a.nbr, a.nbr_stride,
                     a.gmask, a.wpk, a.winv, a.cout, a.scale, a.shift, a.residual, a.ys, a.n_out, a.relu, row_tiles);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

// shapes built per (CIN, NT): (NW, RG, PA).  The weight stage must split evenly over the waves (RingCfg::nB).
#define ISF_RING_TRY(NW_, RG_, PA_)                                                        \
  if (nw == NW_ && rg == RG_ && pa == PA_) {                                               \
    if constexpr ((Conv16Step<CIN, NT>::KCH * NT * 2) % NW_ == 0)                          \
      return half ? launch_ring<CIN, NT, RG_, NW_, PA_, true>(a) : launch_ring<CIN, NT, RG_, NW_, PA_, false>(a); \
  }

template <int CIN, int NT>
static int dispatch_ring(const RingArgs& a, int half) {
  // heuristic shape (see DESIGN.md section 5 for the sweep it comes from); the tuning override wins
  int nw = 4, rg = 2, pa = 2;
  if (NT == 8 && a.cout == 128 && a.n_out >= 8 * 256) nw = 8;
  if (g_ring_nw) nw = g_ring_nw;
  if (g_ring_rg) rg = g_ring_rg;
  if (g_ring_pa) pa = g_ring_pa;
  ISF_RING_TRY(4, 2, 1) ISF_RING_TRY(4, 2, 2) ISF_RING_TRY(4, 2, 3)
  ISF_RING_TRY(8, 2, 1) ISF_RING_TRY(8, 2, 2) ISF_RING_TRY(8, 2, 3)
  ISF_RING_TRY(4, 3, 2) ISF_RING_TRY(8, 3, 2)
  ISF_REQUIRE(false, ISF_ERR_UNSUPPORTED, "sparse_conv_ring: shape (NW %d, RG %d, PA %d) not built for Cin %d, %d columns",
              nw, rg, pa, CIN, 16 * NT);
}
#undef ISF_RING_TRY

template <int CIN>
static int dispatch_ring_cout(const RingArgs& a, int half) {
  switch (a.cout) {
    case 32:  return dispatch_ring<CIN, 2>(a, half);
    case 64:  return dispatch_ring<CIN, 4>(a, half);
    case 128:
    case 256: return dispatch_ring<CIN, 8>(a, half);
  }
  return ISF_ERR_UNSUPPORTED;
}

int sparse_conv_forward_ring_impl(const void* xs, int c_in, const void* packed16, int K, int c_out, const int32_t* nbr,
                                  int nbr_stride, int n_out, const uint32_t* gmask, const float* scale,
                                  const float* shift, const void* residual, int relu, void* ys, int half,
                                  hipStream_t st) {
  if (n_out <= 0) return ISF_OK;
  ISF_REQUIRE(K >= 1 && K <= kMaxTaps, ISF_ERR_UNSUPPORTED, "sparse_conv_ring: %d taps (max 27)", K);
  ISF_REQUIRE(sparse_conv_ring_supported(c_in, c_out), ISF_ERR_UNSUPPORTED, "sparse_conv_ring: (Cin,Cout)=(%d,%d) not built",
              c_in, c_out);
  ISF_REQUIRE(nbr_stride % 128 == 0 && nbr_stride >= n_out && gmask, ISF_ERR_ARG, "sparse_conv_ring: bad rulebook");
  RingArgs a{reinterpret_cast<const uint4*>(xs), nbr, nbr_stride, gmask, reinterpret_cast<const uint4*>(packed16),
             reinterpret_cast<const float*>(reinterpret_cast<const char*>(packed16) + (size_t)K * c_in * c_out * 4),
             c_out, scale, shift, reinterpret_cast<const uint4*>(residual), reinterpret_cast<uint4*>(ys), n_out, relu, st};
  switch (c_in) {
    case 32:  return dispatch_ring_cout<32>(a, half);
    case 64:  return dispatch_ring_cout<64>(a, half);
    case 128: return dispatch_ring_cout<128>(a, half);
    case 256: return dispatch_ring_cout<256>(a, half);
  }
  return ISF_ERR_UNSUPPORTED;
}

}  // namespace isf

extern "C" {

int isf_tune_conv_ring(int enable, int num_waves, int row_groups, int prefetch) {
  isf::g_conv_ring = enable ? 1 : 0;
  isf::g_ring_nw = num_waves;
  isf::g_ring_rg = row_groups;
  isf::g_ring_pa = prefetch;
  return ISF_OK;
}

}  // extern "C"
