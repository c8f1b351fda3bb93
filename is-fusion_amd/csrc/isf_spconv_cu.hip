// isf_spconv_cu.hip -- the f16x3 sparse convolution of the 256-column layers (levels 3 / 4 of the encoder) as ONE
// WORKGROUP PER COMPUTE UNIT over work units of equal matrix work.
//
// Why a different structure (profiles/r03_conv_trace.txt).  The tile kernel (isf_spconv16.hip) cuts
// a level-3 launch -- 40 k rows x 256 columns, 160 rows per CU -- into 128-row x 128-column tiles, 2.5 per CU, whose
// matrix work varies 3x with the density of the scene; the launch ends with its busiest CU (1.38x the mean) while every
// workgroup streams its own 16-KiB weight stage per step through the CU's vector-memory path and the LDS.  Finer tiles
// lose more to the weight stream than they gain in balance (measured: half tiles, equal-work tiles, tap split).  Here:
//   * a launch is cut into UNITS of whole 16-row groups with equal work (taps with a neighbour, summed over the unit's
//     groups -- conv_cu_plan_impl, once per rulebook on the geometry stream), one unit per CU, <= 256 rows each: the CUs
//     finish together;
//   * ONE workgroup of 8 waves owns the unit and ALL 256 output columns: wave w owns columns [32 w, 32 w + 32) of
//     every row group (16 x 2 accumulator tiles = 128 registers).  A weight fragment is needed by exactly one wave, so
//     the weights go global -> VGPR directly (1 KiB coalesced per instruction, one step ahead) and never touch LDS --
//     per step the CU pulls one 32-KiB stage for ALL its rows instead of 16 KiB per 128-row tile;
//   * the gathered rows come in by LDS-DMA, a lane quad fetching the 64 contiguous bytes of ONE row (the address-unit
//     friendly pattern of isf_spconv_dma.hip), each row ONCE per CU (the tile kernel gathers it once per column block),
//     into a three-stage ring: the gathers of step s + 2 are issued at step s, so an L2 miss has two steps to land;
//   * one s_barrier per step hands the stage over; row groups without the tap are skipped by every wave (scalar branch).
// Products and their order per accumulator are those of spconv_f16x3_kernel (chunk outer, taps inner, a_lo b_hi ->
// a_hi b_lo -> a_hi b_hi; a 16-row group multiplies through a tap iff one of its rows has a neighbour there): results
// are BIT-IDENTICAL to it.  Replaces, like it, the reference's gather -> GEMM -> scatter-add loop of
// bevfusion-ops/spconv/include/spconv/spconv_ops.h:260-361.
#include "isf_spconv16.h"
#include "isf_spconv_cu_mult.h"

#include <atomic>

namespace isf {

__device__ uint4 g_zero_line_cu[8];   // 128 zero bytes: what a row without a neighbour reads (one per translation unit: no RDC)

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));


static constexpr int kCuCout = 256;
static constexpr int kCuDepth = 1;                      // production prefetch depth (steps)
static constexpr int kCuWavesProd = 4;                  // production workgroup: 4 waves x 64 columns

template <int D, int NW, int CAP = kCuCapGroups>   // D = prefetch depth in steps; D + 1 stages of gathered rows; CAP groups per unit
struct ConvCuSmem {
  static constexpr int nbr_bytes = kMaxTaps * 16 * CAP * 4;           // [27][16 CAP] int32
  static constexpr int stage_bytes = CAP * 2048;                      // [CAP groups][hi, lo][64 x 16 B]
  static constexpr int ring_bytes = (D + 1) * stage_bytes;
  static constexpr int tapm_bytes = 32 * 4;                           // per tap: bit j = group j multiplies through it
  static constexpr int epi_bytes = NW * Conv16Epi<16 / NW, CAP>::wave_bytes;   // overlays the ring
  static_assert(epi_bytes <= ring_bytes, "epilogue tile must fit the ring");
  static constexpr int bytes = ring_bytes + nbr_bytes + tapm_bytes;
  static_assert(bytes <= 160 * 1024, "LDS of one compute unit");
};

// 16 bytes per lane, global -> VGPR, hidden from hipcc's scoreboard: the loads of the next step's weight fragments stay
// in flight behind the gathers of the step after it, and the loop waits with a counted vmcnt (cu_wait below) instead
// of the vmcnt(0) the compiler would place in front of the first use.
__device__ __forceinline__ void gload16(i32x4& dst, const void* src) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src) : "memory");
}
// wait until at most `allowed` (wave-uniform, >= 0) vector-memory operations of this wave are outstanding -- rounded down
// to an even immediate <= 16 (waiting for more than necessary is safe).  ONE asm statement with the choice inside it, and
// the fragment registers tied to it: no use of them can be scheduled above the wait, and there is no join of several asm
// results in front of which hipcc would copy registers whose data has not landed (separate statements under an
// if / else did exactly that).
#define ISF_CU_W1(N) "s_cmp_ge_u32 %[al], " #N "\n\ts_cbranch_scc0 " #N "0f\n\ts_waitcnt vmcnt(" #N ")\n\ts_branch 99f\n" #N "0:\n\t"
// (counts below 10 -- every count of the 8-wave shapes -- skip the upper half of the chain: each miss is a taken branch,
//  and six of them per step were 0.1 ms of the five 256 -> 256 launches at depth 2, profiles/r06_cu_asm.txt)
#define ISF_CU_WCHAIN                                                                                                 \
  "s_cmp_lt_u32 %[al], 10\n\ts_cbranch_scc1 98f\n\t"                                                                  \
  ISF_CU_W1(24) ISF_CU_W1(20) ISF_CU_W1(16) ISF_CU_W1(14) ISF_CU_W1(12) ISF_CU_W1(10) "98:\n\t" ISF_CU_W1(8) ISF_CU_W1(6)       \
      ISF_CU_W1(4) ISF_CU_W1(2) "s_waitcnt vmcnt(0)\n"                                                                \
                                "99:"
#define ISF_CU_WAIT4_ALL(bn) asm volatile("s_waitcnt vmcnt(0)" : "+v"(bn[0]), "+v"(bn[1]), "+v"(bn[2]), "+v"(bn[3])::"memory")
#define ISF_CU_WAIT8_ALL(bn)                                                                                         \
  asm volatile("s_waitcnt vmcnt(0)"                                                                                  \
               : "+v"(bn[0]), "+v"(bn[1]), "+v"(bn[2]), "+v"(bn[3]), "+v"(bn[4]), "+v"(bn[5]), "+v"(bn[6]), "+v"(bn[7]) \
               :                                                                                                     \
               : "memory")
#define ISF_CU_WAIT4(bn, allowed)                                                                                    \
  asm volatile(ISF_CU_WCHAIN : "+v"(bn[0]), "+v"(bn[1]), "+v"(bn[2]), "+v"(bn[3]) : [al] "s"(allowed) : "memory", "scc")
#define ISF_CU_WAIT8(bn, allowed)                                                                                    \
  asm volatile(ISF_CU_WCHAIN                                                                                         \
               : "+v"(bn[0]), "+v"(bn[1]), "+v"(bn[2]), "+v"(bn[3]), "+v"(bn[4]), "+v"(bn[5]), "+v"(bn[6]), "+v"(bn[7])  \
               : [al] "s"(allowed)                                                                                   \
               : "memory", "scc")

struct CuCursor {
  unsigned rem;   // taps of the current chunk not yet visited
  int tap, ch;
};

// NW: waves per workgroup (8: wave = 16 row groups x 32 columns, 128 accumulator registers, two waves per SIMD; 4: wave =
// 16 row groups x 64 columns, 256 accumulator registers (the AGPR half of the register file), one wave per SIMD: every A
// fragment is read from LDS by 4 waves instead of 8 and a row-group block is 12 MFMAs instead of 6 per fragment read).
// D: prefetch depth in steps (weights: D + 1 register sets, gathered rows: D + 1 LDS stages).  KNOCK: TIMING DIAGNOSTICS
// (results garbage): bit 1 = no gathers, bit 2 = no weight loads.
// PIPE (round 6): the multiply phase as a LOOP over the step's active row groups with the A fragments in two register sets
// used alternately (the loop is unrolled by two) and the accumulators picked by a switch -- see the multiply section.
// CAP (round 6): groups per unit.  16 = one workgroup per compute unit; 8 (with NW = 4: a wave = 8 groups x 64 columns, 128
// accumulator registers) = TWO workgroups per compute unit, each with its own barrier: they drift apart, and the load-issue
// phase of one (index reads, address selects, M0 set-ups, 8 weight loads -- in-order in front of its multiply phase, worth
// 0.3 of the 1.2 ms of the five 256 -> 256 launches by the knock-outs of profiles/r06_cu_asm.txt) runs under the other's MFMAs.
template <int CIN, int NW, int D, int KNOCK, bool PIPE = false, int CAP = kCuCapGroups, bool STAG = false>
__global__ __launch_bounds__(64 * NW, CAP == 8 ? 2 : 1) void spconv_cu_kernel(
    const uint4* __restrict__ xs, const int32_t* __restrict__ nbr, int nbr_stride, const uint4* __restrict__ wpk,
    const float* __restrict__ w_inv_scale, int K, const float* __restrict__ scale, const float* __restrict__ shift,
    const uint4* __restrict__ residual, uint4* __restrict__ ys, int n_out, int relu,
    const int32_t* __restrict__ group_masks, const int2* __restrict__ units, const int32_t* __restrict__ num_units) {
  using S = ConvCuSmem<D, NW, CAP>;
  constexpr int kRows = 16 * CAP;
  constexpr int NCH = CIN / 32, CH8 = CIN / 8, NS = D + 1;
  constexpr int NTW = 16 / NW;          // 16-column tiles per wave
  constexpr int NF = 2 * NTW;           // weight fragments per wave and step: [column tile][hi, lo]
  constexpr int GQ = CAP / NW;          // row groups whose rows a wave gathers: wave + NW q
  constexpr bool NOGATHER = (KNOCK & 1) != 0, NOWEIGHT = (KNOCK & 2) != 0;   // KNOCK & 4: no wait for the loads (step())
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* ring = reinterpret_cast<uint4*>(smem);
  int* nbr_l = reinterpret_cast<int*>(smem + S::ring_bytes);                       // [27][256]
  int* tapm_l = reinterpret_cast<int*>(smem + S::ring_bytes + S::nbr_bytes);       // [27]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kg = lane >> 4;

  // workgroup -> unit: XCD x (workgroups are dealt round-robin to the 8 XCDs) takes one contiguous range of units =
  // of rows, so its L2 holds the sliding window of y / z neighbour rows
  static_assert(CAP == 16 || (CAP == 8 && NW == 4 && PIPE), "shapes: 16 groups per unit, or 8 with 4 waves and the assembly multiply");
  const int U = *num_units;
  const int per_xcd = (U + 7) >> 3;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int u = xcd * per_xcd + slot;
  if (slot >= per_xcd || u >= U) return;
  const int2 un = units[u];
  const int g0 = un.x, n_rg = un.y;              // first 16-row group, groups (1..16)
  const int row0 = g0 * 16;
  const int row_end = min(row0 + n_rg * 16, n_out);

  // ---- prologue: neighbour table of the unit -> LDS; per-tap group masks
  {
    constexpr int NTHR = 64 * NW;
    constexpr int NB_IT = (kMaxTaps * kRows + NTHR - 1) / NTHR;
    int tmp[NB_IT];
#pragma unroll
    for (int it = 0; it < NB_IT; ++it) {
      const int i = tid + it * NTHR;
      const int k = i / kRows, r = i % kRows;
      tmp[it] = -1;
      if (k < K && row0 + r < row_end) tmp[it] = nbr[(size_t)k * nbr_stride + row0 + r];
    }
#pragma unroll
    for (int it = 0; it < NB_IT; ++it) {
      const int i = tid + it * NTHR;
      if (i < kMaxTaps * kRows) nbr_l[i] = tmp[it];
    }
  }
  unsigned unit_mask = 0;     // taps any group of the unit multiplies through
  if (tid < kMaxTaps) {
    unsigned m = 0;
    for (int j = 0; j < n_rg; ++j) m |= (((unsigned)group_masks[g0 + j] >> tid) & 1u) << j;
    tapm_l[tid] = (int)m;
  }
  for (int j = 0; j < n_rg; ++j) unit_mask |= (unsigned)group_masks[g0 + j];
  unit_mask = __builtin_amdgcn_readfirstlane(unit_mask);
  unsigned dm[GQ];            // tap masks of the groups whose rows this wave gathers
#pragma unroll
  for (int q = 0; q < GQ; ++q) {
    const int j = wave + NW * q;
    dm[q] = j < n_rg ? (unsigned)__builtin_amdgcn_readfirstlane(group_masks[g0 + j]) : 0u;
  }
  __syncthreads();
  const int ntaps = __popc(unit_mask);
  const int nsteps = ntaps * NCH;

  f32x4 acc[CAP][NTW];
#pragma unroll
  for (int j = 0; j < CAP; ++j)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[j][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // PIPE: the same accumulators as eight 16-register tuples pinned to v[128:255] (group j, tile nt = accT[j / 2], elements
  // 8 (j & 1) + 4 nt ..): what the assembly multiply phase works on; copied into acc[][] for the epilogue
  f32x16 accT[PIPE ? 8 : 1];
#pragma unroll
  for (int i = 0; i < (PIPE ? 8 : 1); ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) accT[i][e] = 0.f;

  auto advance = [&](CuCursor& c) {     // chunk outer, taps (set bits of unit_mask, increasing) inner
    if (c.rem == 0) {
      c.rem = unit_mask;
      ++c.ch;
    }
    c.tap = __ffs(c.rem) - 1;
    c.rem &= c.rem - 1;
  };

  // gather lane (isf_spconv_dma.hip): row lane >> 2 of a group, piece rotated so that the MFMA-layout read is
  // conflict-free; the reader (row col, k-group kg) finds its piece at position 4 col + ((kg + (col >> 2)) & 3)
  const int grow_l = lane >> 2;
  const int gpiece = ((lane & 3) - (grow_l >> 2)) & 3;
  const int rpos = 4 * col + ((kg + (col >> 2)) & 3);
  const unsigned ring_addr = __builtin_amdgcn_readfirstlane(lds_addr(ring));
  const uint4* zero = g_zero_line_cu;

  auto read_idx = [&](int tap, int (&ix)[GQ]) {     // this lane's gather rows through `tap` (groups wave + NW q)
#pragma unroll
    for (int q = 0; q < GQ; ++q) ix[q] = nbr_l[tap * kRows + (wave + NW * q) * 16 + grow_l];
  };
  auto dma_count = [&](int tap) -> int {
    int n = 0;
    if (!NOGATHER) {
#pragma unroll
      for (int q = 0; q < GQ; ++q) n += 2 * (int)((dm[q] >> tap) & 1u);
    }
    return n;
  };
  auto issue_A = [&](int tap, int ch, int stage, const int (&ix)[GQ]) {
    if (NOGATHER) return;
    const unsigned base = ring_addr + (unsigned)stage * (unsigned)S::stage_bytes;
#pragma unroll
    for (int q = 0; q < GQ; ++q) {
      if ((dm[q] >> tap) & 1u) {
        // KNOCK & 16 (timing only): every gather reads row 0 -- the same instructions, always cache hits
        const uint4* src = ix[q] >= 0 ? xs + ((size_t)((KNOCK & 16) ? 0 : ix[q]) * CH8 + ch * 4) * 2 + gpiece : zero + gpiece;
        glds16(src, base + (unsigned)(wave + NW * q) * 2048u);
        glds16(src + 4, base + (unsigned)(wave + NW * q) * 2048u + 1024u);
      }
    }
  };
  // Weight fragments of this wave, [column tile][hi, lo] -> [2 nt + h], in D + 1 register sets used in rotation: step t
  // multiplies with set t % (D + 1) while the loads of steps t + 1 .. t + D land in the others.  (One "next" set renamed
  // to a "current" set per step does not work with loads the compiler cannot see: it places the renaming copies in front
  // of the counted wait and copies registers whose data has not landed.)
  i32x4 bs[NS][NF];
#pragma unroll
  for (int k = 0; k < NS; ++k)
#pragma unroll
    for (int f = 0; f < NF; ++f) bs[k][f] = i32x4{0, 0, 0, 0};
  auto load_B = [&](int tap, int ch, i32x4 (&bn)[NF]) -> int {
    if (NOWEIGHT) return 0;
    // KNOCK & 8 (timing only): every step reads the weights of (tap 0, chunk 0) -- the same bytes, always cache hits
    const uint4* src = wpk + ((KNOCK & 8) ? (size_t)0 : ((size_t)tap * NCH + ch) * (kCuCout / 16) + NTW * wave) * 128 + lane;
#pragma unroll
    for (int f = 0; f < NF; ++f) gload16(bn[f], src + 64 * f);
    return NF;
  };

  // ---- the pipeline.  Group G(s) = {weights of step s + D, gathers of step s + D}, issued in that order at step s (the
  // fill issues G(-D) .. G(-1)); at the top of step t everything up to G(t - D) must have landed and the D - 1 younger
  // groups may stay in flight: the wait allows exactly their operation count (inq[]).
  CuCursor cg{0u, -1, -1};     // the step whose group is issued next
  CuCursor ci{0u, -1, -1};     // the step whose gather indices are read next (one ahead of cg)
  CuCursor cm{0u, -1, -1};     // the step whose group mask is read next (one ahead of the multiply)
  int idx[GQ];
#pragma unroll
  for (int q = 0; q < GQ; ++q) idx[q] = -1;
  unsigned m_cur = 0;
  int inq[D > 1 ? D - 1 : 1];   // operations of the D - 1 youngest groups, oldest first
#pragma unroll
  for (int i = 0; i < (D > 1 ? D - 1 : 1); ++i) inq[i] = 0;
  int issued_steps = 0;         // steps whose group has been issued
  auto issue_group = [&](i32x4 (&bn)[NF], int stage) -> int {   // the group of step `issued_steps`
    advance(cg);
    int n = load_B(cg.tap, cg.ch, bn);
    issue_A(cg.tap, cg.ch, stage, idx);
    n += dma_count(cg.tap);
    ++issued_steps;
    if (issued_steps < nsteps) {          // indices of the next group's rows
      advance(ci);
      read_idx(ci.tap, idx);
    }
    return n;
  };
  if (nsteps > 0) {
    advance(ci);
    read_idx(ci.tap, idx);
    advance(cm);
    m_cur = (unsigned)tapm_l[cm.tap];
#pragma unroll
    for (int k = 0; k < D; ++k) {         // G(k - D): step k
      int n = 0;
      if (k < nsteps) n = issue_group(bs[k], k);
      if (k >= 1) inq[k - 1] = n;         // G(-D) is the one the first wait drains
    }
  }
  int stage = 0;
  // one step: bn = the set holding B(t), bo = the set B(t + D) is loaded into
  auto step = [&](int t, i32x4 (&bn)[NF], i32x4 (&bo)[NF]) {
    int allowed = 0;
#pragma unroll
    for (int i = 0; i < D - 1; ++i) allowed += inq[i];
    if constexpr ((KNOCK & 4) != 0) {       // TIMING DIAGNOSTIC: no wait for the loads (results garbage): what is left of
                                            // the memory cost is issue + contention, not exposed latency.  The fragment
                                            // registers stay tied to the statement (hipcc must not reuse them while the
                                            // loads it cannot see are in flight); vmcnt(63) waits for nothing
      if constexpr (NF == 4) asm volatile("s_waitcnt vmcnt(63)" : "+v"(bn[0]), "+v"(bn[1]), "+v"(bn[2]), "+v"(bn[3])::"memory");
      else asm volatile("s_waitcnt vmcnt(63)" : "+v"(bn[0]), "+v"(bn[1]), "+v"(bn[2]), "+v"(bn[3]), "+v"(bn[4]), "+v"(bn[5]), "+v"(bn[6]), "+v"(bn[7])::"memory");
    } else if constexpr (D == 1) {          // nothing younger than G(t - 1) exists: drain
      if constexpr (NF == 4) ISF_CU_WAIT4_ALL(bn);
      else ISF_CU_WAIT8_ALL(bn);
    } else {
      if constexpr (NF == 4) ISF_CU_WAIT4(bn, allowed);
      else ISF_CU_WAIT8(bn, allowed);
    }
    __builtin_amdgcn_s_barrier();     // A(t) complete for every wave; every wave is done reading the stage of step t - 1
    asm volatile("" ::: "memory");
    const unsigned m = __builtin_amdgcn_readfirstlane(m_cur);
    if (t + 1 < nsteps) {
      advance(cm);
      m_cur = (unsigned)tapm_l[cm.tap];
    }
    auto issue_next = [&]() {
      int n = 0;
      if (issued_steps < nsteps) {
        int s_new = stage + D;
        if (s_new >= NS) s_new -= NS;
        n = issue_group(bo, s_new);
      }
#pragma unroll
      for (int i = 0; i + 1 < D - 1; ++i) inq[i] = inq[i + 1];
      if (D > 1) inq[D > 1 ? D - 2 : 0] = n;
    };
    // STAG (round 6): the two waves of a SIMD (wave w and w + NW / 2 land on the same SIMD) run the two halves of a step in
    // OPPOSITE order -- the first half of the workgroup issues the loads of step t + D and then multiplies, the second half
    // multiplies first -- so that between two barriers a SIMD's matrix pipe always has one wave in its multiply phase
    // while the other is in its issue phase (index reads, address selects, M0 set-ups, weight loads: in-order in front
    // of the MFMAs otherwise, and both waves of a SIMD at the same time because the barrier aligns them).  Needs D >= 2:
    // the late half's loads are issued only half a step before the next barrier.
    static_assert(!STAG || D >= 2, "staggered halves need a prefetch depth of two steps");
    // (two workgroups per CU, CAP = 8: a workgroup has one wave per SIMD, its SIMD partner is the wave of the CU's OTHER
    //  workgroup -- slots j and j + 32 of an XCD share a compute unit -- so the whole second-round workgroup runs late)
    const bool late = STAG && (CAP == 8 ? ((slot >> 5) & 1) != 0 : wave >= NW / 2);     // wave-uniform
    if (!late) issue_next();
    if (m) {
      const uint4* sa = ring + stage * (S::stage_bytes / 16) + rpos;
      h8 bh[NTW], bl[NTW];
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        bh[nt] = *reinterpret_cast<const h8*>(&bn[2 * nt]);
        bl[nt] = *reinterpret_cast<const h8*>(&bn[2 * nt + 1]);
      }
      // per accumulator: a_lo b_hi -> a_hi b_lo -> a_hi b_hi (the order of spconv_f16x3_kernel), the column tiles
      // interleaved so that consecutive MFMAs are independent
      auto mm = [&](f32x4 (&c)[NTW], const uint4 ahu, const uint4 alu) {
        const h8 ah = *reinterpret_cast<const h8*>(&ahu);
        const h8 al = *reinterpret_cast<const h8*>(&alu);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) c[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[nt], c[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) c[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[nt], c[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) c[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[nt], c[nt], 0, 0, 0);
      };
      if constexpr (PIPE) {
        // the hand-scheduled multiply phase (isf_spconv_cu_mult.h): accumulators pinned to v[128:255]
        static_assert(!PIPE || (NW == 8 && NTW == 2 && CAP == 16) || (NW == 4 && NTW == 4 && CAP == 8),
                      "the assembly multiply phase is written for 8 waves x 32 columns x 16 groups and 4 waves x 64 columns x 8 groups");
        const unsigned vb = lds_addr(sa);            // this lane's fragment position in group 0 of the stage
        i32x4 xh, xl, yh, yl;
        unsigned va;
        int t, t2;
        if constexpr (NTW == 2) {
          asm volatile(ISF_CUM_TEXT
                       : "+{v[128:143]}"(accT[0]), "+{v[144:159]}"(accT[1]), "+{v[160:175]}"(accT[2]),
                         "+{v[176:191]}"(accT[3]), "+{v[192:207]}"(accT[4]), "+{v[208:223]}"(accT[5]),
                         "+{v[224:239]}"(accT[6]), "+{v[240:255]}"(accT[7]), [xh] "=&v"(xh), [xl] "=&v"(xl),
                         [yh] "=&v"(yh), [yl] "=&v"(yl), [va] "=&v"(va), [t] "=&s"(t), [t2] "=&s"(t2)
                       : [b0h] "v"(bn[0]), [b0l] "v"(bn[1]), [b1h] "v"(bn[2]), [b1l] "v"(bn[3]), [vb] "v"(vb), [m] "s"(m)
                       : "scc", "memory");
        } else {
          asm volatile(ISF_CUM4_TEXT
                       : "+{v[128:143]}"(accT[0]), "+{v[144:159]}"(accT[1]), "+{v[160:175]}"(accT[2]),
                         "+{v[176:191]}"(accT[3]), "+{v[192:207]}"(accT[4]), "+{v[208:223]}"(accT[5]),
                         "+{v[224:239]}"(accT[6]), "+{v[240:255]}"(accT[7]), [xh] "=&v"(xh), [xl] "=&v"(xl),
                         [yh] "=&v"(yh), [yl] "=&v"(yl), [va] "=&v"(va), [t] "=&s"(t), [t2] "=&s"(t2)
                       : [b0h] "v"(bn[0]), [b0l] "v"(bn[1]), [b1h] "v"(bn[2]), [b1l] "v"(bn[3]), [b2h] "v"(bn[4]),
                         [b2l] "v"(bn[5]), [b3h] "v"(bn[6]), [b3l] "v"(bn[7]), [vb] "v"(vb), [m] "s"(m)
                       : "scc", "memory");
        }
      } else {
      const int jf = __ffs(m) - 1;
      uint4 ah_n = sa[jf * 128], al_n = sa[jf * 128 + 64];
#pragma unroll
      for (int j = 0; j < CAP; ++j) {
        if ((m >> j) & 1u) {
          const uint4 ahu = ah_n, alu = al_n;
          const unsigned rest = j < CAP - 1 ? m >> (j + 1) : 0u;
          if (rest) {                            // the next group's fragments are read while this one multiplies
            const int jn = j + __ffs(rest);
            ah_n = sa[jn * 128];
            al_n = sa[jn * 128 + 64];
          }
          mm(acc[j], ahu, alu);
        }
      }
      }
    }
    if (late) issue_next();
    if (++stage == NS) stage = 0;
  };
  for (int t = 0; t < nsteps; t += NS) {
#pragma unroll
    for (int k = 0; k < NS; ++k)
      if (t + k < nsteps) step(t + k, bs[k], bs[(k + D) % NS]);
  }
  {
    if constexpr (NF == 4) ISF_CU_WAIT4_ALL(bs[0]);
    else ISF_CU_WAIT8_ALL(bs[0]);
  }
  __syncthreads();   // every wave is done with the ring -> reuse as the epilogue transpose tiles
  if constexpr (PIPE) {
    // the last MFMAs were issued from assembly: hipcc's hazard recogniser has not seen them (XDL write -> VALU / LDS read)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int j = 0; j < CAP; ++j)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (NTW == 2) acc[j][nt][e] = accT[j >> 1][8 * (j & 1) + 4 * nt + e];
          else acc[j][nt][e] = accT[j][4 * nt + e];
        }
  }

  float* tile_l = reinterpret_cast<float*>(smem) + wave * (Conv16Epi<NTW, CAP>::wave_bytes / 4);
  conv16_epilogue<NTW, CAP, false>(acc, tile_l, lane, row0, 16 * NTW * wave, kCuCout, *w_inv_scale, scale, shift,
                                            residual, ys, row_end, relu, n_rg);
}

// ------------------------------------------------------------------------------------------------------ unit plan
// group_masks[g] bit k: some row of 16-row group g has a neighbour through tap k; work[g] = popcount
__global__ __launch_bounds__(256) void cu_group_mask_kernel(const int32_t* __restrict__ nbr, int nbr_stride, int K,
                                                            int n_out, int n_groups, int32_t* __restrict__ masks,
                                                            int32_t* __restrict__ work) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int gbase = (blockIdx.x * 4 + wv) * 4;             // this wave's four groups = 64 rows
  if (gbase >= n_groups) return;
  const int row = gbase * 16 + lane;
  unsigned m[4] = {0u, 0u, 0u, 0u};
  for (int k = 0; k < K; ++k) {
    const bool has = row < n_out && nbr[(size_t)k * nbr_stride + row] >= 0;
    const unsigned long long b = __ballot(has);
#pragma unroll
    for (int q = 0; q < 4; ++q) m[q] |= (((b >> (16 * q)) & 0xffffull) ? 1u : 0u) << k;
  }
  if (lane < 4 && gbase + lane < n_groups) {
    const unsigned mine = lane == 0 ? m[0] : lane == 1 ? m[1] : lane == 2 ? m[2] : m[3];
    masks[gbase + lane] = (int)mine;
    work[gbase + lane] = __popc(mine);
  }
}

// One workgroup: inclusive prefix of the work, the balanced cuts, the split of over-long units, the unit table.
// scratch: W [n_groups] | cuts [U0 + 1] | offs [U0 + 1]
__global__ __launch_bounds__(1024) void cu_plan_kernel(const int32_t* __restrict__ work, int n_groups, int U0, int cap,
                                                       int32_t* W, int32_t* cuts, int32_t* offs,
                                                       int2* __restrict__ units,
                                                       int32_t* __restrict__ num_units) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  auto block_scan = [&](int v, int& total) -> int {   // inclusive scan over the 1024 threads
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(x, d, 64);
      if (lane >= d) x += o;
    }
    __syncthreads();
    if (lane == 63) wsum[wv] = x;
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int s = wsum[w];
      if (w < wv) before += s;
      tot += s;
    }
    total = tot;
    return x + before;
  };
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n_groups; base += 1024) {
    const int i = base + tid;
    int tot;
    const int inc = block_scan(i < n_groups ? work[i] : 0, tot);
    const int carry = carry_s;
    if (i < n_groups) W[i] = inc + carry;
    __syncthreads();
    if (tid == 0) carry_s = carry + tot;
    __syncthreads();
  }
  __threadfence_block();
  __syncthreads();
  for (int uu = tid; uu <= U0; uu += 1024) cuts[uu] = conv_cu_cut(W, n_groups, U0, uu);
  __syncthreads();
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < U0; base += 1024) {
    const int uu = base + tid;
    int tot;
    const int p = uu < U0 ? conv_cu_pieces(cuts[uu + 1] - cuts[uu], cap) : 0;
    const int inc = block_scan(p, tot);
    const int carry = carry_s;
    if (uu < U0) offs[uu] = inc - p + carry;
    __syncthreads();
    if (tid == 0) carry_s = carry + tot;
    __syncthreads();
  }
  if (tid == 0) *num_units = carry_s;
  for (int uu = tid; uu < U0; uu += 1024) {
    const int c = cuts[uu], len = cuts[uu + 1] - c, P = conv_cu_pieces(len, cap), o = offs[uu];
    for (int p = 0; p < P; ++p) {
      int g0, ng;
      conv_cu_piece(c, len, p, g0, ng, cap);
      units[o + p] = make_int2(g0, ng);
    }
  }
}

int conv_group_masks_impl(const int32_t* nbr, int nbr_stride, int K, int n_out, int32_t* masks, int32_t* work,
                          hipStream_t st) {
  const int ng = ceil_div(n_out, 16);
  hipLaunchKernelGGL(cu_group_mask_kernel, dim3(ceil_div(ng, 16)), dim3(256), 0, st, nbr, nbr_stride, K, n_out, ng, masks, work);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

static int cu_count() {
  static std::atomic<int> cus{0};
  int c = cus.load(std::memory_order_acquire);
  if (c == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        c <= 0)
      c = 256;
    cus.store(c, std::memory_order_release);
  }
  return c;
}

bool sparse_conv_cu_supported(int c_in, int c_out) { return c_out == kCuCout && (c_in == 128 || c_in == 256); }

// the unit shape of a kernel variant: 9, 10, 14, 15 = two 4-wave workgroups per compute unit over units of <= 8 groups
int conv_cu_variant_cap(int variant) { return (variant == 9 || variant == 15) ? kCuCapGroups8 : kCuCapGroups; }
static int cu_slots(int cap) { return cap == kCuCapGroups8 ? 2 * cu_count() : cu_count(); }

size_t conv_cu_plan_ints(int n_out) {   // int32 entries a plan of n_out rows needs (masks, work, W, cuts, offs, units, count); either shape
  const int ng = ceil_div(n_out > 0 ? n_out : 1, 16);
  size_t most = 0;
  for (int cap : {kCuCapGroups, kCuCapGroups8}) {
    const int U0 = conv_cu_balanced_units(ng, cu_slots(cap), cap), UM = conv_cu_max_units(ng, cu_slots(cap), cap);
    const size_t n = (size_t)ng * 3 + 2 * (size_t)(U0 + 1) + 2 * (size_t)UM + 64;
    most = n > most ? n : most;
  }
  return most;
}

int conv_cu_plan_impl(const int32_t* nbr, int nbr_stride, int K, int n_out, int32_t* buf, ConvCuPlan* plan,
                      hipStream_t st, int cap) {
  ISF_REQUIRE(nbr && buf && plan && n_out > 0 && K >= 1 && K <= kMaxTaps && nbr_stride >= n_out, ISF_ERR_ARG,
              "sparse_conv_cu_plan: bad arguments");
  ISF_REQUIRE(cap == kCuCapGroups || cap == kCuCapGroups8, ISF_ERR_ARG, "sparse_conv_cu_plan: cap %d", cap);
  const int ng = ceil_div(n_out, 16), slots = cu_slots(cap);
  const int U0 = conv_cu_balanced_units(ng, slots, cap), UM = conv_cu_max_units(ng, slots, cap);
  int32_t* masks = buf;
  int32_t* work = masks + ng;
  int32_t* W = work + ng;
  int32_t* cuts = W + ng;
  int32_t* offs = cuts + (U0 + 1);
  int32_t* cnt = offs + (U0 + 1);
  int2* units = reinterpret_cast<int2*>(cnt + 2 + ((cnt + 2 - buf) & 1));   // 8-byte aligned behind the count
  hipLaunchKernelGGL(cu_group_mask_kernel, dim3(ceil_div(ng, 16)), dim3(256), 0, st, nbr, nbr_stride, K, n_out, ng, masks,
                     work);
  hipLaunchKernelGGL(cu_plan_kernel, dim3(1), dim3(1024), 0, st, work, ng, U0, cap, W, cuts, offs, units, cnt);
  ISF_LAUNCH_CHECK();
  plan->group_masks = masks;
  plan->units = units;
  plan->num_units = cnt;
  plan->max_units = UM;
  plan->n_out = n_out;
  plan->variant = 0;
  plan->cap = cap;
  return ISF_OK;
}

int sparse_conv_forward_cu_impl(const void* xs, int c_in, const void* packed16, int K, int c_out, const int32_t* nbr,
                                int nbr_stride, int n_out, const float* scale, const float* shift, const void* residual,
                                int relu, void* ys, const ConvCuPlan& plan, hipStream_t st) {
  if (n_out <= 0) return ISF_OK;
  ISF_REQUIRE(sparse_conv_cu_supported(c_in, c_out), ISF_ERR_UNSUPPORTED, "sparse_conv_cu: (Cin,Cout)=(%d,%d) not built",
              c_in, c_out);
  ISF_REQUIRE(K >= 1 && K <= kMaxTaps && nbr_stride >= n_out, ISF_ERR_ARG, "sparse_conv_cu: bad rulebook");
  ISF_REQUIRE(plan.group_masks && plan.units && plan.num_units && plan.n_out == n_out && plan.max_units > 0, ISF_ERR_ARG,
              "sparse_conv_cu: the unit plan was built for %d rows, the launch has %d", plan.n_out, n_out);
  ISF_REQUIRE(plan.cap == conv_cu_variant_cap(plan.variant), ISF_ERR_ARG,
              "sparse_conv_cu: variant %d works on units of <= %d groups, the plan was cut for %d", plan.variant,
              conv_cu_variant_cap(plan.variant), plan.cap);
  const uint4* w = reinterpret_cast<const uint4*>(packed16);
  const float* winv = reinterpret_cast<const float*>(reinterpret_cast<const char*>(packed16) + (size_t)K * c_in * c_out * 4);
  const dim3 grid(8 * ceil_div(plan.max_units, 8));
  // variant (0 = round 4's production shape): 1 / 2 / 3 = no gathers / no weight loads / neither (results garbage, timing
  // only); 4 / 5 = 4 waves at prefetch depth 1 / 2, 6 / 7 = 8 waves at depth 1 / 2 (results valid); round 6, assembly
  // multiply phase: 8 = 8 waves, one workgroup per CU; 9 / 10 = 4 waves x 64 columns, units of <= 8 groups, TWO workgroups
  // per CU, depth 1 / 2 (valid); 11 / 12 / 13 = variant 8 without both / gathers / weights, 14 / 15 = variant 9 without
  // both / gathers (timing only)
#define ISF_CU_LAUNCH(CI, WW, DD, KK, PP, CC, SS)                                                                                 \
  do {                                                                                                                  \
    constexpr int smem_bytes = ConvCuSmem<DD, WW, CC>::bytes;                                                               \
    auto kern = spconv_cu_kernel<CI, WW, DD, KK, PP, CC, SS>;                                                                       \
    static std::atomic<int> attr_set{0};                                                                                \
    if (attr_set.load(std::memory_order_acquire) == 0) {                                                                \
      ISF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                      smem_bytes));                                                                     \
      attr_set.store(1, std::memory_order_release);                                                                     \
    }                                                                                                                   \
    hipLaunchKernelGGL(kern, grid, dim3(64 * WW), smem_bytes, st, reinterpret_cast<const uint4*>(xs), nbr,           \
                       nbr_stride, w, winv, K, scale, shift, reinterpret_cast<const uint4*>(residual),                  \
                       reinterpret_cast<uint4*>(ys), n_out, relu, plan.group_masks, plan.units, plan.num_units);        \
  } while (0)
#define ISF_CU_VARIANTS(CI)                                                                                             \
  switch (plan.variant) {                                                                                               \
    case 0: ISF_CU_LAUNCH(CI, kCuWavesProd, kCuDepth, 0, false, 16, false); break;                                             \
    case 1: ISF_CU_LAUNCH(CI, kCuWavesProd, kCuDepth, 1, false, 16, false); break;                                             \
    case 2: ISF_CU_LAUNCH(CI, kCuWavesProd, kCuDepth, 2, false, 16, false); break;                                             \
    case 3: ISF_CU_LAUNCH(CI, kCuWavesProd, kCuDepth, 3, false, 16, false); break;                                             \
    case 4: ISF_CU_LAUNCH(CI, 4, 1, 0, false, 16, false); break;                                                               \
    case 5: ISF_CU_LAUNCH(CI, 4, 2, 0, false, 16, false); break;                                                               \
    case 6: ISF_CU_LAUNCH(CI, 8, 1, 0, false, 16, false); break;                                                               \
    case 7: ISF_CU_LAUNCH(CI, 8, 2, 0, false, 16, false); break;                                                               \
    case 8: ISF_CU_LAUNCH(CI, 8, 1, 0, true, 16, false); break;                                                                \
    case 9: ISF_CU_LAUNCH(CI, 4, 1, 0, true, 8, false); break;                                                                 \
    case 10: ISF_CU_LAUNCH(CI, 8, 2, 0, true, 16, true); break;                                                                \
    case 11: ISF_CU_LAUNCH(CI, 8, 1, 24, true, 16, false); break;                                                               \
    case 12: ISF_CU_LAUNCH(CI, 8, 1, 4, true, 16, false); break;                                                               \
    case 13: ISF_CU_LAUNCH(CI, 8, 1, 16, true, 16, false); break;                                                               \
    case 14: ISF_CU_LAUNCH(CI, 8, 2, 0, true, 16, false); break;                                                                \
    case 15: ISF_CU_LAUNCH(CI, 4, 2, 0, true, 8, true); break;                                                                \
    default: ISF_REQUIRE(false, ISF_ERR_ARG, "sparse_conv_cu: variant %d", plan.variant);                               \
  }
  if (c_in == 128) { ISF_CU_VARIANTS(128) } else { ISF_CU_VARIANTS(256) }
#undef ISF_CU_VARIANTS
#undef ISF_CU_LAUNCH
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

}  // namespace isf

extern "C" {

int isf_sparse_conv_cu_plan_ints(int num_out, size_t* num_ints) {
  ISF_REQUIRE(num_ints && num_out >= 0, ISF_ERR_ARG, "sparse_conv_cu_plan_ints: bad arguments");
  *num_ints = isf::conv_cu_plan_ints(num_out);
  return ISF_OK;
}

int isf_sparse_conv_cu_plan(const int32_t* nbr, int nbr_stride, int num_taps, int num_out, int32_t* plan_buf,
                            isf_conv_cu_plan* plan, isf_stream_t stream) {
  ISF_REQUIRE(plan, ISF_ERR_ARG, "sparse_conv_cu_plan: null plan");
  isf::ConvCuPlan p;
  const int variant = plan->variant;    // INPUT: the kernel variant the plan is for decides the unit shape (0: 16 groups)
  ISF_TRY(isf::conv_cu_plan_impl(nbr, nbr_stride, num_taps, num_out, plan_buf, &p, isf::as_stream(stream),
                                 isf::conv_cu_variant_cap(variant >= 0 && variant < 16 ? variant : 0)));
  plan->group_masks = p.group_masks;
  plan->units = reinterpret_cast<const int32_t*>(p.units);
  plan->num_units = p.num_units;
  plan->max_units = p.max_units;
  plan->num_out = p.n_out;
  plan->variant = variant >= 0 && variant < 16 ? variant : 0;
  plan->cap = p.cap;
  return ISF_OK;
}

int isf_sparse_conv_forward_cu(const void* features_split, int num_in, int c_in, const void* packed16, int num_taps,
                               int c_out, const int32_t* nbr, int nbr_stride, int num_out, const float* scale,
                               const float* shift, const void* residual_split, int relu, void* out_split,
                               const isf_conv_cu_plan* plan, isf_stream_t stream) {
  ISF_REQUIRE(num_in >= 0 && num_out >= 0 && c_in > 0 && c_out > 0 && num_taps > 0 && plan, ISF_ERR_ARG,
              "sparse_conv_forward_cu: bad arguments");
  if (num_out == 0) return ISF_OK;
  ISF_REQUIRE(features_split && packed16 && nbr && out_split && ((scale == nullptr) == (shift == nullptr)), ISF_ERR_ARG,
              "sparse_conv_forward_cu: null pointer");
  isf::ConvCuPlan p;
  p.group_masks = plan->group_masks;
  p.units = reinterpret_cast<const int2*>(plan->units);
  p.num_units = plan->num_units;
  p.max_units = plan->max_units;
  p.n_out = plan->num_out;
  p.variant = plan->variant;
  p.cap = plan->cap;
  return isf::sparse_conv_forward_cu_impl(features_split, c_in, packed16, num_taps, c_out, nbr, nbr_stride, num_out, scale,
                                          shift, residual_split, relu, out_split, p, isf::as_stream(stream));
}

// the plan arithmetic on the host (tests / tools: no device work): work [num_groups] -> units [max_units][2]
int isf_sparse_conv_cu_plan_host(const int32_t* work, int num_groups, int cus, int32_t* units, int max_units,
                                 int* num_units) {
  ISF_REQUIRE(work && units && num_units && num_groups > 0 && cus > 0, ISF_ERR_ARG, "sparse_conv_cu_plan_host: bad arguments");
  ISF_REQUIRE(max_units >= isf::conv_cu_max_units(num_groups, cus), ISF_ERR_ARG,
              "sparse_conv_cu_plan_host: units holds %d entries, %d needed", max_units, isf::conv_cu_max_units(num_groups, cus));
  std::vector<int32_t> W((size_t)num_groups);
  long long s = 0;
  for (int i = 0; i < num_groups; ++i) {
    s += work[i];
    W[(size_t)i] = (int32_t)s;
  }
  const int U0 = isf::conv_cu_balanced_units(num_groups, cus);
  int n = 0;
  for (int uu = 0; uu < U0; ++uu) {
    const int c = isf::conv_cu_cut(W.data(), num_groups, U0, uu), e = isf::conv_cu_cut(W.data(), num_groups, U0, uu + 1);
    const int P = isf::conv_cu_pieces(e - c);
    for (int p = 0; p < P; ++p) {
      int g0, ng;
      isf::conv_cu_piece(c, e - c, p, g0, ng);
      units[2 * n] = g0;
      units[2 * n + 1] = ng;
      ++n;
    }
  }
  *num_units = n;
  return ISF_OK;
}

int isf_sparse_conv_cu_max_units(int num_groups, int cus) { return isf::conv_cu_max_units(num_groups, cus); }

}  // extern "C"
