// isf_spconv_deep.hip -- the f16x3 sparse convolution of the DEEP layers (>= 128 input channels, 128 / 256 output columns:
// levels 2 - 4 of the encoder, 2.2 of the headline step's 4.2 ms) with a software-pipelined, hand-scheduled step.
//
// What the tile kernel (isf_spconv16.hip) is bound by, measured (profiles/r06_att_256.txt, its phase trace with the
// assembly multiply section): a wave's step is wait 920 -> barrier 80 -> ISSUE 976 -> multiply 1 072 cycles, one after the
// other; the issue phase is eight vector-memory instructions that stall ~125 cycles each because the CU's address unit is
// saturated -- a gather in the MFMA operand layout puts four different rows into every lane quad (four cache lines per
// quad, ~45 address cycles per instruction: 12 waves x (4 x 45 + 4 x 12) = 2 700 of the ~3 300 cycles of a round of
// steps) -- and the matrix pipe (3 x 768 cycles per round) waits for it.  Both halves of that are removed here:
//   * GATHERS BY LDS-DMA, A LANE QUAD = 64 CONTIGUOUS BYTES OF ONE ROW (the narrow layers' pattern, isf_spconv_dma.hip):
//     16 address cycles per instruction; the rows land in a wave-private 4-KiB LDS transit and are read from there in
//     the MFMA layout (4 conflict-free ds_read_b128) at the top of the step;
//   * ONE INSTRUCTION STREAM PER STEP (isf_spconv_deep_asm.h): the products of step s with the gathers and the weight
//     pieces of step s + 1 issued in between, one per three MFMAs -- no issue phase, no burst at the address unit.
// No neighbour table in LDS (a lane's gather-row index comes from global memory two steps ahead: 64 bytes per row group
// and step): 32 KiB weight ring + 16 KiB transit = 48 KiB per workgroup, three workgroups per CU like the tile kernel,
// whose tile plan, tile order, column-block / XCD mapping, epilogue and ORDER OF OPERATIONS per accumulator (chunk outer,
// taps inner, a_lo b_hi -> a_hi b_lo -> a_hi b_hi) it keeps: results are BIT-IDENTICAL to spconv_f16x3_kernel.
// Replaces, like it, the reference's per-tap gather -> GEMM -> scatter-add (spconv_ops.h:260-361).
#include "isf_spconv16.h"
#include "isf_spconv_deep_asm.h"

#include <atomic>

namespace isf {

typedef int i32x4d __attribute__((ext_vector_type(4)));

__device__ uint4 g_zero_line_deep[8];   // 128 zero bytes: what a row without a neighbour reads

template <int NW>
struct ConvDeepSmem {
  static constexpr int NT = 8, RG = 2;
  static constexpr int TM = 16 * RG * NW;
  static constexpr int bbuf_bytes = 2 * NT * 2048;        // double-buffered weight stage
  static constexpr int transit_bytes = NW * RG * 2048;    // per wave: [row group][hi, lo][64 pieces of 16 B]
  static constexpr int epi_bytes = NW * Conv16Epi<NT, RG>::wave_bytes;
  static constexpr int main_bytes = bbuf_bytes + transit_bytes;
  static constexpr int work_bytes = main_bytes > epi_bytes ? main_bytes : epi_bytes;
  static constexpr int bytes = work_bytes + 256;          // + wave masks
};

template <int CIN, int NW>
__global__ __launch_bounds__(64 * NW, 3) void spconv_deep_kernel(
    const uint4* __restrict__ xs, const int32_t* __restrict__ nbr, int nbr_stride, const uint4* __restrict__ wpk,
    const float* __restrict__ w_inv_scale, int K, int cout, const float* __restrict__ scale,
    const float* __restrict__ shift, const uint4* __restrict__ residual, uint4* __restrict__ ys, int n_out, int relu,
    Conv16Plan plan, const int32_t* __restrict__ order) {
  constexpr int NT = 8, RG = 2;
  using S = ConvDeepSmem<NW>;
  constexpr int TM = S::TM, WR = 16 * RG;
  constexpr int NCH = CIN / 32, CH8 = CIN / 8, BN = 16 * NT;
  static_assert(NW == 4, "a wave's weight share is one run of four 1-KiB pieces");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* bbuf = reinterpret_cast<uint4*>(smem);                                   // [2][NT][2][64]
  int* misc = reinterpret_cast<int*>(smem + S::work_bytes);                       // [NW] wave masks

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
  unsigned rgm[RG] = {0u, 0u};
  {
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
  const int nsteps = __popc(wg_mask) * NCH;

  f32x4 acc[RG][NT];
#pragma unroll
  for (int rg = 0; rg < RG; ++rg)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[rg][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // gather lane: row (lane >> 2) of a 16-row group, piece rotated so that the MFMA-layout read is conflict-free
  // (isf_spconv_dma.hip): LDS position 4 r + t holds piece (t - (r >> 2)) & 3 of row r; the reader (row col, k-group
  // kg) finds its piece at position 4 col + ((kg + (col >> 2)) & 3)
  const int grow_l = lane >> 2;
  const int gpiece = ((lane & 3) - (grow_l >> 2)) & 3;
  const int rpos = 4 * col + ((kg + (col >> 2)) & 3);
  const unsigned bbuf_addr = __builtin_amdgcn_readfirstlane(lds_addr(bbuf));
  const unsigned transit_addr = __builtin_amdgcn_readfirstlane(lds_addr(smem + S::bbuf_bytes) + (unsigned)wave * (RG * 2048u));
  const unsigned va = transit_addr + (unsigned)rpos * 16u;          // this lane's fragment position in the transit
  const uint4* zero = g_zero_line_deep;
  constexpr int PW = NT * 128 / NW;                                 // 16-byte weight pieces per wave and step (256)

  // step s -> (chunk, tap): chunk outer, taps (set bits of wg_mask, increasing) inner -- the tile kernel's order
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
  // this lane's gather-row index through `tap`, per row group (-1: none / group without the tap)
  auto load_idx = [&](int tap, int (&idx)[RG]) {
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      const int row = row0w + rg * 16 + grow_l;
      idx[rg] = -1;
      if (((rgm[rg] >> tap) & 1u) && row < row_end) idx[rg] = nbr[(size_t)tap * nbr_stride + row];
    }
  };
  auto row_ptr = [&](int idx, int ch) -> const uint4* {
    return idx >= 0 ? xs + ((size_t)idx * CH8 + ch * 4) * 2 + gpiece : zero + gpiece;
  };
  auto weight_ptr = [&](int tap, int ch) -> const uint4* {
    return wpk + (((size_t)tap * NCH + ch) * ntiles_total + cb * NT) * 128 + wave * PW + lane;
  };

  Cursor c0{0u, -1, -1}, c1{0u, -1, -1}, c2{0u, -1, -1};
  int i1[RG] = {-1, -1}, i2[RG] = {-1, -1};
  if (nsteps > 0) {     // step 0's rows and weights the plain way; the indices of step 1
    advance(c0);
    int i0[RG];
    load_idx(c0.tap, i0);
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      if ((rgm[rg] >> c0.tap) & 1u) {
        const uint4* src = row_ptr(i0[rg], c0.ch);
        glds16(src, transit_addr + (unsigned)(rg * 2) * 1024u);
        glds16(src + 4, transit_addr + (unsigned)(rg * 2 + 1) * 1024u);
      }
    }
    glds16_run<PW / 64>(weight_ptr(c0.tap, c0.ch), bbuf_addr + (unsigned)(wave * PW) * 16u);
    c1 = c0;
    if (nsteps > 1) {
      advance(c1);
      load_idx(c1.tap, i1);
    }
  }
  for (int s = 0; s < nsteps; ++s) {
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's rows and weight share of step s, the indices of step s + 1
    __syncthreads();                      // the weights of step s are complete; everyone is done with the other buffer
    if (s > 0) {                          // (rotated here, behind the wait hipcc knows: it adds none of its own)
      c0 = c1;
      c1 = c2;
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) i1[rg] = i2[rg];
    }
    const bool more = s + 1 < nsteps;
    // step s + 1: what to gather, from where
    const unsigned g0 = more ? (rgm[0] >> c1.tap) & 1u : 0u, g1 = more ? (rgm[1] >> c1.tap) & 1u : 0u;
    const uint4* p0 = row_ptr(g0 ? i1[0] : -1, c1.ch);
    const uint4* p1 = row_ptr(g1 ? i1[1] : -1, c1.ch);
    const uint4* bsrc = weight_ptr(more ? c1.tap : c0.tap, more ? c1.ch : c0.ch);
    const unsigned bdst = (unsigned)__builtin_amdgcn_readfirstlane((int)(bbuf_addr + (unsigned)(((s + 1) & 1) * (NT * 128) + wave * PW) * 16u));
    // the indices of step s + 2 (in flight over this step; landed at the next vmcnt(0))
    c2 = c1;
    if (s + 2 < nsteps) {
      advance(c2);
      load_idx(c2.tap, i2);
    }
    // (readfirstlane: the values are wave-uniform, but hipcc must also KNOW it to put them into scalar registers)
    const unsigned fl = (unsigned)__builtin_amdgcn_readfirstlane((int)(g0 | (g1 << 1) | ((more ? 1u : 0u) << 2) |
                                                                       (((rgm[0] >> c0.tap) & 1u) << 3) |
                                                                       (((rgm[1] >> c0.tap) & 1u) << 4)));
    const unsigned vb = bbuf_addr + (unsigned)((s & 1) * (NT * 128) + lane) * 16u;
    unsigned m0s;
    asm volatile(ISF_DA_TEXT
                 : [c00] "+v"(acc[0][0]), [c01] "+v"(acc[0][1]), [c02] "+v"(acc[0][2]), [c03] "+v"(acc[0][3]),
                   [c04] "+v"(acc[0][4]), [c05] "+v"(acc[0][5]), [c06] "+v"(acc[0][6]), [c07] "+v"(acc[0][7]),
                   [c10] "+v"(acc[1][0]), [c11] "+v"(acc[1][1]), [c12] "+v"(acc[1][2]), [c13] "+v"(acc[1][3]),
                   [c14] "+v"(acc[1][4]), [c15] "+v"(acc[1][5]), [c16] "+v"(acc[1][6]), [c17] "+v"(acc[1][7]),
                   [m0s] "=&s"(m0s)
                 : [va] "v"(va), [vb] "v"(vb), [p0] "v"(p0), [p1] "v"(p1), [bsrc] "v"(bsrc), [fl] "s"(fl),
                   [tdst] "s"(transit_addr), [bdst] "s"(bdst)
                 : ISF_DA_CLOBBERS);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  // the last MFMAs were issued from assembly: hipcc's hazard recogniser has not seen them (XDL write -> VALU / LDS read)
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  __syncthreads();  // all waves done with the weight buffers -> reuse as the epilogue transpose tile

  float* tile_l = reinterpret_cast<float*>(smem) + wave * (Conv16Epi<NT, RG>::wave_bytes / 4);
  conv16_epilogue<NT, RG, false>(acc, tile_l, lane, row0w, cb * BN, cout, *w_inv_scale, scale, shift, residual, ys,
                                 row_end, relu, half_tile ? RG / 2 : RG);
}

bool sparse_conv_deep_supported(int c_in, int c_out) {
  return (c_in == 128 || c_in == 256) && (c_out == 128 || c_out == 256);
}

// the tile kernel's launch (launch16<CIN, 8, 2, 4>): same plan, same tile order tables, same query
template <int CIN>
static int launch_deep(bool balance, bool table, const uint4* xs, const uint4* wpk, const float* winv, int K, int cout,
                       const int32_t* nbr, int nbr_stride, int n_out, const float* scale, const float* shift,
                       const uint4* residual, int relu, uint4* ys, hipStream_t st, const int32_t* order,
                       Conv16LaunchInfo* query) {
  constexpr int NW = 4;
  using S = ConvDeepSmem<NW>;
  auto kern = spconv_deep_kernel<CIN, NW>;
  static std::atomic<int> wgs_per_cu{0}, cus_per_xcd{0};
  if (wgs_per_cu.load(std::memory_order_acquire) == 0) {
    if (S::bytes > 48 * 1024)
      ISF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, S::bytes));
    int dev = 0, cus = 0, occ = 0;
    ISF_HIP_TRY(hipGetDevice(&dev));
    ISF_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    ISF_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64 * NW, S::bytes));
    cus_per_xcd.store(cus >= 8 ? cus / 8 : 1, std::memory_order_relaxed);
    wgs_per_cu.store(occ > 0 ? occ : 1, std::memory_order_release);
  }
  const int ncb = cout / 128;
  ISF_REQUIRE(ncb == 1 || ncb == 2, ISF_ERR_UNSUPPORTED, "sparse_conv_deep: %d column blocks", ncb);
  Conv16Plan plan = conv16_plan(n_out, S::TM, ncb, wgs_per_cu.load(std::memory_order_relaxed),
                                cus_per_xcd.load(std::memory_order_relaxed), balance);
  if (table && !query) plan = Conv16Plan{wgs_per_cu.load(std::memory_order_relaxed) * cus_per_xcd.load(std::memory_order_relaxed),
                                         -1, plan.part_rows};
  if (query) {
    *query = Conv16LaunchInfo{plan.full, plan.half, plan.part_rows, S::TM, ncb, wgs_per_cu.load(std::memory_order_relaxed),
                              cus_per_xcd.load(std::memory_order_relaxed)};
    return ISF_OK;
  }
  hipLaunchKernelGGL(kern, dim3(conv16_grid_blocks(plan)), dim3(64 * NW), S::bytes, st, xs, nbr, nbr_stride, wpk, winv, K,
                     cout, scale, shift, residual, ys, n_out, relu, plan, order);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int sparse_conv_forward_deep_impl(bool balance, bool table, const uint4* xs, int c_in, const uint4* wpk, const float* winv,
                                  int K, int c_out, const int32_t* nbr, int nbr_stride, int n_out, const float* scale,
                                  const float* shift, const uint4* residual, int relu, uint4* ys, hipStream_t st,
                                  const int32_t* order, Conv16LaunchInfo* query) {
  ISF_REQUIRE(sparse_conv_deep_supported(c_in, c_out), ISF_ERR_UNSUPPORTED, "sparse_conv_deep: (Cin,Cout)=(%d,%d) not built",
              c_in, c_out);
  if (c_in == 128)
    return launch_deep<128>(balance, table, xs, wpk, winv, K, c_out, nbr, nbr_stride, n_out, scale, shift, residual, relu, ys,
                            st, order, query);
  return launch_deep<256>(balance, table, xs, wpk, winv, K, c_out, nbr, nbr_stride, n_out, scale, shift, residual, relu, ys, st,
                          order, query);
}

}  // namespace isf
