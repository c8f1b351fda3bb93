// isf_spconv_wgrad16.hip -- SURVEY 8f #2 (round 5): sparse-convolution FILTER gradient on the f16 matrix cores.
//
// Reference (spconv_ops.h:363-456, indice_conv_backward): per tap k, gather the x rows and the dY rows of the tap's pairs
// into two HBM buffers and run one GEMM  dW[k] = X_k^T dY_k  (torch::mm_out; half instantiation all.cc:35-51).
// The fp32-MFMA kernel of isf_spconv_bwd.hip (v_mfma_f32_16x16x4_f32, one k per lane so that no transposition is
// needed) runs at 16-31 TFLOP/s: 10-20 % of the fp32 matrix peak, 1 % of what the f16 pipe delivers.  Here:
//
//   * pair lists: the spconv-1 interchange format the reference itself uses (indice_pairs [K][2][cap], indice_num [K]),
//     built once per rulebook by an ordered two-pass compaction (pair_count / pair_write kernels) -- only rows that
//     contribute are ever touched (6-15 of 27 taps per row), and the work is cut by PAIRS, not by rows, so every
//     workgroup of a tap carries the same load.
//   * arithmetic: the f16x3 split of the forward kernels.  x arrives in the split format the forward pass already
//     stores (hi | lo halves, isf_common.h), dY is converted once per layer (isf_grad_to_split: scaled by a power of two
//     into f16's normal range -- gradients of 1e-6 .. 1e-9 would otherwise sit in its subnormals -- and split); a product
//     is x_lo*g_hi + x_hi*g_lo + x_hi*g_hi on v_mfma_f32_16x16x32_f16, fp32 accumulate: the error of an fp32 GEMM.
//   * the MFMA's reduction index is the PAIR (row) index, but memory is channel-contiguous: a lane must hold 8
//     consecutive rows of ONE channel.  Each wave stages its own 32-pair x (BM + BN)-channel tile through a private LDS
//     region: 16-byte row pieces come in quad-friendly (a lane quad reads 64 contiguous bytes of one row), four rows per
//     lane are transposed in registers (shift / mask pairs) into 8-byte (4 rows x 1 channel) ds_write_b64, channel-major with an
//     XOR swizzle of record and row-octet slot that makes both the ds_write_b64 of the staging and the ds_read_b128 of
//     the MFMA fragments bank-conflict-free (wg16_addr).
//     No workgroup barrier in the loop: the four waves of a workgroup work on four consecutive pair ranges and add their
//     accumulators through LDS at the end (fixed order: deterministic).
//   * pair chunks write partial blocks, a second pass adds them in chunk order and applies the gradient's inverse
//     scale (deterministic, no float atomics).
#include <algorithm>

#include "isf_spconv16.h"

namespace isf {

// ------------------------------------------------------------------------------------------------ pair lists
constexpr int kPairBlockRows = 2048;

__global__ __launch_bounds__(256) void pair_count_kernel(const int32_t* __restrict__ nbr, int nbr_stride, int n_out,
                                                         int nblk, int32_t* __restrict__ cnt /* [K][nblk] */) {
  const int b = blockIdx.x, k = blockIdx.y;
  const int32_t* nk = nbr + (size_t)k * nbr_stride;
  int c = 0;
  for (int r = b * kPairBlockRows + threadIdx.x; r < min(n_out, (b + 1) * kPairBlockRows); r += 256) c += nk[r] >= 0;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
  __shared__ int wsum[4];
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) cnt[k * nblk + b] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(256) void pair_write_kernel(const int32_t* __restrict__ nbr, int nbr_stride, int n_out,
                                                         int nblk, const int32_t* __restrict__ cnt, int cap,
                                                         int32_t* __restrict__ pairs, int32_t* __restrict__ num) {
  const int b = blockIdx.x, k = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int32_t* nk = nbr + (size_t)k * nbr_stride;
  int32_t* pin = pairs + ((size_t)k * 2 + 0) * cap;
  int32_t* pout = pairs + ((size_t)k * 2 + 1) * cap;
  __shared__ int wsum[4];
  __shared__ int base_s;
  int c = 0;
  for (int i = threadIdx.x; i < b; i += 256) c += cnt[k * nblk + i];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
  if (lane == 0) wsum[wave] = c;
  __syncthreads();
  if (threadIdx.x == 0) base_s = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  int base = base_s;
  const int r_end = min(n_out, (b + 1) * kPairBlockRows);
  for (int r0 = b * kPairBlockRows; r0 < r_end; r0 += 256) {
    const int r = r0 + threadIdx.x;
    const int v = r < r_end ? nk[r] : -1;
    const unsigned long long m = __ballot(v >= 0);
    __syncthreads();                                   // wsum is free again
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (v >= 0) {
      const int pos = off + __popcll(m & ((1ull << lane) - 1));
      if (pos < cap) { pin[pos] = v; pout[pos] = r; }
    }
    base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }
  if (b == nblk - 1) {                                 // the last block knows the tap's total: count + -1 padding
    const int total = min(base, cap);
    if (threadIdx.x == 0) num[k] = total;
    const int pad_end = min(cap, ((total + 31) & ~31) + 32);
    for (int s = total + threadIdx.x; s < pad_end; s += 256) { pin[s] = -1; pout[s] = -1; }
  }
}

// ------------------------------------------------------------------------------------------------ gradient -> split
__global__ __launch_bounds__(256) void grad_absmax_kernel(const float* __restrict__ g, size_t n, float* __restrict__ bmax) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float a = fabsf(g[i]);
    m = (a <= 3.0e38f) ? fmaxf(m, a) : m;             // non-finite entries do not set the scale (they pass through it)
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
  __shared__ float w[4];
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) bmax[blockIdx.x] = fmaxf(fmaxf(w[0], w[1]), fmaxf(w[2], w[3]));
}

// gs = split(g * s), s = 2^(10 - e) with max|g| = m * 2^e, m in [0.5, 1): the largest entry lands in [2^9, 2^10)
__global__ __launch_bounds__(256) void grad_split_kernel(const float* __restrict__ g, size_t n8, const float* __restrict__ bmax,
                                                         int nb, uint4* __restrict__ gs, float* __restrict__ scale_out) {
  __shared__ float w[4];
  float m = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) m = fmaxf(m, bmax[i]);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(w[0], w[1]), fmaxf(w[2], w[3]));
  int e = 0;
  if (m > 0.f) (void)frexpf(m, &e);
  const float s = m > 0.f ? ldexpf(1.f, min(10 - e, 126)) : 1.f;   // clamped: a denormal max |g| must not give s = inf (0 * inf = NaN)
  if (blockIdx.x == 0 && threadIdx.x == 0) { scale_out[0] = s; scale_out[1] = 1.f / s; }
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const f32x8 v = *reinterpret_cast<const f32x8*>(g + i * 8) * s;
  uint4 hi, lo;
  split8(v, hi, lo);
  const size_t o = (i >> 2) * 8 + (i & 3);
  gs[o] = hi;
  gs[o + 4] = lo;
}

// out = g * s (fp32 rows, not split): the fused linear's dX GEMM takes fp32 rows and splits them itself
__global__ __launch_bounds__(256) void grad_scale_kernel(const float* __restrict__ g, size_t n, const float* __restrict__ bmax,
                                                         int nb, float* __restrict__ out, float* __restrict__ scale_out) {
  __shared__ float w[4];
  float m = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) m = fmaxf(m, bmax[i]);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(w[0], w[1]), fmaxf(w[2], w[3]));
  int e = 0;
  if (m > 0.f) (void)frexpf(m, &e);
  const float s = m > 0.f ? ldexpf(1.f, min(10 - e, 126)) : 1.f;   // clamped: a denormal max |g| must not give s = inf (0 * inf = NaN)
  if (blockIdx.x == 0 && threadIdx.x == 0) { scale_out[0] = s; scale_out[1] = 1.f / s; }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = g[i] * s;
}

__global__ void split_to_f32_scaled_kernel(const uint4* __restrict__ xs, size_t n8, const float* __restrict__ mul,
                                           float* __restrict__ x) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const size_t o = (i >> 2) * 8 + (i & 3);
  *reinterpret_cast<f32x8*>(x + i * 8) = join8(xs[o], xs[o + 4]) * (mul ? *mul : 1.f);
}

// ------------------------------------------------------------------------------------------------ dW
// LDS image of one operand tile of a wave: [hi | lo][channel record][32 rows] halves = 64 bytes per channel and half.
// Channel c sits in record c ^ ((c >> 3) & 1), the 16-byte slot of its row octet kg at slot kg ^ g((c >> 2) & 3) ^
// h((c >> 4) & 3) with g(x) = -x & 3, h = 2-bit reversal.  Checked exhaustively on the host (tests/test_kernel_algebra.py):
// the four 16-lane groups a ds_read_b128 is served in ({0-3, 12-15, 20-27}, ...) touch 16 different 16-byte slots of a
// 256-byte bank row, and the 16 contiguous lanes a ds_write_b64 is served in touch 16 different 8-byte slots of 128 bytes.
__host__ __device__ __forceinline__ int wg16_g(int x) { return (-x) & 3; }
__host__ __device__ __forceinline__ int wg16_h(int x) { return ((x & 1) << 1) | ((x >> 1) & 1); }
__host__ __device__ __forceinline__ int wg16_addr(int c, int kg) {   // byte offset of (channel c, row octet kg) in an image
  return ((c ^ ((c >> 3) & 1)) << 6) + (((kg ^ wg16_g((c >> 2) & 3) ^ wg16_h((c >> 4) & 3)) & 3) << 4);
}

// Staging of one operand tile (32 pairs x BC channels, hi and lo halves) by one wave.  Everything lane-dependent that does
// not change from step to step is computed ONCE (Wg16Lane): the byte offset of the lane's 16-byte pieces inside a row and
// the two LDS base addresses its eight ds_write_b64 per item go to (immediate offsets from there).  The loads of a step
// are UNCONDITIONAL 32-bit-offset loads (an `if (row >= 0) load` form compiled to a branch and an s_waitcnt vmcnt(0) per
// load -- sixteen serialised memory round trips per step: 0.09 of the roofline, profiles/r05_wgrad16.txt); positions past
// the wave's pair range load a valid row (index clamped to 0 / the next wave's pairs) and are zeroed on the x side in the
// one step that can contain them.
template <int BC>
struct Wg16Lane {
  unsigned piece_off[BC / 32];   // byte offset of this lane's piece inside a row, per item (hi / lo)
  unsigned lds0[BC / 32];        // LDS byte offset (within the operand image) of records 0..3 / 4..7 of the lane's channel
  unsigned lds1[BC / 32];        // octet, per item
  bool swap;                     // odd channel octet: records 2d / 2d + 1 hold channels 2d + 1 / 2d (wg16_addr's record XOR)
};

template <int BC>
__device__ __forceinline__ Wg16Lane<BC> wg16_lane(int lane, int chunk0) {
  Wg16Lane<BC> L;
  const int cg = BC == 64 ? (lane & 7) : (lane & 3);
  const int q = BC == 64 ? ((lane >> 3) & 7) : ((lane >> 2) & 7);
  L.swap = (cg & 1) != 0;
#pragma unroll
  for (int it = 0; it < BC / 32; ++it) {
    const int hl = BC == 64 ? it : (lane >> 5);
    L.piece_off[it] = (unsigned)(((chunk0 + (cg >> 2)) * 8 + hl * 4 + (cg & 3)) * 16);
    const int base = hl * (BC * 64) + (q & 1) * 8;
    // channel 8 cg + e sits in record 8 cg + (e ^ swap); its slot depends on e >> 2 only: wg16_addr(8 cg + e, q >> 1) =
    // wg16_addr(8 cg + (e & 4), q >> 1) + 64 * ((e ^ swap) & 3)   (checked against wg16_addr in the host test)
    L.lds0[it] = (unsigned)(base + (wg16_addr(8 * cg, q >> 1) & ~0x1c0) + 0);
    L.lds1[it] = (unsigned)(base + (wg16_addr(8 * cg + 4, q >> 1) & ~0x1c0) + 256);
  }
  return L;
}

template <int BC, bool HALF = false>   // HALF: the hi halves only (BC = 64: item 0; BC = 32 keeps both half-waves busy)
__device__ __forceinline__ void wg16_load(const char* __restrict__ src, unsigned row_bytes, const Wg16Lane<BC>& L,
                                          const int (&idx)[4], uint4 (&r)[BC / 32][4]) {
#pragma unroll
  for (int it = 0; it < ((HALF && BC == 64) ? 1 : BC / 32); ++it)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      r[it][j] = *reinterpret_cast<const uint4*>(src + (__umul24((unsigned)idx[j], row_bytes) + L.piece_off[it]));
}

template <int BC, bool HALF = false>
__device__ __forceinline__ void wg16_store(char* img /* this operand's LDS image */, const Wg16Lane<BC>& L,
                                           const uint4 (&r)[BC / 32][4]) {
#pragma unroll
  for (int it = 0; it < ((HALF && BC == 64) ? 1 : BC / 32); ++it) {
    char* a0 = reinterpret_cast<char*>(__builtin_assume_aligned(img + L.lds0[it], 8));
    char* a1 = reinterpret_cast<char*>(__builtin_assume_aligned(img + L.lds1[it], 8));
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const unsigned r0 = d == 0 ? r[it][0].x : d == 1 ? r[it][0].y : d == 2 ? r[it][0].z : r[it][0].w;
      const unsigned r1 = d == 0 ? r[it][1].x : d == 1 ? r[it][1].y : d == 2 ? r[it][1].z : r[it][1].w;
      const unsigned r2 = d == 0 ? r[it][2].x : d == 1 ? r[it][2].y : d == 2 ? r[it][2].z : r[it][2].w;
      const unsigned r3 = d == 0 ? r[it][3].x : d == 1 ? r[it][3].y : d == 2 ? r[it][3].z : r[it][3].w;
      // channel 2 d of the lane's octet: the low halves of the four rows; 2 d + 1: the high halves
      // (v_perm_b32: selector bytes 0-3 pick from the second operand, 4-7 from the first)
      const uint2 ev = make_uint2(__builtin_amdgcn_perm(r1, r0, 0x05040100u), __builtin_amdgcn_perm(r3, r2, 0x05040100u));
      const uint2 od = make_uint2(__builtin_amdgcn_perm(r1, r0, 0x07060302u), __builtin_amdgcn_perm(r3, r2, 0x07060302u));
      const uint2 lo = L.swap ? od : ev, hi = L.swap ? ev : od;
      char* a = d < 2 ? a0 : a1;
      *reinterpret_cast<uint2*>(a + 64 * ((2 * d) & 3)) = lo;          // ds_write_b64, immediate offsets
      *reinterpret_cast<uint2*>(a + 64 * ((2 * d + 1) & 3)) = hi;
    }
  }
}

// WM x WN waves of a workgroup own WM x WN adjacent BM x BN blocks of dW[k] and walk the SAME pairs (the x slice of a
// block row is fetched by WN waves, the dY slice of a block column by WM: the second fetch hits the CU's L1), the
// remaining factor WK = 4 / (WM * WN) splits the workgroup's pair range; waves with the same block are added through LDS.
// Workgroup ids are dealt to the 8 XCDs round-robin by the dispatcher: id -> (XCD, slot), and the (tap, pair chunk) GROUP of
// a workgroup is chosen so that all blocks of a group run on ONE XCD, back to back -- the group's x / dY rows are read
// from HBM once and then served by that XCD's L2 (launch order put 2 of a group's 16 blocks on each XCD: every XCD
// streamed every row).
// HALF: single-pass f16 -- only the hi halves of x and dY are fetched, staged and multiplied (one MFMA per product): the
// arithmetic of the reference's indice_conv_backward<at::Half> (fp16 operands, fp32 accumulation), what its sparse convs
// run under autocast (functional.py:24,46 custom_fwd(cast_inputs=torch.half)).
template <int BM, int BN, int WM, int WN, bool HALF>
__global__ __launch_bounds__(256, 2) void wgrad16_kernel(const uint4* __restrict__ xs, int cin,
                                                         const uint4* __restrict__ gs, int cout,
                                                         const int32_t* __restrict__ pairs,
                                                         const int32_t* __restrict__ counts, int cap, int chunk_pairs,
                                                         int chunks, int K, float* __restrict__ partial) {
  constexpr int MT = BM / 16, NTL = BN / 16, WK = 4 / (WM * WN);
  constexpr int REGION = (BM + BN) * 128;               // bytes per wave: x image (hi, lo) then dY image
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int blocks = (cin / (BM * WM)) * (cout / (BN * WN));
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int group = (slot / blocks) * 8 + xcd;           // (tap, chunk) group: taps outer, chunks inner
  const int blk = slot % blocks;
  if (group >= K * chunks) return;
  const int k = group / chunks, chunk = group % chunks;
  const int cnt = counts[k];
  const int p_begin = chunk * chunk_pairs;
  if (p_begin >= cnt) return;                            // whole workgroup: nothing of this tap in this chunk
  const int co_blocks = cout / (BN * WN);
  const int wk = wave / (WM * WN), wm = (wave % (WM * WN)) / WN, wn = wave % WN;
  const int ci_base = ((blk / co_blocks) * WM + wm) * BM, co_base = ((blk % co_blocks) * WN + wn) * BN;
  const int wp = chunk_pairs / WK;                       // pairs per wave (a multiple of 32)
  const int my_begin = p_begin + wk * wp;
  const int my_end = min(min(my_begin + wp, p_begin + chunk_pairs), cnt);
  const int32_t* pin = pairs + ((size_t)k * 2 + 0) * cap;
  const int32_t* pout = pairs + ((size_t)k * 2 + 1) * cap;
  char* region = smem + wave * REGION;
  char* xr = region;
  char* gr = region + BM * 128;

  f32x4 acc[MT][NTL];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int qx = BM == 64 ? ((lane >> 3) & 7) : ((lane >> 2) & 7);   // this lane's row quad in the x tile / the dY tile
  const int qg = BN == 64 ? ((lane >> 3) & 7) : ((lane >> 2) & 7);
  const Wg16Lane<BM> LX = wg16_lane<BM>(lane, ci_base >> 5);
  const Wg16Lane<BN> LG = wg16_lane<BN>(lane, co_base >> 5);
  const char* xb = reinterpret_cast<const char*>(xs);
  const char* gb = reinterpret_cast<const char*>(gs);
  const unsigned x_row = (unsigned)cin * 4u, g_row = (unsigned)cout * 4u;   // bytes per split row
  // the four pair entries of this lane's row quad at step position p0: unconditional load from a clamped position (the
  // lists are readable up to cap), entries clamped to row 0 where the list holds its -1 padding
  auto fetch_idx = [&](const int32_t* list, int p0, int q, int (&idx)[4]) {
    const int p = min(p0 + 4 * q, cap - 4);              // p0 multiple of 32, cap of 32: 16-byte aligned, in bounds
    const int4 v = *reinterpret_cast<const int4*>(list + p);
    idx[0] = max(v.x, 0); idx[1] = max(v.y, 0); idx[2] = max(v.z, 0); idx[3] = max(v.w, 0);
  };
  const int m16 = lane & 15, kg = lane >> 4;             // fragment lane: channel m16 of a 16-channel tile, row octet kg

  if (my_begin < my_end) {
    int ix[4], ig[4], ix_n[4], ig_n[4];
    uint4 rx[BM / 32][4], rg[BN / 32][4];
    fetch_idx(pin, my_begin, qx, ix);
    fetch_idx(pout, my_begin, qg, ig);
    fetch_idx(pin, my_begin + 32, qx, ix_n);
    fetch_idx(pout, my_begin + 32, qg, ig_n);
    wg16_load<BM, HALF>(xb, x_row, LX, ix, rx);
    wg16_load<BN, HALF>(gb, g_row, LG, ig, rg);
    for (int p0 = my_begin; p0 < my_end; p0 += 32) {
      if (p0 + 32 > my_end) {                            // the one step with positions past the range: zero their x rows
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (p0 + 4 * qx + j >= my_end) {
#pragma unroll
            for (int it = 0; it < BM / 32; ++it) rx[it][j] = make_uint4(0, 0, 0, 0);
          }
        }
      }
      wg16_store<BM, HALF>(xr, LX, rx);                  // waits for this step's rows
      wg16_store<BN, HALF>(gr, LG, rg);
#pragma unroll
      for (int j = 0; j < 4; ++j) { ix[j] = ix_n[j]; ig[j] = ig_n[j]; }
      fetch_idx(pin, p0 + 64, qx, ix_n);                 // indices two steps ahead, rows one step ahead
      fetch_idx(pout, p0 + 64, qg, ig_n);
      wg16_load<BM, HALF>(xb, x_row, LX, ix, rx);
      wg16_load<BN, HALF>(gb, g_row, LG, ig, rg);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // wave-private images: LDS ops of a wave complete in order
      uint4 ah[MT], al[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        ah[mt] = *reinterpret_cast<const uint4*>(xr + wg16_addr(mt * 16 + m16, kg));
        al[mt] = make_uint4(0, 0, 0, 0);
        if (!HALF) al[mt] = *reinterpret_cast<const uint4*>(xr + BM * 64 + wg16_addr(mt * 16 + m16, kg));
      }
#pragma unroll
      for (int nt = 0; nt < NTL; ++nt) {
        const uint4 bhu = *reinterpret_cast<const uint4*>(gr + wg16_addr(nt * 16 + m16, kg));
        uint4 blu = make_uint4(0, 0, 0, 0);
        if (!HALF) blu = *reinterpret_cast<const uint4*>(gr + BN * 64 + wg16_addr(nt * 16 + m16, kg));
        const h8 bh = *reinterpret_cast<const h8*>(&bhu);
        const h8 bl = *reinterpret_cast<const h8*>(&blu);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const h8 xh = *reinterpret_cast<const h8*>(&ah[mt]);
          const h8 xl = *reinterpret_cast<const h8*>(&al[mt]);
          if (!HALF) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, bh, acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, bl, acc[mt][nt], 0, 0, 0);
          }
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, bh, acc[mt][nt], 0, 0, 0);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the fragment reads precede the next step's image writes
    }
  }
  // accumulators -> this wave's region as a [BM][BN] fp32 block (C layout: row 4 * (lane >> 4) + r, column lane & 15),
  // then the WK waves of a block are added in wave order and written as this chunk's partial
  float* blkp = reinterpret_cast<float*>(region);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) blkp[(mt * 16 + 4 * kg + r) * BN + nt * 16 + m16] = acc[mt][nt][r];
  __syncthreads();
  float* out = partial + ((size_t)chunk * K + k) * cin * cout;
  const int cb0 = (blk / co_blocks) * WM * BM, cn0 = (blk % co_blocks) * WN * BN;   // the workgroup's first channels
  for (int e = threadIdx.x; e < WM * WN * BM * BN / 4; e += 256) {
    const int b = e / (BM * BN / 4), o = e % (BM * BN / 4);   // block (wm, wn) = b, float4 o of it
    f32x4 sum = *reinterpret_cast<const f32x4*>(smem + b * REGION + o * 16);
#pragma unroll
    for (int w = 1; w < WK; ++w) sum += *reinterpret_cast<const f32x4*>(smem + (w * WM * WN + b) * REGION + o * 16);
    const int ci = cb0 + (b / WN) * BM + (o * 4) / BN, co = cn0 + (b % WN) * BN + (o * 4) % BN;
    *reinterpret_cast<f32x4*>(out + (size_t)ci * cout + co) = sum;
  }
}

__global__ void wgrad16_reduce_kernel(const float* __restrict__ partial, const int32_t* __restrict__ counts,
                                      int chunk_pairs, size_t tap_elems4 /* Cin * Cout / 4 */, int K,
                                      const float* __restrict__ inv_scale, f32x4* __restrict__ dw) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= tap_elems4 * K) return;
  const int k = (int)(e / tap_elems4);
  const int chunks = (counts[k] + chunk_pairs - 1) / chunk_pairs;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4* p = reinterpret_cast<const f32x4*>(partial);
  for (int c = 0; c < chunks; ++c) acc += p[(size_t)c * tap_elems4 * K + e];
  dw[e] = acc * (inv_scale ? *inv_scale : 1.f);
}

template <int BM, int BN, int WM, int WN, bool HALF>
static int launch_wgrad16(const uint4* xs, int cin, const uint4* gs, int cout, const int32_t* pairs, const int32_t* counts,
                          int cap, int chunk_pairs, int chunks, int K, float* partial, hipStream_t st) {
  constexpr int bytes = 4 * (BM + BN) * 128;
  auto kern = wgrad16_kernel<BM, BN, WM, WN, HALF>;
  if (bytes > 48 * 1024) {
    static bool once = false;
    if (!once) {
      ISF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      once = true;
    }
  }
  const int blocks = (cin / (BM * WM)) * (cout / (BN * WN));
  const int slots = ceil_div(K * chunks, 8) * blocks;    // per XCD: its groups, all blocks of a group back to back
  hipLaunchKernelGGL(kern, dim3(8 * slots), dim3(256), bytes, st, xs, cin, gs, cout, pairs, counts, cap, chunk_pairs, chunks,
                     K, partial);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

// workgroup shape per channel shape: blocks of 64 (32 for 32-channel sides), up to 2 x 2 blocks per workgroup
static inline void wgrad16_shape(int c_in, int c_out, int& BM, int& BN, int& WM, int& WN) {
  BM = c_in >= 64 ? 64 : 32;
  BN = c_out >= 64 ? 64 : 32;
  WM = c_in / BM >= 2 ? 2 : 1;
  WN = c_out / BN >= 2 ? 2 : 1;
}

}  // namespace isf

extern "C" {

int isf_pair_list_capacity(int num_in, int num_out) {
  const int m = num_in < num_out ? num_in : num_out;     // a tap pairs a row of either side at most once
  return ((m > 0 ? m : 1) + 31) / 32 * 32 + 32;
}

int isf_rulebook_pair_lists(const int32_t* nbr, int nbr_stride, int num_out, int num_taps, int capacity,
                            int32_t* indice_pairs, int32_t* indice_num, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(nbr && indice_pairs && indice_num && num_taps > 0 && num_out >= 0 && capacity >= 32 && capacity % 32 == 0 &&
                  nbr_stride >= num_out, ISF_ERR_ARG, "rulebook_pair_lists: bad arguments (capacity: isf_pair_list_capacity)");
  hipStream_t st = as_stream(stream);
  if (num_out == 0) {
    ISF_HIP_TRY(hipMemsetAsync(indice_num, 0, sizeof(int32_t) * num_taps, st));
    ISF_HIP_TRY(hipMemsetAsync(indice_pairs, 0xFF, sizeof(int32_t) * 2 * (size_t)num_taps * capacity, st));
    return ISF_OK;
  }
  const int nblk = ceil_div(num_out, kPairBlockRows);
  Arena& a = arena_for_stream(st);
  ISF_TRY(a.reset());
  int32_t* cnt = nullptr;
  ISF_TRY(a.alloc_n(&cnt, (size_t)num_taps * nblk));
  hipLaunchKernelGGL(pair_count_kernel, dim3(nblk, num_taps), dim3(256), 0, st, nbr, nbr_stride, num_out, nblk, cnt);
  hipLaunchKernelGGL(pair_write_kernel, dim3(nblk, num_taps), dim3(256), 0, st, nbr, nbr_stride, num_out, nblk, cnt, capacity,
                     indice_pairs, indice_num);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int isf_grad_to_split(const float* grad, size_t num_elems, void* grad_split, float* scale_out, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(scale_out && (num_elems == 0 || (grad && grad_split)), ISF_ERR_ARG, "grad_to_split: null pointer");
  ISF_REQUIRE(num_elems % 32 == 0, ISF_ERR_ARG, "grad_to_split: element count must be a multiple of 32");
  hipStream_t st = as_stream(stream);
  Arena& a = arena_for_stream(st);
  ISF_TRY(a.reset());
  const int nb = (int)std::min<size_t>(1024, std::max<size_t>(1, (num_elems + 4095) / 4096));
  float* bmax = nullptr;
  ISF_TRY(a.alloc_n(&bmax, 1024));
  hipLaunchKernelGGL(grad_absmax_kernel, dim3(nb), dim3(256), 0, st, grad, num_elems, bmax);
  const size_t n8 = num_elems / 8;
  hipLaunchKernelGGL(grad_split_kernel, dim3(std::max<size_t>(1, (n8 + 255) / 256)), dim3(256), 0, st, grad, n8, bmax, nb,
                     reinterpret_cast<uint4*>(grad_split), scale_out);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int isf_grad_rescale(const float* grad, size_t num_elems, float* out, float* scale_out, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(scale_out && (num_elems == 0 || (grad && out)), ISF_ERR_ARG, "grad_rescale: null pointer");
  hipStream_t st = as_stream(stream);
  Arena& a = arena_for_stream(st);
  ISF_TRY(a.reset());
  const int nb = (int)std::min<size_t>(1024, std::max<size_t>(1, (num_elems + 4095) / 4096));
  float* bmax = nullptr;
  ISF_TRY(a.alloc_n(&bmax, 1024));
  hipLaunchKernelGGL(grad_absmax_kernel, dim3(nb), dim3(256), 0, st, grad, num_elems, bmax);
  hipLaunchKernelGGL(grad_scale_kernel, dim3((unsigned)std::min<size_t>(4096, std::max<size_t>(1, (num_elems + 1023) / 1024))),
                     dim3(256), 0, st, grad, num_elems, bmax, nb, out, scale_out);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int isf_split_to_f32_scaled(const void* xs, size_t num_elems, const float* mul, float* x, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(num_elems == 0 || (x && xs), ISF_ERR_ARG, "split_to_f32_scaled: null pointer");
  ISF_REQUIRE(num_elems % 32 == 0, ISF_ERR_ARG, "split_to_f32_scaled: element count must be a multiple of 32");
  if (num_elems == 0) return ISF_OK;
  hipLaunchKernelGGL(split_to_f32_scaled_kernel, dim3(ceil_div((long long)(num_elems / 8), 256)), dim3(256), 0,
                     as_stream(stream), reinterpret_cast<const uint4*>(xs), num_elems / 8, mul, x);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int isf_sparse_conv_backward_filter_f16x3(const void* features_split, int num_in, int c_in, const void* grad_out_split,
                                          int num_out, int c_out, const int32_t* indice_pairs, const int32_t* indice_num,
                                          int capacity, int num_taps, const float* grad_inv_scale, float* grad_filters,
                                          int mode, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(num_in >= 0 && num_out >= 0 && num_taps > 0 && grad_filters && mode >= 0 && mode <= 3, ISF_ERR_ARG,
              "sparse_conv_backward_filter_f16x3: bad arguments (mode 0 = f16x3 split, 1 = single-pass f16, +2 = every tap "
              "list is full (dense grid): larger chunks)");
  const bool full_taps = (mode & 2) != 0;
  mode &= 1;
  ISF_REQUIRE(sparse_conv_f16x3_supported(c_in, c_out), ISF_ERR_UNSUPPORTED,
              "sparse_conv_backward_filter_f16x3: (Cin,Cout)=(%d,%d) not built (32 / 64 / 128 / 256)", c_in, c_out);
  hipStream_t st = as_stream(stream);
  const size_t elems = (size_t)num_taps * c_in * c_out;
  if (num_in == 0 || num_out == 0) {
    ISF_HIP_TRY(hipMemsetAsync(grad_filters, 0, sizeof(float) * elems, st));
    return ISF_OK;
  }
  ISF_REQUIRE(features_split && grad_out_split && indice_pairs && indice_num && capacity >= 32 && capacity % 32 == 0,
              ISF_ERR_ARG, "sparse_conv_backward_filter_f16x3: null pointer / capacity (isf_pair_list_capacity)");
  ISF_REQUIRE((unsigned long long)num_in * c_in * 4 < (1ull << 32) && (unsigned long long)num_out * c_out * 4 < (1ull << 32) &&
                  num_in < (1 << 24) && num_out < (1 << 24), ISF_ERR_UNSUPPORTED,
              "sparse_conv_backward_filter_f16x3: feature buffers of 4 GiB / 16 M rows and more are not addressed (32-bit "
              "row offsets)");
  int BM, BN, WM, WN;
  wgrad16_shape(c_in, c_out, BM, BN, WM, WN);
  const int blocks = (c_in / (BM * WM)) * (c_out / (BN * WN));
  // pairs per workgroup: ~3000 workgroups if every tap were full (the centre tap of a SubM layer is, the others hold
  // 0.2-0.6 of it), at least 256 pairs, at most what keeps the partial blocks under 256 MiB
  // (+2, a dense grid's rulebook -- dense_train.py: every tap holds `capacity` pairs, so the 3000 are real: 254 chunks of 256
  // pairs and 150 MB of partial blocks for a 128 -> 128 conv on 2 x 180 x 180 cells; three workgroups per CU, >= 512 pairs)
  long long chunk = ((long long)num_taps * capacity * blocks / (full_taps ? 768 : 3000) + 127) / 128 * 128;
  chunk = std::max<long long>(full_taps ? 512 : 256, std::min<long long>(chunk, 16384));
  while (((long long)capacity + chunk - 1) / chunk * (long long)elems * 4 > (256ll << 20) && chunk < (1 << 24)) chunk *= 2;
  const int chunk_pairs = (int)chunk, chunks = ceil_div(capacity, chunk_pairs);
  Arena& a = arena_for_stream(st);
  ISF_TRY(a.reset());
  float* partial = nullptr;
  ISF_TRY(a.alloc_n(&partial, elems * (size_t)chunks));
  const uint4* x = reinterpret_cast<const uint4*>(features_split);
  const uint4* g = reinterpret_cast<const uint4*>(grad_out_split);
  int rc = ISF_ERR_UNSUPPORTED;
#define ISF_WG16(bm, bn, wm, wn)                                                                                       \
  if (BM == bm && BN == bn && WM == wm && WN == wn)                                                                    \
    rc = mode == 1 ? launch_wgrad16<bm, bn, wm, wn, true>(x, c_in, g, c_out, indice_pairs, indice_num, capacity,       \
                                                          chunk_pairs, chunks, num_taps, partial, st)                  \
                   : launch_wgrad16<bm, bn, wm, wn, false>(x, c_in, g, c_out, indice_pairs, indice_num, capacity,      \
                                                           chunk_pairs, chunks, num_taps, partial, st)
  ISF_WG16(64, 64, 2, 2); ISF_WG16(64, 64, 2, 1); ISF_WG16(64, 64, 1, 2); ISF_WG16(64, 64, 1, 1);
  ISF_WG16(64, 32, 2, 1); ISF_WG16(64, 32, 1, 1); ISF_WG16(32, 64, 1, 2); ISF_WG16(32, 64, 1, 1); ISF_WG16(32, 32, 1, 1);
#undef ISF_WG16
  ISF_TRY(rc);
  const size_t tap4 = (size_t)c_in * c_out / 4;
  hipLaunchKernelGGL(wgrad16_reduce_kernel, dim3(ceil_div((long long)(tap4 * num_taps), 256)), dim3(256), 0, st, partial,
                     indice_num, chunk_pairs, tap4, num_taps, grad_inv_scale, reinterpret_cast<f32x4*>(grad_filters));
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

}  // extern "C"
