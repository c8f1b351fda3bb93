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
  const float s = m > 0.f ? ldexpf(1.f, 10 - e) : 1.f;
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

template <int BC>   // stage one operand tile: 32 pairs x BC channels (hi and lo) of the rows idx[0..3] (this lane's quad)
__device__ __forceinline__ void wg16_load(const uint4* __restrict__ src, int row_u4 /* uint4 per row */, int chunk0,
                                          const int (&idx)[4], int lane, uint4 (&r)[BC / 32][4]) {
  const int cg = BC == 64 ? (lane & 7) : (lane & 3);
#pragma unroll
  for (int it = 0; it < BC / 32; ++it) {
    const int hl = BC == 64 ? it : (lane >> 5);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      r[it][j] = make_uint4(0, 0, 0, 0);
      if (idx[j] >= 0) r[it][j] = src[(size_t)idx[j] * row_u4 + (chunk0 + (cg >> 2)) * 8 + hl * 4 + (cg & 3)];
    }
  }
}

template <int BC>
__device__ __forceinline__ void wg16_store(char* region /* this operand's LDS image */, int lane,
                                           const uint4 (&r)[BC / 32][4]) {
  const int cg = BC == 64 ? (lane & 7) : (lane & 3);
  const int q = BC == 64 ? ((lane >> 3) & 7) : ((lane >> 2) & 7);
#pragma unroll
  for (int it = 0; it < BC / 32; ++it) {
    const int hl = BC == 64 ? it : (lane >> 5);
    char* base = region + hl * (BC * 64) + (q & 1) * 8;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const unsigned r0 = d == 0 ? r[it][0].x : d == 1 ? r[it][0].y : d == 2 ? r[it][0].z : r[it][0].w;
      const unsigned r1 = d == 0 ? r[it][1].x : d == 1 ? r[it][1].y : d == 2 ? r[it][1].z : r[it][1].w;
      const unsigned r2 = d == 0 ? r[it][2].x : d == 1 ? r[it][2].y : d == 2 ? r[it][2].z : r[it][2].w;
      const unsigned r3 = d == 0 ? r[it][3].x : d == 1 ? r[it][3].y : d == 2 ? r[it][3].z : r[it][3].w;
      // channel 8 cg + 2 d: the low halves of the four rows; 8 cg + 2 d + 1: the high halves
      const uint2 ev = make_uint2((r0 & 0xffffu) | (r1 << 16), (r2 & 0xffffu) | (r3 << 16));
      const uint2 od = make_uint2((r0 >> 16) | (r1 & 0xffff0000u), (r2 >> 16) | (r3 & 0xffff0000u));
      const int c = 8 * cg + 2 * d;
      *reinterpret_cast<uint2*>(__builtin_assume_aligned(base + wg16_addr(c, q >> 1), 8)) = ev;        // ds_write_b64
      *reinterpret_cast<uint2*>(__builtin_assume_aligned(base + wg16_addr(c + 1, q >> 1), 8)) = od;
    }
  }
}

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void wgrad16_kernel(const uint4* __restrict__ xs, int cin,
                                                         const uint4* __restrict__ gs, int cout,
                                                         const int32_t* __restrict__ pairs,
                                                         const int32_t* __restrict__ counts, int cap, int chunk_pairs,
                                                         int K, float* __restrict__ partial) {
  constexpr int MT = BM / 16, NTL = BN / 16;
  constexpr int REGION = (BM + BN) * 128;               // bytes per wave: x image (hi, lo) then dY image
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k = blockIdx.y;
  const int cnt = counts[k];
  const int p_begin = blockIdx.z * chunk_pairs;
  if (p_begin >= cnt) return;                            // whole workgroup: nothing of this tap in this chunk
  const int co_blocks = cout / BN;
  const int ci_base = (blockIdx.x / co_blocks) * BM, co_base = (blockIdx.x % co_blocks) * BN;
  const int wp = chunk_pairs >> 2;                       // pairs per wave (a multiple of 32)
  const int my_begin = p_begin + wave * wp;
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
  auto fetch_idx = [&](const int32_t* list, int p0, int q, int (&idx)[4]) {
    const int p = p0 + 4 * q;                            // p0 is a multiple of 32, cap of 32: 16-byte aligned, in bounds
    int4 v = make_int4(-1, -1, -1, -1);
    if (p < my_end) v = *reinterpret_cast<const int4*>(list + p);
    idx[0] = v.x;
    idx[1] = p + 1 < my_end ? v.y : -1;
    idx[2] = p + 2 < my_end ? v.z : -1;
    idx[3] = p + 3 < my_end ? v.w : -1;
  };
  // fragment read offsets: lane (m = lane & 15, kg = lane >> 4) -> channel m of a 16-channel tile, row octet kg
  const int m16 = lane & 15, kg = lane >> 4;

  if (my_begin < my_end) {
    int ix[4], ig[4], ix_n[4], ig_n[4];
    uint4 rx[BM / 32][4], rg[BN / 32][4];
    fetch_idx(pin, my_begin, qx, ix);
    fetch_idx(pout, my_begin, qg, ig);
    fetch_idx(pin, my_begin + 32, qx, ix_n);
    fetch_idx(pout, my_begin + 32, qg, ig_n);
    wg16_load<BM>(xs, cin >> 2, ci_base >> 5, ix, lane, rx);
    wg16_load<BN>(gs, cout >> 2, co_base >> 5, ig, lane, rg);
    for (int p0 = my_begin; p0 < my_end; p0 += 32) {
      wg16_store<BM>(xr, lane, rx);                      // waits for this step's rows
      wg16_store<BN>(gr, lane, rg);
#pragma unroll
      for (int j = 0; j < 4; ++j) { ix[j] = ix_n[j]; ig[j] = ig_n[j]; }
      fetch_idx(pin, p0 + 64, qx, ix_n);                 // indices two steps ahead, rows one step ahead
      fetch_idx(pout, p0 + 64, qg, ig_n);
      wg16_load<BM>(xs, cin >> 2, ci_base >> 5, ix, lane, rx);
      wg16_load<BN>(gs, cout >> 2, co_base >> 5, ig, lane, rg);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // wave-private images: LDS ops of a wave complete in order
      uint4 ah[MT], al[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        ah[mt] = *reinterpret_cast<const uint4*>(xr + wg16_addr(mt * 16 + m16, kg));
        al[mt] = *reinterpret_cast<const uint4*>(xr + BM * 64 + wg16_addr(mt * 16 + m16, kg));
      }
#pragma unroll
      for (int nt = 0; nt < NTL; ++nt) {
        const uint4 bhu = *reinterpret_cast<const uint4*>(gr + wg16_addr(nt * 16 + m16, kg));
        const uint4 blu = *reinterpret_cast<const uint4*>(gr + BN * 64 + wg16_addr(nt * 16 + m16, kg));
        const h8 bh = *reinterpret_cast<const h8*>(&bhu);
        const h8 bl = *reinterpret_cast<const h8*>(&blu);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const h8 xh = *reinterpret_cast<const h8*>(&ah[mt]);
          const h8 xl = *reinterpret_cast<const h8*>(&al[mt]);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, bh, acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, bl, acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, bh, acc[mt][nt], 0, 0, 0);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the fragment reads precede the next step's image writes
    }
  }
  // accumulators -> this wave's region as a [BM][BN] fp32 block (C layout: row 4 * (lane >> 4) + r, column lane & 15),
  // then the four waves' blocks are added in wave order and written as this chunk's partial
  float* blk = reinterpret_cast<float*>(region);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) blk[(mt * 16 + 4 * kg + r) * BN + nt * 16 + m16] = acc[mt][nt][r];
  __syncthreads();
  float* out = partial + ((size_t)blockIdx.z * K + k) * cin * cout;
  for (int e = threadIdx.x; e < BM * BN / 4; e += 256) {
    f32x4 s = *reinterpret_cast<const f32x4*>(smem + 0 * REGION + e * 16);
    s += *reinterpret_cast<const f32x4*>(smem + 1 * REGION + e * 16);
    s += *reinterpret_cast<const f32x4*>(smem + 2 * REGION + e * 16);
    s += *reinterpret_cast<const f32x4*>(smem + 3 * REGION + e * 16);
    const int ci = (e * 4) / BN, co = (e * 4) % BN;
    *reinterpret_cast<f32x4*>(out + (size_t)(ci_base + ci) * cout + co_base + co) = s;
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

template <int BM, int BN>
static int launch_wgrad16(const uint4* xs, int cin, const uint4* gs, int cout, const int32_t* pairs, const int32_t* counts,
                          int cap, int chunk_pairs, int chunks, int K, float* partial, hipStream_t st) {
  constexpr int bytes = 4 * (BM + BN) * 128;
  auto kern = wgrad16_kernel<BM, BN>;
  if (bytes > 48 * 1024) {
    static bool once = false;
    if (!once) {
      ISF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      once = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3((cin / BM) * (cout / BN), K, chunks), dim3(256), bytes, st, xs, cin, gs, cout, pairs, counts,
                     cap, chunk_pairs, K, partial);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
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
                                          isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(num_in >= 0 && num_out >= 0 && num_taps > 0 && grad_filters, ISF_ERR_ARG,
              "sparse_conv_backward_filter_f16x3: bad arguments");
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
  const int BM = c_in >= 64 ? 64 : 32, BN = c_out >= 64 ? 64 : 32;
  const int blocks = (c_in / BM) * (c_out / BN);
  // pairs per workgroup: ~3000 workgroups if every tap were full (the centre tap of a SubM layer is, the others hold
  // 0.2-0.6 of it), at least 256 pairs (64 per wave), at most what keeps the partial blocks under 256 MiB
  long long chunk = ((long long)num_taps * capacity * blocks / 3000 + 127) / 128 * 128;
  chunk = std::max<long long>(256, std::min<long long>(chunk, 16384));
  while (((long long)capacity + chunk - 1) / chunk * (long long)elems * 4 > (256ll << 20) && chunk < (1 << 24)) chunk *= 2;
  const int chunk_pairs = (int)chunk, chunks = ceil_div(capacity, chunk_pairs);
  Arena& a = arena_for_stream(st);
  ISF_TRY(a.reset());
  float* partial = nullptr;
  ISF_TRY(a.alloc_n(&partial, elems * (size_t)chunks));
  const uint4* x = reinterpret_cast<const uint4*>(features_split);
  const uint4* g = reinterpret_cast<const uint4*>(grad_out_split);
  int rc;
  if (BM == 64 && BN == 64) rc = launch_wgrad16<64, 64>(x, c_in, g, c_out, indice_pairs, indice_num, capacity, chunk_pairs, chunks, num_taps, partial, st);
  else if (BM == 64) rc = launch_wgrad16<64, 32>(x, c_in, g, c_out, indice_pairs, indice_num, capacity, chunk_pairs, chunks, num_taps, partial, st);
  else if (BN == 64) rc = launch_wgrad16<32, 64>(x, c_in, g, c_out, indice_pairs, indice_num, capacity, chunk_pairs, chunks, num_taps, partial, st);
  else rc = launch_wgrad16<32, 32>(x, c_in, g, c_out, indice_pairs, indice_num, capacity, chunk_pairs, chunks, num_taps, partial, st);
  ISF_TRY(rc);
  const size_t tap4 = (size_t)c_in * c_out / 4;
  hipLaunchKernelGGL(wgrad16_reduce_kernel, dim3(ceil_div((long long)(tap4 * num_taps), 256)), dim3(256), 0, st, partial,
                     indice_num, chunk_pairs, tap4, num_taps, grad_inv_scale, reinterpret_cast<f32x4*>(grad_filters));
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

}  // extern "C"
