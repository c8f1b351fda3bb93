// isf_window_block.hip -- A10/A11: the attention half of an SST encoder layer (sst_basic_block_v2.py:41-75, :104-116) on
// the dense BEV grid as ONE kernel on the matrix cores:
//     y = LayerNorm1( x + OutProj( WindowAttention( (x + pos) Wq, (x + pos) Wk, x Wv ) ) )
// Before: fused qkv linear (writes [M, 3d]: 199 MB at 180 x 180, B = 4) -> fp32 vector-pipe attention (36 of 64 lanes
// live) -> fused out-projection, three launches and four passes over HBM.  Here a wave owns one 6 x 6 window (36 tokens
// padded to three 16-row MFMA tiles) end to end; q, k, v, the scores and the attention output never leave registers:
//   per head:  Q^T = Wq X^T, K^T = Wk X^T (weights as the A operand, token fragments as B), V = X Wv   (16x16x32 f16x3)
//              S^T = K Q^T          -- the K^T / Q^T accumulator tiles ARE the A / B operand layouts (16x16x16)
//              softmax over the keys (registers + two cross-lane steps), invalid slots masked
//              O^T = V^T P^T        -- V's accumulator tile is the A layout, S^T's the B layout
//              Y  += O Wout_h^T     -- O^T's accumulator tile is the A layout
//   epilogue:  + bias + residual, LayerNorm across the 16 column lanes and the column tiles, row-major fp32 store.
// All products in the hi/lo f16 split arithmetic of the other kernels (fp32-class; three MFMAs per product).
// The position embedding enters as (x + pos) W = x W + (pos W): a [36, 2d] table added to the q / k accumulators.
// Workgroup = 4 waves = 4 windows; the weights of a head (32 KiB in consumption order, host-packed by
// isf_pack_window_block) are LDS-DMA'd once per workgroup into a double buffer and shared by the 4 waves.
// Built for d = 128 (head_dim 16), the 180 x 180 level: 0.45 -> 0.36 ms per two-layer block at B = 2, 0.82 -> 0.72 at
// B = 4 (with its feed-forward linears; profiles/r02_call12_window_block.txt).
#include "isf_spconv16.h"

namespace isf {

typedef _Float16 h4 __attribute__((ext_vector_type(4)));

static constexpr int kWin = 6, kSlots = 36, kTT = 3;      // 6 x 6 windows, 36 tokens = 3 tiles of 16 (12 padding rows)
static constexpr int kStageBytes = 32 * 1024;

template <int D, int HD>
struct WinCfg {
  static constexpr int HEADS = D / HD;
  static constexpr int HT = HD / 16;                      // 16-wide tiles per head
  static constexpr int KC = D / 32;                       // 32-channel chunks of the model dimension
  static constexpr int CT = D / 16;                       // 16-column tiles of the model dimension
  static constexpr int proj_bytes = HT * KC * 2048;       // one of Wq / Wk / Wv for a head: [u][kc][hi|lo][64][8 halves]
  static constexpr int out_bytes = HT * CT * 1024;        // Wout for a head: [u][ct][hi|lo][64][4 halves]
  // a head's weights {q, k, v, out} are ONE 32-KiB stage in consumption order.  (A d = 256 / head_dim 32 variant --
  // four stages per head, token fragments re-read per head, 474 registers -- was built and measured: 0.79 ms against
  // 0.42 ms for the three-launch form at S = 90, B = 2: too few windows, one wave per SIMD.  Removed; the wide level
  // stays on the unfused path.  profiles/r02_call12_window_block.txt)
  static_assert(3 * proj_bytes + out_bytes == kStageBytes, "one stage per head");
};

__device__ __forceinline__ void split4(const f32x4 v, h4& hi, h4& lo) {
  hi = __builtin_convertvector(v, h4);
  lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), h4);
}

__device__ __forceinline__ f32x4 mma16x3(const h4 ah, const h4 al, const h4 bh, const h4 bl, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bl, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 mma32x3(const h8 ah, const h8 al, const h8 bh, const h8 bl, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
}

// packed: [HEADS][32 KiB] then the header {1/scale_qkv, 1/scale_out}
template <int D, int HD>
__global__ __launch_bounds__(256, 1) void window_block_kernel(
    const float* __restrict__ x, int B, int S, int shift, const uint4* __restrict__ packed,
    const float* __restrict__ header, const float* __restrict__ bqkv, const float* __restrict__ table /* [36][3D] */,
    const float* __restrict__ bout, const float* __restrict__ ln_g, const float* __restrict__ ln_b, float eps,
    float* __restrict__ y, int num_windows) {
  using C = WinCfg<D, HD>;
  constexpr int HT = C::HT, KC = C::KC, CT = C::CT, HEADS = C::HEADS;
  extern __shared__ __attribute__((aligned(16))) char smem[];           // 2 x 32 KiB weight stages
  const unsigned smem_addr = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 15, kg = lane >> 4;

  // ---- this wave's window
  const int off = shift ? kWin / 2 : 0;
  const int nwin = shift ? (S - 1 + kWin / 2) / kWin + 1 : (S + kWin - 1) / kWin;
  const int wid = blockIdx.x * 4 + wave;
  const bool live = wid < num_windows;                                   // dead waves still stage weights and meet barriers
  const int wb = live ? wid / (nwin * nwin) : 0;
  const int wy = live ? (wid / nwin) % nwin : 0, wx = live ? wid % nwin : 0;
  const int y0 = wy * kWin - off, x0 = wx * kWin - off;
  // slot s (= in-window position, the row of the position table) -> token row of x / y, or -1
  auto token_of = [&](int s) -> long long {
    if (!live || s >= kSlots) return -1;
    const int yy = y0 + s / kWin, xx = x0 + s % kWin;
    if (yy < 0 || yy >= S || xx < 0 || xx >= S) return -1;
    return ((long long)wb * S + yy) * S + xx;
  };
  unsigned long long vmask = 0;                                          // valid slots, wave-uniform
  {
    const unsigned long long m = __ballot(token_of(lane) >= 0);
    vmask = m;
  }
  long long tok[kTT];                                                    // token of slot 16 j + col (this lane's row / column)
#pragma unroll
  for (int j = 0; j < kTT; ++j) tok[j] = token_of(16 * j + col);

  // token fragments: lane (slot 16 j + col, k-group kg) holds channels 32 kc + 8 kg .. + 7 as hi / lo f16 -- the A
  // operand of V = X Wv and, unchanged, the B operand of Q^T = Wq X^T
  auto load_x = [&](int j, int kc, h8& hi, h8& lo) {
    f32x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (tok[j] >= 0) v = *reinterpret_cast<const f32x8*>(x + tok[j] * D + 32 * kc + 8 * kg);
    hi = __builtin_convertvector(v, h8);
    lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x8), h8);
  };
  h8 xh[kTT][KC], xl[kTT][KC];                            // resident for the whole window: 96 registers at d = 128
#pragma unroll
  for (int j = 0; j < kTT; ++j)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) load_x(j, kc, xh[j][kc], xl[j][kc]);

  f32x4 yacc[kTT][CT];
#pragma unroll
  for (int j = 0; j < kTT; ++j)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) yacc[j][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- weight stages through a double-buffered LDS ring
  constexpr int NST = HEADS;
  auto stage_in = [&](int s) {                                            // 32 KiB = 32 LDS-DMA instructions of 1 KiB
    const uint4* src = packed + (size_t)s * (kStageBytes / 16);
    const unsigned dst = smem_addr + (unsigned)((s & 1) * kStageBytes);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int piece = (i * 4 + wave) * 64;
      glds16(src + piece + lane, dst + (unsigned)piece * 16u);
    }
  };
  const float inv_qkv = header[0], inv_out = header[1];
  static_assert(HD == 16, "1 / sqrt(head_dim) below");
  const float qscale = inv_qkv * 0.25f;                                   // 1 / sqrt(HD)
  stage_in(0);
  int st = 0;                                                             // stage being consumed
  auto next_stage = [&]() {                                               // -> LDS base of stage `st`, prefetches st + 1
    __builtin_amdgcn_s_waitcnt(0x0F70);                                   // vmcnt(0): my share of stage st has landed
    __syncthreads();                                                      // everyone's has; stage st - 1 is free
    if (st + 1 < NST) stage_in(st + 1);
    const char* base = smem + (st & 1) * kStageBytes;
    ++st;
    return base;
  };

  for (int head = 0; head < HEADS; ++head) {
    f32x4 qT[HT][kTT], kT[HT][kTT], vv[kTT][HT];                          // Q^T, K^T: [h tile][token tile]; V: [token tile][h tile]
    const char* wbase = next_stage();
    // ---- projections: which = 0 (Q^T), 1 (K^T), 2 (V)
#pragma unroll
    for (int which = 0; which < 3; ++which) {
      const uint4* w = reinterpret_cast<const uint4*>(wbase + which * C::proj_bytes) + lane;
#pragma unroll
      for (int u = 0; u < HT; ++u) {
        f32x4 acc[kTT];
#pragma unroll
        for (int j = 0; j < kTT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          const uint4 whu = w[((u * KC + kc) * 2 + 0) * 64], wlu = w[((u * KC + kc) * 2 + 1) * 64];
          const h8 wh = *reinterpret_cast<const h8*>(&whu), wl = *reinterpret_cast<const h8*>(&wlu);
#pragma unroll
          for (int j = 0; j < kTT; ++j) {
            const h8 ah = xh[j][kc], al = xl[j][kc];
            if (which == 2) acc[j] = mma32x3(ah, al, wh, wl, acc[j]);    // V = X Wv^T: tokens are rows
            else acc[j] = mma32x3(wh, wl, ah, al, acc[j]);               // Q^T / K^T = W X^T: tokens are columns
          }
        }
        // bias (+ position table for q, k); accumulator element t: row 4 kg + t, column col
#pragma unroll
        for (int j = 0; j < kTT; ++j) {
          if (which == 2) {      // rows = tokens, column = channel 16 u + col of the head
            const float bv = bqkv[2 * D + head * HD + 16 * u + col];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[j][t] = fmaf(acc[j][t], inv_qkv, bv);
            vv[j][u] = acc[j];
          } else {               // rows = channels 16 u + 4 kg + t of the head, column = slot 16 j + col
            const int c0 = which * D + head * HD + 16 * u + 4 * kg;
            f32x4 add = *reinterpret_cast<const f32x4*>(bqkv + c0);
            const int slot = 16 * j + col;
            if (slot < kSlots) add += *reinterpret_cast<const f32x4*>(table + (size_t)slot * (3 * D) + c0);
            const float sc = which == 0 ? qscale : inv_qkv;
            const float bs = which == 0 ? qscale / inv_qkv : 1.0f;       // the bias / table are scaled with q too
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[j][t] = fmaf(acc[j][t], sc, add[t] * bs);
            if (which == 0) qT[u][j] = acc[j]; else kT[u][j] = acc[j];
          }
        }
      }
    }
    // ---- S^T[a][b] = K_a Q_b^T over the head dimension (keys a = rows, queries b = columns)
    f32x4 sT[kTT][kTT];
    {
      h4 qh[HT][kTT], ql[HT][kTT];
#pragma unroll
      for (int u = 0; u < HT; ++u)
#pragma unroll
        for (int b = 0; b < kTT; ++b) split4(qT[u][b], qh[u][b], ql[u][b]);
#pragma unroll
      for (int a = 0; a < kTT; ++a) {
        h4 kh[HT], kl[HT];
#pragma unroll
        for (int u = 0; u < HT; ++u) split4(kT[u][a], kh[u], kl[u]);
#pragma unroll
        for (int b = 0; b < kTT; ++b) {
          f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int u = 0; u < HT; ++u) s = mma16x3(kh[u], kl[u], qh[u][b], ql[u][b], s);
          sT[a][b] = s;
        }
      }
    }
    // ---- softmax over the keys: element t of sT[a][b] is key 16 a + 4 kg + t for query column 16 b + col
#pragma unroll
    for (int b = 0; b < kTT; ++b) {
      float m = -INFINITY;
#pragma unroll
      for (int a = 0; a < kTT; ++a)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const bool ok = (vmask >> (16 * a + 4 * kg + t)) & 1ull;
          sT[a][b][t] = ok ? sT[a][b][t] : -INFINITY;
          m = fmaxf(m, sT[a][b][t]);
        }
      m = fmaxf(m, __shfl_xor(m, 16, 64));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      if (m == -INFINITY) m = 0.f;                                        // a dead wave: every key masked
      float sum = 0.f;
#pragma unroll
      for (int a = 0; a < kTT; ++a)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          sT[a][b][t] = __expf(sT[a][b][t] - m);                          // masked: exp(-inf) = 0
          sum += sT[a][b][t];
        }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float inv = sum > 0.f ? 1.f / sum : 0.f;
#pragma unroll
      for (int a = 0; a < kTT; ++a) sT[a][b] *= inv;
    }
    // ---- O^T[u][b] = V_u^T P_b^T over the keys; then Y += O Wout_head^T
    const uint2* wo = reinterpret_cast<const uint2*>(wbase + 3 * C::proj_bytes) + lane;
    h4 vh[kTT][HT], vl[kTT][HT];
#pragma unroll
    for (int a = 0; a < kTT; ++a)
#pragma unroll
      for (int u = 0; u < HT; ++u) split4(vv[a][u], vh[a][u], vl[a][u]);
#pragma unroll
    for (int b = 0; b < kTT; ++b) {
      h4 ph[kTT], pl[kTT];
#pragma unroll
      for (int a = 0; a < kTT; ++a) split4(sT[a][b], ph[a], pl[a]);
#pragma unroll
      for (int u = 0; u < HT; ++u) {
        f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < kTT; ++a) o = mma16x3(vh[a][u], vl[a][u], ph[a], pl[a], o);
        h4 oh, ol;
        split4(o, oh, ol);                                                // O^T's accumulator = O as an A operand
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const uint2 bhu = wo[((u * CT + ct) * 2 + 0) * 64], blu = wo[((u * CT + ct) * 2 + 1) * 64];
          yacc[b][ct] = mma16x3(oh, ol, *reinterpret_cast<const h4*>(&bhu), *reinterpret_cast<const h4*>(&blu), yacc[b][ct]);
        }
      }
    }
  }
  if (!live) return;

  // ---- epilogue: rows 16 j + 4 kg + t, column 16 ct + col: + bias + residual, LayerNorm over the D channels
  float bo[CT], g[CT], bt[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) { bo[ct] = bout[16 * ct + col]; g[ct] = ln_g[16 * ct + col]; bt[ct] = ln_b[16 * ct + col]; }
#pragma unroll
  for (int j = 0; j < kTT; ++j) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int slot = 16 * j + 4 * kg + t;
      const long long row = token_of(slot);                               // uniform over the 16 column lanes of this kg
      float v[CT];
      float sum = 0.f;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const float r = row >= 0 ? x[row * D + 16 * ct + col] : 0.f;
        v[ct] = fmaf(yacc[j][ct][t], inv_out, bo[ct]) + r;
        sum += v[ct];
      }
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) sum += __shfl_xor(sum, d, 64);
      const float mean = sum * (1.0f / D);
      float var = 0.f;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) { v[ct] -= mean; var = fmaf(v[ct], v[ct], var); }
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) var += __shfl_xor(var, d, 64);
      const float rstd = rsqrtf(var * (1.0f / D) + eps);
      if (row >= 0) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) y[row * D + 16 * ct + col] = fmaf(v[ct] * rstd, g[ct], bt[ct]);
      }
    }
  }
}

// weights -> the kernel's streaming order.  w_in [3D, D] (in_proj_weight), w_out [D, D]; scales 2^sq / 2^so keep the lo
// halves in f16's normal range (as isf_pack_linear).  One thread per 16-byte (projection) / 8-byte (out) lane piece.
template <int D, int HD>
__global__ void pack_window_block_kernel(const float* __restrict__ w_in, const float* __restrict__ w_out,
                                         const unsigned* __restrict__ amax_bits /* [2] */, char* __restrict__ packed) {
  using C = WinCfg<D, HD>;
  constexpr int HT = C::HT, KC = C::KC, CT = C::CT, HEADS = C::HEADS;
  int e = 0;
  float a = __uint_as_float(amax_bits[0]);
  if (a > 0.f) (void)frexpf(a, &e);
  const int sq = a > 0.f ? 13 - e : 0;
  a = __uint_as_float(amax_bits[1]);
  e = 0;
  if (a > 0.f) (void)frexpf(a, &e);
  const int so = a > 0.f ? 13 - e : 0;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  float* header = reinterpret_cast<float*>(packed + (size_t)HEADS * (3 * C::proj_bytes + C::out_bytes));
  if (t == 0) { header[0] = ldexpf(1.f, -sq); header[1] = ldexpf(1.f, -so); }
  constexpr long long proj_items = (long long)HEADS * 3 * HT * KC * 64;   // (head, which, u, kc, lane): hi + lo
  constexpr long long out_items = (long long)HEADS * HT * CT * 64;        // (head, u, ct, lane): hi + lo
  const size_t head_bytes = 3 * C::proj_bytes + C::out_bytes;
  if (t < proj_items) {
    const int ln = (int)(t & 63);
    long long r = t >> 6;
    const int kc = (int)(r % KC); r /= KC;
    const int u = (int)(r % HT); r /= HT;
    const int which = (int)(r % 3);
    const int head = (int)(r / 3);
    const int orow = which * D + head * HD + 16 * u + (ln & 15);
    f32x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = ldexpf(w_in[(size_t)orow * D + 32 * kc + 8 * (ln >> 4) + j], sq);
    const h8 hi = __builtin_convertvector(v, h8);
    const h8 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x8), h8);
    char* base = packed + head * head_bytes + which * C::proj_bytes + (size_t)((u * KC + kc) * 2) * 1024;
    reinterpret_cast<h8*>(base)[ln] = hi;
    reinterpret_cast<h8*>(base + 1024)[ln] = lo;
  } else if (t < proj_items + out_items) {
    long long r = t - proj_items;
    const int ln = (int)(r & 63); r >>= 6;
    const int ct = (int)(r % CT); r /= CT;
    const int u = (int)(r % HT);
    const int head = (int)(r / HT);
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      v[j] = ldexpf(w_out[(size_t)(16 * ct + (ln & 15)) * D + head * HD + 16 * u + 4 * (ln >> 4) + j], so);
    h4 hi, lo;
    split4(v, hi, lo);
    char* base = packed + head * head_bytes + 3 * C::proj_bytes + (size_t)((u * CT + ct) * 2) * 512;
    reinterpret_cast<h4*>(base)[ln] = hi;
    reinterpret_cast<h4*>(base + 512)[ln] = lo;
  }
}

__global__ void absmax2_kernel(const float* __restrict__ w0, size_t n0, const float* __restrict__ w1, size_t n1,
                               unsigned* __restrict__ out /* [2] */) {
  const float* w = blockIdx.y == 0 ? w0 : w1;
  const size_t n = blockIdx.y == 0 ? n0 : n1;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float m = 0.f;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out + blockIdx.y, __float_as_uint(m));
}

}  // namespace isf

extern "C" {

size_t isf_packed_window_block_bytes(int embed_dims) { return (size_t)embed_dims * embed_dims * 16 + 64; }

int isf_pack_window_block(const float* in_proj_weight, const float* out_proj_weight, int embed_dims, int num_heads,
                          void* packed, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(in_proj_weight && out_proj_weight && packed, ISF_ERR_ARG, "pack_window_block: null pointer");
  ISF_REQUIRE(num_heads == 8 && embed_dims == 128, ISF_ERR_UNSUPPORTED,
              "pack_window_block: built for 8 heads, d = 128 (got %d heads, d %d)", num_heads, embed_dims);
  hipStream_t st = as_stream(stream);
  Arena& a = arena_for_stream(st);
  ISF_TRY(a.reset());
  unsigned* amax = nullptr;
  ISF_TRY(a.alloc_n(&amax, 64));
  ISF_HIP_TRY(hipMemsetAsync(amax, 0, 2 * sizeof(unsigned), st));
  const size_t n_in = (size_t)3 * embed_dims * embed_dims, n_out = (size_t)embed_dims * embed_dims;
  hipLaunchKernelGGL(absmax2_kernel, dim3(64, 2), dim3(256), 0, st, in_proj_weight, n_in, out_proj_weight, n_out, amax);
  const long long items = (long long)8 * 3 * 1 * 4 * 64 + (long long)8 * 1 * 8 * 64;
  hipLaunchKernelGGL((pack_window_block_kernel<128, 16>), dim3(ceil_div(items, 256)), dim3(256), 0, st, in_proj_weight,
                     out_proj_weight, amax, reinterpret_cast<char*>(packed));
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int isf_window_block_forward(const float* x, int batch_size, int grid_size, int embed_dims, int num_heads, int window,
                             int shift, const void* packed, const float* in_proj_bias, const float* pos_table,
                             const float* out_proj_bias, const float* ln_gamma, const float* ln_beta, float ln_eps,
                             float* y, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(batch_size >= 0 && grid_size > 0, ISF_ERR_ARG, "window_block: bad sizes");
  if (batch_size == 0) return ISF_OK;
  ISF_REQUIRE(x && packed && in_proj_bias && pos_table && out_proj_bias && ln_gamma && ln_beta && y, ISF_ERR_ARG,
              "window_block: null pointer");
  ISF_REQUIRE(num_heads == 8 && window == 6 && embed_dims == 128, ISF_ERR_UNSUPPORTED,
              "window_block: built for 8 heads, 6x6 windows, d = 128 (got %d heads, win %d, d %d)", num_heads, window,
              embed_dims);
  const int nwin = shift ? (grid_size - 1 + window / 2) / window + 1 : (grid_size + window - 1) / window;
  const int num_windows = batch_size * nwin * nwin;
  const float* header = reinterpret_cast<const float*>(reinterpret_cast<const char*>(packed) +
                                                       (size_t)embed_dims * embed_dims * 16);
  const dim3 grid(ceil_div(num_windows, 4)), block(256);
  static bool attr_set = false;
  if (!attr_set) {
    ISF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&window_block_kernel<128, 16>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kStageBytes));
    attr_set = true;
  }
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL((window_block_kernel<128, 16>), grid, block, 2 * kStageBytes, st, x, batch_size, grid_size, shift,
                     reinterpret_cast<const uint4*>(packed), header, in_proj_bias, pos_table, out_proj_bias, ln_gamma,
                     ln_beta, ln_eps, y, num_windows);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

}  // extern "C"
