// isf_attention.hip -- the three softmax-attention cores of HSF Grid-to-Region (A10/A11) and IGF (A13/A14).
//
//  window_attention   36-token (win x win) windows of a DENSE S x S token grid; membership is arithmetic
//                     (reference: get_window_coors + flat2window/window2flat gathers, sst_ops.py:219-268,
//                     sst_basic_block_v2.py:41-75).  One wave per (window, head) on the matrix cores.
//  small_key_attention  Lk <= 256 keys resident in LDS, one thread per query, online softmax: the 200 x 200
//                     instance self-attention (fusion_encoder.py:664) and the 32400 x 200 instance-to-scene
//                     cross attention (fusion_encoder.py:489-494).
//  (the per-channel map attention of A14 lives in isf_channel_attn.hip)
// head_dim 16 (every use on the IS-Fusion path) runs attention_mfma16_kernel on the matrix cores, the window attention of
// both levels window_attention_mfma_kernel; the fp32 VALU kernels serve head_dim 32 of the generic entry point.
#include "isf_common.h"

namespace isf {

// ----------------------------------------------------------------------------------------------------------------
// window attention: qkv [B*S*S, 3d] (q | k | v, head h at columns h*HD); token row = (b*S + y)*S + x.
// window (wy, wx) covers y in [wy*win - off, +win), off = win (shift 0: aligned) or win/2 (shift 1) minus the
// first window index; out [B*S*S, d].  Kernel: window_attention_mfma_kernel below (the one-lane-per-token VALU kernel of
// rounds 1-3 is gone: 52 us per d = 256 launch at B = 2).
// ----------------------------------------------------------------------------------------------------------------
// q [B*Lq, ldq], k/v [B*Lk, ldk] (column offsets applied by the host), head h at columns h*HD; out [B*Lq, ldo]
template <int HD>
__global__ __launch_bounds__(256) void small_key_attention_kernel(const float* __restrict__ q, int ldq,
                                                                  const float* __restrict__ k,
                                                                  const float* __restrict__ v, int ldk, int Lq,
                                                                  int Lk, float scale, float* __restrict__ out,
                                                                  int ldo) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ks = smem;               // [Lk][HD]
  float* vs = smem + (size_t)Lk * HD;
  const int head = blockIdx.y, b = blockIdx.z;
  for (int i = threadIdx.x; i < Lk * (HD / 4); i += blockDim.x) {
    const int j = i / (HD / 4), c = (i % (HD / 4)) * 4;
    const size_t off = ((size_t)b * Lk + j) * ldk + head * HD + c;
    *reinterpret_cast<float4*>(ks + j * HD + c) = *reinterpret_cast<const float4*>(k + off);
    *reinterpret_cast<float4*>(vs + j * HD + c) = *reinterpret_cast<const float4*>(v + off);
  }
  __syncthreads();
  const int qi = blockIdx.x * blockDim.x + threadIdx.x;
  if (qi >= Lq) return;
  float qr[HD], o[HD];
  const float* qp = q + ((size_t)b * Lq + qi) * ldq + head * HD;
#pragma unroll
  for (int c = 0; c < HD; c += 4) {
    const float4 t = *reinterpret_cast<const float4*>(qp + c);
    qr[c] = t.x * scale; qr[c + 1] = t.y * scale; qr[c + 2] = t.z * scale; qr[c + 3] = t.w * scale;
    o[c] = o[c + 1] = o[c + 2] = o[c + 3] = 0.f;
  }
  float m = -INFINITY, sum = 0.f;
  for (int j = 0; j < Lk; ++j) {
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const float4 kk = *reinterpret_cast<const float4*>(ks + j * HD + c);
      a = fmaf(qr[c], kk.x, a); a = fmaf(qr[c + 1], kk.y, a); a = fmaf(qr[c + 2], kk.z, a); a = fmaf(qr[c + 3], kk.w, a);
    }
    const float mn = fmaxf(m, a);
    const float corr = __expf(m - mn);   // first step: exp(-inf) = 0
    const float p = __expf(a - mn);
    sum = sum * corr + p;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const float4 vv = *reinterpret_cast<const float4*>(vs + j * HD + c);
      o[c] = fmaf(p, vv.x, o[c] * corr); o[c + 1] = fmaf(p, vv.y, o[c + 1] * corr);
      o[c + 2] = fmaf(p, vv.z, o[c + 2] * corr); o[c + 3] = fmaf(p, vv.w, o[c + 3] * corr);
    }
    m = mn;
  }
  const float inv = 1.f / sum;
  float* op = out + ((size_t)b * Lq + qi) * ldo + head * HD;
#pragma unroll
  for (int c = 0; c < HD; c += 4)
    *reinterpret_cast<float4*>(op + c) = make_float4(o[c] * inv, o[c + 1] * inv, o[c + 2] * inv, o[c + 3] * inv);
}

// ----------------------------------------------------------------------------------------------------------------
// Many keys, few queries (TransFusion head: 200 queries x 32400 BEV keys, transfusion_head_v2.py:104-106): the keys
// are split into chunks of <= 512 (one workgroup per (query tile, head, batch x chunk), chunk resident in LDS); each
// thread runs the online softmax of its query over the chunk and writes (max, sum, unnormalised output); a second
// kernel merges the chunks.  part: [B][heads][nsplit][Lq][HD + 2].
template <int HD>
__global__ __launch_bounds__(256) void split_key_attention_kernel(const float* __restrict__ q, int ldq,
                                                                  const float* __restrict__ k,
                                                                  const float* __restrict__ v, int ldk, int Lq,
                                                                  int Lk, int kps, int nsplit, float scale,
                                                                  float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int head = blockIdx.y, b = blockIdx.z / nsplit, sp = blockIdx.z % nsplit;
  const int k0 = sp * kps;
  const int kn = min(kps, Lk - k0);
  float* ks = smem;
  float* vs = smem + (size_t)kps * HD;
  for (int i = threadIdx.x; i < kn * (HD / 4); i += blockDim.x) {
    const int j = i / (HD / 4), c = (i % (HD / 4)) * 4;
    const size_t off = ((size_t)b * Lk + k0 + j) * ldk + head * HD + c;
    *reinterpret_cast<float4*>(ks + j * HD + c) = *reinterpret_cast<const float4*>(k + off);
    *reinterpret_cast<float4*>(vs + j * HD + c) = *reinterpret_cast<const float4*>(v + off);
  }
  __syncthreads();
  const int qi = blockIdx.x * blockDim.x + threadIdx.x;
  if (qi >= Lq) return;
  float qr[HD], o[HD];
  const float* qp = q + ((size_t)b * Lq + qi) * ldq + head * HD;
#pragma unroll
  for (int c = 0; c < HD; c += 4) {
    const float4 t = *reinterpret_cast<const float4*>(qp + c);
    qr[c] = t.x * scale; qr[c + 1] = t.y * scale; qr[c + 2] = t.z * scale; qr[c + 3] = t.w * scale;
    o[c] = o[c + 1] = o[c + 2] = o[c + 3] = 0.f;
  }
  float m = -INFINITY, sum = 0.f;
  for (int j = 0; j < kn; ++j) {
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const float4 kk = *reinterpret_cast<const float4*>(ks + j * HD + c);
      a = fmaf(qr[c], kk.x, a); a = fmaf(qr[c + 1], kk.y, a); a = fmaf(qr[c + 2], kk.z, a); a = fmaf(qr[c + 3], kk.w, a);
    }
    const float mn = fmaxf(m, a);
    const float corr = __expf(m - mn);
    const float p = __expf(a - mn);
    sum = sum * corr + p;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const float4 vv = *reinterpret_cast<const float4*>(vs + j * HD + c);
      o[c] = fmaf(p, vv.x, o[c] * corr); o[c + 1] = fmaf(p, vv.y, o[c + 1] * corr);
      o[c + 2] = fmaf(p, vv.z, o[c + 2] * corr); o[c + 3] = fmaf(p, vv.w, o[c + 3] * corr);
    }
    m = mn;
  }
  float* pp = part + ((((size_t)b * gridDim.y + head) * nsplit + sp) * Lq + qi) * (HD + 2);
  pp[0] = m;
  pp[1] = sum;
#pragma unroll
  for (int c = 0; c < HD; ++c) pp[2 + c] = o[c];
}

// ----------------------------------------------------------------------------------------------------------------
// head_dim 16 on the matrix cores (both shapes above: few keys / many queries, and key splits of many keys).  A wave
// owns a 16-query tile and walks the workgroup's keys (LDS resident) 16 at a time, flash style:
//     S^T = K Q^T   (v_mfma_f32_16x16x16_f16, f16x3 split: lane (q = lane & 15, g = lane >> 4) gets the scores of
//                    query q against keys 4g .. 4g+3 of the tile)
//     online softmax: tile maximum / sum of a query = its 4 registers + two cross-lane steps (lanes q, q+16, q+32, q+48)
//     O^T += V^T P^T (the score registers ARE the B operand; V^T comes transposed out of LDS; the accumulator of lane
//                    (q, g) holds O[q][4g .. 4g+3]: one 16-byte store per lane)
// The VALU kernels spent 157 us on the 32400 x 200 instance-to-scene attention and 223 us on the head's 200 x 32400
// cross attention; their head_dim-32 instantiations remain for other widths.
typedef _Float16 ah4 __attribute__((ext_vector_type(4)));
typedef float af4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void attn_split4(const af4 v, ah4& hi, ah4& lo) {
  hi = __builtin_convertvector(v, ah4);
  lo = __builtin_convertvector(v - __builtin_convertvector(hi, af4), ah4);
}

__device__ __forceinline__ af4 attn_mma3(const ah4 ah, const ah4 al, const ah4 bh, const ah4 bl, af4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bl, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, c, 0, 0, 0);
}

static constexpr int kAttnLs = 20;           // LDS row stride in floats: 16-byte reads of 16 consecutive rows hit 64 banks once
static constexpr float kAttnNegBig = -3.0e38f;   // finite "-inf": exp(kAttnNegBig - m) == 0 without inf - inf

// grid (ceil(Lq / 256), heads, B * nsplit); workgroup = 4 waves x 4 query tiles; keys [k0, k0 + kn) of split sp.
// PARTIAL: write (max, sum, unnormalised O) for merge_key_splits_kernel instead of the normalised output.
// ----------------------------------------------------------------------------------------------------------------
// Window attention on the matrix cores (both head dims of the path: 16 at d = 128, 32 at d = 256).  One WAVE per
// (window, head), no LDS, no workgroup barrier: the 36 tokens of a 6 x 6 window are three 16-row MFMA tiles (12 padding
// rows masked);  S^T = K Q^T (v_mfma_f32_16x16x16_f16 in the f16x3 split, HD / 16 k-steps: K = head_dim exactly for
// head_dim 16, two steps for 32), the 3 x 3 score tiles of the window stay in registers, softmax of a query = its
// registers + two cross-lane steps, O^T = V^T P^T with the score registers as the B operand and V^T read transposed
// straight from the qkv rows (16 lanes = 64 contiguous bytes of one token).  Lane (q = lane & 15, g = lane >> 4) ends with
// O[q][4g .. 4g + 3] of each 16-wide slice of the head: one 16-byte store per slice.  Replaces the VALU kernel above for
// the d = 256 level (52 us per launch at B = 2: one lane per token, 36 x 32 FMAs twice) -- "windowed BEV attention on
// MFMA" for both levels; the d = 128 level's inference path is the fused window block (isf_window_block.hip).
// Reference: sst_basic_block_v2.py:41-75 (nn.MultiheadAttention over the padded window batch).
template <int HD>
__global__ __launch_bounds__(256) void window_attention_mfma_kernel(const float* __restrict__ qkv, int S, int d, int nwin,
                                                                   int y_off, int total_waves, float scale,
                                                                   float* __restrict__ out) {
  constexpr int WIN = 6, T = 36, NT = 3, KS = HD / 16;
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= total_waves) return;                    // wave-uniform
  const int head = wid & 7;
  int w = wid >> 3;
  const int wx = w % nwin;
  w /= nwin;
  const int wy = w % nwin, b = w / nwin;
  const int y0 = wy * WIN - y_off, x0 = wx * WIN - y_off;
  const int col = lane & 15, g = lane >> 4;
  // token of tile row (tile, r): index tile * 16 + r inside the window; -1 when padding or outside the grid
  auto token_row = [&](int j) -> long long {
    if (j >= T) return -1;
    const int y = y0 + j / WIN, x = x0 + j % WIN;
    if (y < 0 || y >= S || x < 0 || x >= S) return -1;
    return ((long long)b * S + y) * S + x;
  };
  long long row_c[NT];                               // the token this lane's column index names, per tile
#pragma unroll
  for (int t = 0; t < NT; ++t) row_c[t] = token_row(t * 16 + col);
  // Q (B operand) and K (A operand) fragments: [tile][k-step], element (row = col, dims 16 ks + 4 g ..)
  ah4 qh[NT][KS], ql[NT][KS], kh[NT][KS], kl[NT][KS];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      af4 qv = {0.f, 0.f, 0.f, 0.f}, kv = {0.f, 0.f, 0.f, 0.f};
      if (row_c[t] >= 0) {
        const float* base = qkv + (size_t)row_c[t] * (size_t)(3 * d) + head * HD + 16 * ks + 4 * g;
        const float4 a = *reinterpret_cast<const float4*>(base);
        const float4 c = *reinterpret_cast<const float4*>(base + d);
        qv = af4{a.x * scale, a.y * scale, a.z * scale, a.w * scale};
        kv = af4{c.x, c.y, c.z, c.w};
      }
      attn_split4(qv, qh[t][ks], ql[t][ks]);
      attn_split4(kv, kh[t][ks], kl[t][ks]);
    }
  // key validity of this lane's score registers: keys kt * 16 + 4 g + t
  bool kvalid[NT][4];
  long long krow[NT][4];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      krow[kt][t] = token_row(kt * 16 + 4 * g + t);
      kvalid[kt][t] = krow[kt][t] >= 0;
    }
#pragma unroll
  for (int qt = 0; qt < NT; ++qt) {
    af4 sc[NT];
    float m = kAttnNegBig;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      af4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a = attn_mma3(kh[kt][ks], kl[kt][ks], qh[qt][ks], ql[qt][ks], a);   // S^T[key][q]
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (!kvalid[kt][t]) a[t] = kAttnNegBig;
        m = fmaxf(m, a[t]);
      }
      sc[kt] = a;
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        sc[kt][t] = __expf(sc[kt][t] - m);
        l += sc[kt][t];
      }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
    ah4 ph[NT], pl[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) attn_split4(sc[kt], ph[kt], pl[kt]);
#pragma unroll
    for (int dt = 0; dt < KS; ++dt) {
      af4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        af4 vv = {0.f, 0.f, 0.f, 0.f};     // A: V^T[dim = 16 dt + col][key 4 g + t]
#pragma unroll
        for (int t = 0; t < 4; ++t)
          if (kvalid[kt][t]) vv[t] = qkv[(size_t)krow[kt][t] * (size_t)(3 * d) + 2 * d + head * HD + 16 * dt + col];
        ah4 vh, vl;
        attn_split4(vv, vh, vl);
        acc = attn_mma3(vh, vl, ph[kt], pl[kt], acc);                 // O^T[dim 4 g + t][q = col]
      }
      if (row_c[qt] >= 0)
        *reinterpret_cast<float4*>(out + (size_t)row_c[qt] * d + head * HD + 16 * dt + 4 * g) =
            make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
    }
  }
}

// grid (ceil(Lq / 256), heads, B * nsplit) ...
template <bool PARTIAL>
__global__ __launch_bounds__(256) void attention_mfma16_kernel(const float* __restrict__ q, int ldq,
                                                               const float* __restrict__ k, const float* __restrict__ v,
                                                               int ldk, int Lq, int Lk, int kps, int nsplit, float scale,
                                                               float* __restrict__ out, int ldo, float* __restrict__ part,
                                                               AttnDrop drop) {
  constexpr int HD = 16, LS = kAttnLs;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int head = blockIdx.y, b = blockIdx.z / nsplit, sp = blockIdx.z % nsplit;
  const int k0 = sp * kps;
  const int kn = min(kps, Lk - k0);
  const int ktiles = (kn + 15) >> 4;
  float* ks = smem;                                   // [ktiles * 16][LS]
  float* vs = smem + (size_t)ktiles * 16 * LS;
  for (int i = threadIdx.x; i < ktiles * 16 * (HD / 4); i += blockDim.x) {
    const int j = i >> 2, c = (i & 3) * 4;
    float4 kk = make_float4(0.f, 0.f, 0.f, 0.f), vv = kk;     // rows past kn: zeros (their scores are masked below)
    if (j < kn) {
      const size_t off = ((size_t)b * Lk + k0 + j) * ldk + head * HD + c;
      kk = *reinterpret_cast<const float4*>(k + off);
      vv = *reinterpret_cast<const float4*>(v + off);
    }
    *reinterpret_cast<float4*>(ks + j * LS + c) = kk;
    *reinterpret_cast<float4*>(vs + j * LS + c) = vv;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, g = lane >> 4;
  const int qtiles = (Lq + 15) >> 4;
  const int qt_end = min(qtiles, ((int)blockIdx.x + 1) * 16);
  for (int qt = blockIdx.x * 16 + wave; qt < qt_end; qt += 4) {
    const int qi = qt * 16 + col;
    af4 qv = {0.f, 0.f, 0.f, 0.f};
    if (qi < Lq) {
      const float4 t = *reinterpret_cast<const float4*>(q + ((size_t)b * Lq + qi) * ldq + head * HD + 4 * g);
      qv = af4{t.x * scale, t.y * scale, t.z * scale, t.w * scale};
    }
    ah4 qh, ql;
    attn_split4(qv, qh, ql);
    float m = kAttnNegBig, l = 0.f;
    af4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < ktiles; ++kt) {
      const float4 kr = *reinterpret_cast<const float4*>(ks + (kt * 16 + col) * LS + 4 * g);   // A: K[key = col][4g ..]
      ah4 kh, kl;
      attn_split4(af4{kr.x, kr.y, kr.z, kr.w}, kh, kl);
      af4 sc = attn_mma3(kh, kl, qh, ql, af4{0.f, 0.f, 0.f, 0.f});      // S^T[key 4g + t][q = col]
      const int kbase = kt * 16 + 4 * g;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (kbase + t >= kn) sc[t] = kAttnNegBig;
      float tmax = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float mn = fmaxf(m, tmax);
      const float corr = __expf(m - mn);
      af4 pr;
#pragma unroll
      for (int t = 0; t < 4; ++t) pr[t] = __expf(sc[t] - mn);
      float ps = (pr[0] + pr[1]) + (pr[2] + pr[3]);
      ps += __shfl_xor(ps, 16, 64);
      ps += __shfl_xor(ps, 32, 64);
      l = l * corr + ps;
      acc *= corr;
      m = mn;
      if (drop.thresh) {   // training: dropout on the probabilities (the row sum l keeps every term, the products do not)
        const unsigned bh = (unsigned)(b * gridDim.y + head);
#pragma unroll
        for (int t = 0; t < 4; ++t)
          pr[t] = attn_keep(drop.seed, bh, (unsigned)qi, (unsigned)(k0 + kbase + t), drop.thresh) ? pr[t] * drop.inv_keep : 0.f;
      }
      // A: V^T[d = col][key 4g + t] -- transposed read of the LDS tile (16 consecutive banks per register)
      const float* vp = vs + (kbase)*LS + col;
      ah4 vh, vl, ph, pl;
      attn_split4(af4{vp[0], vp[LS], vp[2 * LS], vp[3 * LS]}, vh, vl);
      attn_split4(pr, ph, pl);                                         // B: P^T[key 4g + t][q = col] = the score registers
      acc = attn_mma3(vh, vl, ph, pl, acc);                            // O^T[d = 4g + t][q = col]
    }
    if (qi < Lq) {
      if (PARTIAL) {
        float* pp = part + ((((size_t)b * gridDim.y + head) * nsplit + sp) * Lq + qi) * (HD + 2);
        if (g == 0) { pp[0] = m; pp[1] = l; }
#pragma unroll
        for (int t = 0; t < 4; ++t) pp[2 + 4 * g + t] = acc[t];
      } else {
        const float inv = 1.f / l;
        *reinterpret_cast<float4*>(out + ((size_t)b * Lq + qi) * ldo + head * HD + 4 * g) =
            make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
      }
    }
  }
}

template <int HD>
__global__ void merge_key_splits_kernel(const float* __restrict__ part, int B, int heads, int nsplit, int Lq,
                                        float* __restrict__ out, int ldo) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)B * heads * Lq * HD) return;
  const int c = (int)(t % HD);
  const int qi = (int)((t / HD) % Lq);
  const int head = (int)((t / ((long long)HD * Lq)) % heads);
  const int b = (int)(t / ((long long)HD * Lq * heads));
  const float* pb = part + (((size_t)b * heads + head) * nsplit * Lq + qi) * (HD + 2);
  const size_t sstride = (size_t)Lq * (HD + 2);
  float M = -INFINITY;
  for (int s = 0; s < nsplit; ++s) M = fmaxf(M, pb[s * sstride]);
  float num = 0.f, den = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float w = __expf(pb[s * sstride] - M);
    den = fmaf(pb[s * sstride + 1], w, den);
    num = fmaf(pb[s * sstride + 2 + c], w, num);
  }
  out[((size_t)b * Lq + qi) * ldo + head * HD + c] = num / den;
}

}  // namespace isf

extern "C" {

int isf_window_attention_forward(const float* qkv, int batch_size, int grid_size, int embed_dims, int num_heads,
                                 int window, int shift, float* out, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(batch_size >= 0 && grid_size > 0, ISF_ERR_ARG, "window_attention: bad sizes");
  if (batch_size == 0) return ISF_OK;
  ISF_REQUIRE(qkv && out, ISF_ERR_ARG, "window_attention: null pointer");
  ISF_REQUIRE(num_heads == 8 && window == 6 && (embed_dims == 128 || embed_dims == 256), ISF_ERR_UNSUPPORTED,
              "window_attention: built for 8 heads, 6x6 windows, d in {128, 256} (got %d heads, win %d, d %d)",
              num_heads, window, embed_dims);
  // sst_ops.py:236-248: shift 0 adds `win` to the coordinates, shift 1 adds win/2, then floor-divides.
  // window index 0 of this launch is the first window that holds a grid cell.
  const int y_off = shift ? window / 2 : 0;
  const int nwin = shift ? (grid_size - 1 + window / 2) / window + 1 : (grid_size + window - 1) / window;
  const int hd = embed_dims / num_heads;
  const float scale = 1.0f / sqrtf((float)hd);
  hipStream_t st = as_stream(stream);
  // one wave per (window, head) on the matrix cores (window_attention_mfma_kernel)
  const int total = batch_size * nwin * nwin * 8;
  if (hd == 16)
    hipLaunchKernelGGL((window_attention_mfma_kernel<16>), dim3(ceil_div(total, 4)), dim3(256), 0, st, qkv, grid_size,
                       embed_dims, nwin, y_off, total, scale, out);
  else
    hipLaunchKernelGGL((window_attention_mfma_kernel<32>), dim3(ceil_div(total, 4)), dim3(256), 0, st, qkv, grid_size,
                       embed_dims, nwin, y_off, total, scale, out);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

static int attention_forward_impl(const float* q, int ldq, const float* k, const float* v, int ldkv, int batch_size,
                                  int num_queries, int num_keys, int embed_dims, int num_heads, float* out, int ldo,
                                  isf::AttnDrop drop, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(drop.thresh == 0 || (num_heads > 0 && embed_dims == 16 * num_heads && num_keys <= 512), ISF_ERR_UNSUPPORTED,
              "attention: probability dropout is built for head_dim 16 and <= 512 keys (the training path's two shapes)");
  ISF_REQUIRE(batch_size >= 0 && num_queries >= 0 && num_keys > 0, ISF_ERR_ARG, "attention: bad sizes");
  if (batch_size == 0 || num_queries == 0) return ISF_OK;
  ISF_REQUIRE(q && k && v && out, ISF_ERR_ARG, "attention: null pointer");
  ISF_REQUIRE(embed_dims % num_heads == 0 && ldq % 4 == 0 && ldkv % 4 == 0 && ldo % 4 == 0, ISF_ERR_ARG,
              "attention: strides must be multiples of 4 floats");
  const int hd = embed_dims / num_heads;
  ISF_REQUIRE(hd == 16 || hd == 32, ISF_ERR_UNSUPPORTED, "attention: built for head_dim 16 / 32 (got %d)", hd);
  if (hd == 16) {   // matrix-core kernel: the keys of a workgroup (all of them, or a <= 512-key split) resident in LDS
    hipStream_t st = as_stream(stream);
    // <= 512 keys: one resident set; more (the head's 32400): 512-key splits + merge (256-key splits measured the
    // same: 157 us against 159 us per call incl. the merge, tools/attention_time.py)
    const int kps = num_keys > 512 ? 512 : num_keys, nsplit = ceil_div(num_keys, kps);
    const float scale = 0.25f;   // 1 / sqrt(16)
    const dim3 grid(ceil_div(num_queries, 256), num_heads, batch_size * nsplit), block(256);
    const size_t lds = (size_t)round_up(kps, 16) * kAttnLs * 2 * sizeof(float);
    static bool mfma_attr_set = false;
    if (!mfma_attr_set) {   // 512 keys x 20 floats x (k, v) = 80 KiB
      ISF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_mfma16_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      ISF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_mfma16_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      mfma_attr_set = true;
    }
    if (nsplit == 1) {
      hipLaunchKernelGGL((attention_mfma16_kernel<false>), grid, block, lds, st, q, ldq, k, v, ldkv, num_queries, num_keys,
                         kps, 1, scale, out, ldo, static_cast<float*>(nullptr), drop);
    } else {
      Arena& a = arena_for_stream(st);
      ISF_TRY(a.reset());
      float* part = nullptr;
      ISF_TRY(a.alloc_n(&part, (size_t)batch_size * num_heads * nsplit * num_queries * (hd + 2)));
      hipLaunchKernelGGL((attention_mfma16_kernel<true>), grid, block, lds, st, q, ldq, k, v, ldkv, num_queries, num_keys,
                         kps, nsplit, scale, out, ldo, part, drop);
      const long long total = (long long)batch_size * num_heads * num_queries * hd;
      hipLaunchKernelGGL((merge_key_splits_kernel<16>), dim3(ceil_div(total, 256)), dim3(256), 0, st, part, batch_size,
                         num_heads, nsplit, num_queries, out, ldo);
    }
    ISF_LAUNCH_CHECK();
    return ISF_OK;
  }
  if (num_keys > 512) {   // keys split into <= 512-key chunks + merge
    hipStream_t st = as_stream(stream);
    const int kps = 512, nsplit = ceil_div(num_keys, kps);
    Arena& a = arena_for_stream(as_stream(stream));
    ISF_TRY(a.reset());
    float* part = nullptr;
    ISF_TRY(a.alloc_n(&part, (size_t)batch_size * num_heads * nsplit * num_queries * (hd + 2)));
    static bool split_attr_set = false;
    if (!split_attr_set) {
      ISF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&split_key_attention_kernel<16>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      ISF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&split_key_attention_kernel<32>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
      split_attr_set = true;
    }
    const dim3 grid(ceil_div(num_queries, 256), num_heads, batch_size * nsplit), block(256);
    const size_t lds = (size_t)kps * hd * 2 * sizeof(float);
    const float scale = 1.0f / sqrtf((float)hd);
    const long long total = (long long)batch_size * num_heads * num_queries * hd;
    if (hd == 16) {
      hipLaunchKernelGGL((split_key_attention_kernel<16>), grid, block, lds, st, q, ldq, k, v, ldkv, num_queries,
                         num_keys, kps, nsplit, scale, part);
      hipLaunchKernelGGL((merge_key_splits_kernel<16>), dim3(ceil_div(total, 256)), dim3(256), 0, st, part, batch_size,
                         num_heads, nsplit, num_queries, out, ldo);
    } else {
      hipLaunchKernelGGL((split_key_attention_kernel<32>), grid, block, lds, st, q, ldq, k, v, ldkv, num_queries,
                         num_keys, kps, nsplit, scale, part);
      hipLaunchKernelGGL((merge_key_splits_kernel<32>), dim3(ceil_div(total, 256)), dim3(256), 0, st, part, batch_size,
                         num_heads, nsplit, num_queries, out, ldo);
    }
    ISF_LAUNCH_CHECK();
    return ISF_OK;
  }
  const dim3 grid(ceil_div(num_queries, 256), num_heads, batch_size), block(256);
  const size_t lds = (size_t)num_keys * hd * 2 * sizeof(float);
  const float scale = 1.0f / sqrtf((float)hd);
  hipStream_t st = as_stream(stream);
  static bool attr_set = false;
  if (!attr_set) {   // 512 keys x 32 dims x (k, v) = 128 KB of the 160 KB LDS
    ISF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&small_key_attention_kernel<16>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    ISF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&small_key_attention_kernel<32>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    attr_set = true;
  }
  if (hd == 16)
    hipLaunchKernelGGL((small_key_attention_kernel<16>), grid, block, lds, st, q, ldq, k, v, ldkv, num_queries,
                       num_keys, scale, out, ldo);
  else
    hipLaunchKernelGGL((small_key_attention_kernel<32>), grid, block, lds, st, q, ldq, k, v, ldkv, num_queries,
                       num_keys, scale, out, ldo);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int isf_attention_forward(const float* q, int ldq, const float* k, const float* v, int ldkv, int batch_size,
                          int num_queries, int num_keys, int embed_dims, int num_heads, float* out, int ldo,
                          isf_stream_t stream) {
  return attention_forward_impl(q, ldq, k, v, ldkv, batch_size, num_queries, num_keys, embed_dims, num_heads, out, ldo,
                                isf::attn_drop_of(0.f, 0ull), stream);
}

int isf_attention_forward_dropout(const float* q, int ldq, const float* k, const float* v, int ldkv, int batch_size,
                                  int num_queries, int num_keys, int embed_dims, int num_heads, float dropout_p,
                                  unsigned long long seed, float* out, int ldo, isf_stream_t stream) {
  ISF_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, ISF_ERR_ARG, "attention: dropout probability %g", (double)dropout_p);
  return attention_forward_impl(q, ldq, k, v, ldkv, batch_size, num_queries, num_keys, embed_dims, num_heads, out, ldo,
                                isf::attn_drop_of(dropout_p, seed), stream);
}
}  // extern "C"
