// isf_attention_bwd.hip -- SURVEY 8f #2: backward of the softmax-attention cores (isf_attention.hip).
//
// Reference: autograd through nn.MultiheadAttention / multi_head_attention_forward (sst_basic_block_v2.py:41-75,
// fusion_encoder.py:371-470): bmm, softmax, bmm backward kernels with the [B*heads, Lq, Lk] probability matrix
// materialised in HBM (32400 x 200 x 8 heads x B floats for the instance-to-scene attention).
//
// Here nothing of size Lq x Lk touches HBM: probabilities are recomputed from q, k and the row statistics
//   L_i = logsumexp_j(s_ij),   D_i = dO_i . O_i          (s = q k^T / sqrt(hd)),
// and every gradient row is produced by exactly one wave in a fixed order (no atomics, deterministic):
//   rows    wave per (b, head, query i), lanes over keys:   L_i, D_i, dQ_i = scale * sum_j p_ij (dO_i.v_j - D_i) k_j
//   cols    wave per (b, head, key j),   lanes over queries: dV_j = sum_i p_ij dO_i,
//                                                            dK_j = scale * sum_i p_ij (dO_i.v_j - D_i) q_i
//   window  the 36-token windows of the dense grid: one wave per (window, head) does both roles out of LDS.
// fp32 VALU like the forward cores (head dim 16: these are dot products of 16 floats).
#include <algorithm>

#include "isf_common.h"

namespace isf {

template <int HD>
__device__ __forceinline__ void load_row(const float* __restrict__ p, float r[HD]) {
#pragma unroll
  for (int c = 0; c < HD; c += 4) {
    const float4 t = *reinterpret_cast<const float4*>(p + c);
    r[c] = t.x; r[c + 1] = t.y; r[c + 2] = t.z; r[c + 3] = t.w;
  }
}

template <int HD>
__device__ __forceinline__ float dot_row(const float a[HD], const float b[HD]) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < HD; ++c) s = fmaf(a[c], b[c], s);
  return s;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}

// ---------------------------------------------------------------------------------------------------------------- rows
template <int HD>
__global__ __launch_bounds__(256) void attention_bwd_rows_kernel(
    const float* __restrict__ q, int ldq, const float* __restrict__ k, const float* __restrict__ v, int ldkv,
    const float* __restrict__ out, const float* __restrict__ gout, int ldo, int Lq, int Lk, float scale,
    float* __restrict__ gq, int ldgq, float* __restrict__ stat /* [B, heads, Lq, 2] = (L, D) */, AttnDrop drop) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int head = blockIdx.y, b = blockIdx.z, heads = gridDim.y;
  if (i >= Lq) return;                                     // wave-uniform
  float qi[HD], go[HD], oi[HD];
  load_row<HD>(q + ((size_t)b * Lq + i) * ldq + head * HD, qi);
  load_row<HD>(gout + ((size_t)b * Lq + i) * ldo + head * HD, go);
  load_row<HD>(out + ((size_t)b * Lq + i) * ldo + head * HD, oi);
  const float D = dot_row<HD>(go, oi);
  const float* kb = k + (size_t)b * Lk * ldkv + head * HD;
  const float* vb = v + (size_t)b * Lk * ldkv + head * HD;
  // pass 1: L = logsumexp
  float m = -INFINITY, sum = 0.f;
  for (int j = lane; j < Lk; j += 64) {
    float kj[HD];
    load_row<HD>(kb + (size_t)j * ldkv, kj);
    const float s = scale * dot_row<HD>(qi, kj);
    const float nm = fmaxf(m, s);
    sum = sum * __expf(m - nm) + __expf(s - nm);
    m = nm;
  }
  const float M = wave_max(m);
  sum = wave_sum(m == -INFINITY ? 0.f : sum * __expf(m - M));
  const float L = M + __logf(sum);
  // pass 2: dQ
  float acc[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) acc[c] = 0.f;
  for (int j = lane; j < Lk; j += 64) {
    float kj[HD], vj[HD];
    load_row<HD>(kb + (size_t)j * ldkv, kj);
    load_row<HD>(vb + (size_t)j * ldkv, vj);
    const float p = __expf(scale * dot_row<HD>(qi, kj) - L);
    float dp = dot_row<HD>(go, vj);
    if (drop.thresh)   // the forward's keep / drop decision for (query i, key j), recomputed
      dp = attn_keep(drop.seed, (unsigned)(b * heads + head), (unsigned)i, (unsigned)j, drop.thresh) ? dp * drop.inv_keep : 0.f;
    const float ds = p * (dp - D);
#pragma unroll
    for (int c = 0; c < HD; ++c) acc[c] = fmaf(ds, kj[c], acc[c]);
  }
#pragma unroll
  for (int c = 0; c < HD; ++c) acc[c] = wave_sum(acc[c]) * scale;
  if (lane == 0) {
    float* g = gq + ((size_t)b * Lq + i) * ldgq + head * HD;
#pragma unroll
    for (int c = 0; c < HD; c += 4) *reinterpret_cast<float4*>(g + c) = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
    float* st = stat + (((size_t)b * heads + head) * Lq + i) * 2;
    st[0] = L;
    st[1] = D;
  }
}

// ---------------------------------------------------------------------------------------------------------------- cols
// Round 5: the query range is cut into `chunks` pieces (grid.x = key blocks x chunks): with one wave per (key, head,
// sample) walking ALL queries, the 32400 x 200 instance-to-scene attention had 3 200 waves of 506 iterations each -- a
// dozen waves per CU -- and took 0.72 ms per call (profiles/r05_train_step_after.txt).  chunks > 1: gk / gv are the
// partial buffers [chunks][B * Lk][ldgkv], added in chunk order by attention_bwd_cols_reduce_kernel (deterministic).
template <int HD>
__global__ __launch_bounds__(256) void attention_bwd_cols_kernel(
    const float* __restrict__ q, int ldq, const float* __restrict__ k, const float* __restrict__ v, int ldkv,
    const float* __restrict__ gout, int ldo, int Lq, int Lk, float scale, const float* __restrict__ stat,
    float* __restrict__ gk, float* __restrict__ gv, int ldgkv, int chunks, int batch, AttnDrop drop) {
  const int lane = threadIdx.x & 63;
  const int kblocks = (Lk + 3) / 4;
  const int chunk = blockIdx.x / kblocks;
  const int j = (blockIdx.x % kblocks) * 4 + (threadIdx.x >> 6);
  const int head = blockIdx.y, b = blockIdx.z, heads = gridDim.y;
  if (j >= Lk) return;                                     // wave-uniform
  const int qc = (Lq + chunks - 1) / chunks;
  const int i_begin = chunk * qc, i_end = min(Lq, i_begin + qc);
  gk += (size_t)chunk * batch * Lk * ldgkv;
  gv += (size_t)chunk * batch * Lk * ldgkv;
  float kj[HD], vj[HD], dk[HD], dv[HD];
  load_row<HD>(k + ((size_t)b * Lk + j) * ldkv + head * HD, kj);
  load_row<HD>(v + ((size_t)b * Lk + j) * ldkv + head * HD, vj);
#pragma unroll
  for (int c = 0; c < HD; ++c) dk[c] = dv[c] = 0.f;
  const float* qb = q + (size_t)b * Lq * ldq + head * HD;
  const float* gb = gout + (size_t)b * Lq * ldo + head * HD;
  const float* st = stat + ((size_t)b * heads + head) * Lq * 2;
  for (int i = i_begin + lane; i < i_end; i += 64) {
    float qi[HD], go[HD];
    load_row<HD>(qb + (size_t)i * ldq, qi);
    load_row<HD>(gb + (size_t)i * ldo, go);
    const float2 ld = *reinterpret_cast<const float2*>(st + (size_t)i * 2);
    const float p = __expf(scale * dot_row<HD>(qi, kj) - ld.x);
    float dp = dot_row<HD>(go, vj), pm = p;
    if (drop.thresh) {
      const float mk = attn_keep(drop.seed, (unsigned)(b * heads + head), (unsigned)i, (unsigned)j, drop.thresh) ? drop.inv_keep : 0.f;
      dp *= mk;
      pm *= mk;
    }
    const float ds = p * (dp - ld.y);
#pragma unroll
    for (int c = 0; c < HD; ++c) {
      dv[c] = fmaf(pm, go[c], dv[c]);
      dk[c] = fmaf(ds, qi[c], dk[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < HD; ++c) {
    dv[c] = wave_sum(dv[c]);
    dk[c] = wave_sum(dk[c]) * scale;
  }
  if (lane == 0) {
    float* g1 = gk + ((size_t)b * Lk + j) * ldgkv + head * HD;
    float* g2 = gv + ((size_t)b * Lk + j) * ldgkv + head * HD;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      *reinterpret_cast<float4*>(g1 + c) = make_float4(dk[c], dk[c + 1], dk[c + 2], dk[c + 3]);
      *reinterpret_cast<float4*>(g2 + c) = make_float4(dv[c], dv[c + 1], dv[c + 2], dv[c + 3]);
    }
  }
}

// ---------------------------------------------------------------------------------------------- many queries (round 5)
// The two kernels above put a WAVE on a row (or column) and spread the other index over its lanes: every lane fetches its
// own 64-byte key / query row from L1 for a handful of FMAs, and every result ends in a 64-lane reduction -- 20 GB of L1
// traffic and 2.6 + 2.2 ms per training step on the 32400 x 200 instance-to-scene attention
// (profiles/r05_train_step_after.txt).  With thousands of queries the natural owner of a row is a LANE:
//   rows_lanes   lane = query i (q_i, dO_i, O_i in registers), the (b, head)'s keys and values staged in LDS tile by tile
//                and read as BROADCASTS (all lanes the same address: conflict-free, no per-lane loads); two passes over
//                the keys (L_i, then dQ_i); no cross-lane reduction at all.
//   cols_lanes   lane = key j (k_j, v_j, dK_j, dV_j in registers), a tile of queries (q_i, dO_i, L_i, D_i) staged in LDS and
//                broadcast; query chunks -> partial buffers -> the ordered reduce above.
// Same sums as the wave kernels in a different (fixed) order: deterministic, equal to fp32 rounding.
constexpr int kAttTile = 128;   // keys (rows_lanes) / queries (cols_lanes) per LDS tile

template <int HD>
__global__ __launch_bounds__(256) void attention_bwd_rows_lanes_kernel(
    const float* __restrict__ q, int ldq, const float* __restrict__ k, const float* __restrict__ v, int ldkv,
    const float* __restrict__ out, const float* __restrict__ gout, int ldo, int Lq, int Lk, float scale,
    float* __restrict__ gq, int ldgq, float* __restrict__ stat, AttnDrop drop) {
  __shared__ __attribute__((aligned(16))) float ks[kAttTile][HD];
  __shared__ __attribute__((aligned(16))) float vs[kAttTile][HD];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int head = blockIdx.y, b = blockIdx.z, heads = gridDim.y;
  const bool live = i < Lq;
  const int ic = live ? i : Lq - 1;
  float qi[HD], go[HD], oi[HD];
  load_row<HD>(q + ((size_t)b * Lq + ic) * ldq + head * HD, qi);
  load_row<HD>(gout + ((size_t)b * Lq + ic) * ldo + head * HD, go);
  load_row<HD>(out + ((size_t)b * Lq + ic) * ldo + head * HD, oi);
#pragma unroll
  for (int c = 0; c < HD; ++c) qi[c] *= scale;                 // s_ij = (scale q_i) . k_j
  const float D = dot_row<HD>(go, oi);
  const float* kb = k + (size_t)b * Lk * ldkv + head * HD;
  const float* vb = v + (size_t)b * Lk * ldkv + head * HD;
  auto stage = [&](int j0, bool with_v) {                      // rows j0 .. j0 + kAttTile of K (and V) -> LDS
    __syncthreads();
    for (int e = threadIdx.x; e < kAttTile * (HD / 4); e += 256) {
      const int r = e / (HD / 4), c4 = (e % (HD / 4)) * 4;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), bb = a;
      if (j0 + r < Lk) {
        a = *reinterpret_cast<const float4*>(kb + (size_t)(j0 + r) * ldkv + c4);
        if (with_v) bb = *reinterpret_cast<const float4*>(vb + (size_t)(j0 + r) * ldkv + c4);
      }
      *reinterpret_cast<float4*>(&ks[r][c4]) = a;
      if (with_v) *reinterpret_cast<float4*>(&vs[r][c4]) = bb;
    }
    __syncthreads();
  };
  // pass 1: L_i = logsumexp_j s_ij (running maximum per lane)
  float m = -INFINITY, sum = 0.f;
  for (int j0 = 0; j0 < Lk; j0 += kAttTile) {
    stage(j0, false);
    const int nj = min(kAttTile, Lk - j0);
    for (int j = 0; j < nj; ++j) {
      float sdot = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) sdot = fmaf(qi[c], ks[j][c], sdot);
      const float nm = fmaxf(m, sdot);
      sum = sum * __expf(m - nm) + __expf(sdot - nm);
      m = nm;
    }
  }
  const float L = m + __logf(sum);
  // pass 2: dQ_i = scale * sum_j p_ij (dO_i . v_j - D_i) k_j
  float acc[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) acc[c] = 0.f;
  for (int j0 = 0; j0 < Lk; j0 += kAttTile) {
    stage(j0, true);
    const int nj = min(kAttTile, Lk - j0);
    for (int j = 0; j < nj; ++j) {
      float sdot = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) {
        sdot = fmaf(qi[c], ks[j][c], sdot);
        dp = fmaf(go[c], vs[j][c], dp);
      }
      if (drop.thresh)
        dp = attn_keep(drop.seed, (unsigned)(b * heads + head), (unsigned)i, (unsigned)(j0 + j), drop.thresh) ? dp * drop.inv_keep : 0.f;
      const float ds = __expf(sdot - L) * (dp - D);
#pragma unroll
      for (int c = 0; c < HD; ++c) acc[c] = fmaf(ds, ks[j][c], acc[c]);
    }
  }
  if (live) {
    float* g = gq + ((size_t)b * Lq + i) * ldgq + head * HD;
#pragma unroll
    for (int c = 0; c < HD; c += 4)
      *reinterpret_cast<float4*>(g + c) = make_float4(acc[c] * scale, acc[c + 1] * scale, acc[c + 2] * scale, acc[c + 3] * scale);
    float* st = stat + (((size_t)b * heads + head) * Lq + i) * 2;
    st[0] = L;
    st[1] = D;
  }
}

// grid.x = key blocks (256 keys each) x chunks of the query range; writes the chunk's partial dK / dV rows
template <int HD>
__global__ __launch_bounds__(256) void attention_bwd_cols_lanes_kernel(
    const float* __restrict__ q, int ldq, const float* __restrict__ k, const float* __restrict__ v, int ldkv,
    const float* __restrict__ gout, int ldo, int Lq, int Lk, float scale, const float* __restrict__ stat,
    float* __restrict__ gk, float* __restrict__ gv, int ldgkv, int chunks, int batch, AttnDrop drop) {
  __shared__ __attribute__((aligned(16))) float qs[kAttTile][HD];
  __shared__ __attribute__((aligned(16))) float gs[kAttTile][HD];
  __shared__ float2 ld_s[kAttTile];
  const int kblocks = (Lk + 255) / 256;
  const int chunk = blockIdx.x / kblocks;
  const int j = (blockIdx.x % kblocks) * 256 + threadIdx.x;
  const int head = blockIdx.y, b = blockIdx.z, heads = gridDim.y;
  const bool live = j < Lk;
  const int jc = live ? j : Lk - 1;
  float kj[HD], vj[HD], dk[HD], dv[HD];
  load_row<HD>(k + ((size_t)b * Lk + jc) * ldkv + head * HD, kj);
  load_row<HD>(v + ((size_t)b * Lk + jc) * ldkv + head * HD, vj);
#pragma unroll
  for (int c = 0; c < HD; ++c) { kj[c] *= scale; dk[c] = dv[c] = 0.f; }   // s_ij = q_i . (scale k_j)
  const int qc = (Lq + chunks - 1) / chunks;
  const int i_begin = chunk * qc, i_end = min(Lq, i_begin + qc);
  const float* qb = q + (size_t)b * Lq * ldq + head * HD;
  const float* gb = gout + (size_t)b * Lq * ldo + head * HD;
  const float* st = stat + ((size_t)b * heads + head) * Lq * 2;
  for (int i0 = i_begin; i0 < i_end; i0 += kAttTile) {
    __syncthreads();
    for (int e = threadIdx.x; e < kAttTile * (HD / 4); e += 256) {
      const int r = e / (HD / 4), c4 = (e % (HD / 4)) * 4;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), bb = a;
      if (i0 + r < i_end) {
        a = *reinterpret_cast<const float4*>(qb + (size_t)(i0 + r) * ldq + c4);
        bb = *reinterpret_cast<const float4*>(gb + (size_t)(i0 + r) * ldo + c4);
      }
      *reinterpret_cast<float4*>(&qs[r][c4]) = a;
      *reinterpret_cast<float4*>(&gs[r][c4]) = bb;
    }
    if (threadIdx.x < kAttTile)
      ld_s[threadIdx.x] = i0 + (int)threadIdx.x < i_end ? *reinterpret_cast<const float2*>(st + (size_t)(i0 + threadIdx.x) * 2)
                                                        : make_float2(INFINITY, 0.f);   // p = exp(s - inf) = 0
    __syncthreads();
    const int ni = min(kAttTile, i_end - i0);
    for (int i = 0; i < ni; ++i) {
      float sdot = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) {
        sdot = fmaf(qs[i][c], kj[c], sdot);
        dp = fmaf(gs[i][c], vj[c], dp);
      }
      const float2 ld = ld_s[i];
      const float p = __expf(sdot - ld.x);
      float pm = p;
      if (drop.thresh) {
        const float mk = attn_keep(drop.seed, (unsigned)(b * heads + head), (unsigned)(i0 + i), (unsigned)j, drop.thresh) ? drop.inv_keep : 0.f;
        dp *= mk;
        pm *= mk;
      }
      const float ds = p * (dp - ld.y);
#pragma unroll
      for (int c = 0; c < HD; ++c) {
        dv[c] = fmaf(pm, gs[i][c], dv[c]);
        dk[c] = fmaf(ds, qs[i][c], dk[c]);
      }
    }
  }
  if (live) {
    float* g1 = gk + ((size_t)chunk * batch * Lk + (size_t)b * Lk + j) * ldgkv + head * HD;
    float* g2 = gv + ((size_t)chunk * batch * Lk + (size_t)b * Lk + j) * ldgkv + head * HD;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      *reinterpret_cast<float4*>(g1 + c) = make_float4(dk[c] * scale, dk[c + 1] * scale, dk[c + 2] * scale, dk[c + 3] * scale);
      *reinterpret_cast<float4*>(g2 + c) = make_float4(dv[c], dv[c + 1], dv[c + 2], dv[c + 3]);
    }
  }
}

// gk / gv [rows][ld] (first `cols` columns) = sum over chunks, in order, of the partial buffers
__global__ void attention_bwd_cols_reduce_kernel(const float* __restrict__ pk, const float* __restrict__ pv, int chunks,
                                                 size_t rows, int cols, int ld, float* __restrict__ gk,
                                                 float* __restrict__ gv) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * (size_t)cols) return;
  const size_t r = e / cols, o = r * ld + e % cols;
  float a = 0.f, b = 0.f;
  for (int c = 0; c < chunks; ++c) {
    a += pk[(size_t)c * rows * ld + o];
    b += pv[(size_t)c * rows * ld + o];
  }
  gk[o] = a;
  gv[o] = b;
}

// ---------------------------------------------------------------------------------------------------------------- window
// same geometry as window_attention_kernel; HPB heads per workgroup (one wave each)
template <int HD, int WIN, int HPB>
__global__ __launch_bounds__(64 * HPB) void window_attention_bwd_kernel(const float* __restrict__ qkv,
                                                                       const float* __restrict__ gout, int S, int d,
                                                                       int y_off, float scale, int head_groups,
                                                                       float* __restrict__ gqkv) {
  constexpr int T = WIN * WIN;
  static_assert(T <= 64, "window must fit one wave");
  __shared__ __attribute__((aligned(16))) float sm[HPB][4][T][HD];   // q | k | v | dO of the wave's head
  __shared__ float st[HPB][2][T];                                    // L, D per query
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.z / head_groups, head = (blockIdx.z % head_groups) * HPB + w;
  const int y0 = (int)blockIdx.y * WIN - y_off, x0 = (int)blockIdx.x * WIN - y_off;
  const int y = y0 + lane / WIN, x = x0 + lane % WIN;
  const bool valid = lane < T && y >= 0 && y < S && x >= 0 && x < S;
  const size_t row = ((size_t)b * S + (valid ? y : 0)) * S + (valid ? x : 0);
  float qi[HD], ki[HD], vi[HD], go[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) qi[c] = ki[c] = vi[c] = go[c] = 0.f;
  if (valid) {
    const float* base = qkv + row * (size_t)(3 * d) + head * HD;
    load_row<HD>(base, qi);
    load_row<HD>(base + d, ki);
    load_row<HD>(base + 2 * d, vi);
    load_row<HD>(gout + row * (size_t)d + head * HD, go);
  }
  if (lane < T) {
#pragma unroll
    for (int c = 0; c < HD; ++c) {
      sm[w][0][lane][c] = qi[c];
      sm[w][1][lane][c] = ki[c];
      sm[w][2][lane][c] = vi[c];
      sm[w][3][lane][c] = go[c];
    }
  }
  __syncthreads();
  float dq[HD], dk[HD], dv[HD];
#pragma unroll
  for (int c = 0; c < HD; ++c) dq[c] = dk[c] = dv[c] = 0.f;
  // ---- query role: L_i, D_i, dQ_i
  float L = 0.f, D = 0.f;
  if (valid) {
    float m = -INFINITY;
    float s[T];
#pragma unroll
    for (int j = 0; j < T; ++j) {
      const int yy = y0 + j / WIN, xx = x0 + j % WIN;
      const bool ok = yy >= 0 && yy < S && xx >= 0 && xx < S;   // wave-uniform
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) a = fmaf(qi[c], sm[w][1][j][c], a);
      s[j] = ok ? a * scale : -INFINITY;
      m = fmaxf(m, s[j]);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < T; ++j) sum += __expf(s[j] - m);
    L = m + __logf(sum);
    // D_i = sum_j p_ij dO_i . v_j
#pragma unroll
    for (int j = 0; j < T; ++j) {
      float dp = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) dp = fmaf(go[c], sm[w][2][j][c], dp);
      D = fmaf(__expf(s[j] - L), dp, D);
    }
#pragma unroll
    for (int j = 0; j < T; ++j) {
      float dp = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) dp = fmaf(go[c], sm[w][2][j][c], dp);
      const float ds = __expf(s[j] - L) * (dp - D);   // masked keys: exp(-inf) = 0
#pragma unroll
      for (int c = 0; c < HD; ++c) dq[c] = fmaf(ds, sm[w][1][j][c], dq[c]);
    }
  }
  if (lane < T) {
    st[w][0][lane] = L;
    st[w][1][lane] = D;
  }
  __syncthreads();
  // ---- key role: dK_j, dV_j
  if (valid) {
#pragma unroll 4
    for (int i = 0; i < T; ++i) {
      const int yy = y0 + i / WIN, xx = x0 + i % WIN;
      if (!(yy >= 0 && yy < S && xx >= 0 && xx < S)) continue;   // wave-uniform
      float a = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < HD; ++c) {
        a = fmaf(sm[w][0][i][c], ki[c], a);
        dp = fmaf(sm[w][3][i][c], vi[c], dp);
      }
      const float p = __expf(a * scale - st[w][0][i]);
      const float ds = p * (dp - st[w][1][i]);
#pragma unroll
      for (int c = 0; c < HD; ++c) {
        dv[c] = fmaf(p, sm[w][3][i][c], dv[c]);
        dk[c] = fmaf(ds, sm[w][0][i][c], dk[c]);
      }
    }
    float* g = gqkv + row * (size_t)(3 * d) + head * HD;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      *reinterpret_cast<float4*>(g + c) =
          make_float4(dq[c] * scale, dq[c + 1] * scale, dq[c + 2] * scale, dq[c + 3] * scale);
      *reinterpret_cast<float4*>(g + d + c) =
          make_float4(dk[c] * scale, dk[c + 1] * scale, dk[c + 2] * scale, dk[c + 3] * scale);
      *reinterpret_cast<float4*>(g + 2 * d + c) = make_float4(dv[c], dv[c + 1], dv[c + 2], dv[c + 3]);
    }
  }
}

}  // namespace isf

extern "C" {

static int attention_backward_impl(const float* q, int ldq, const float* k, const float* v, int ldkv, const float* out,
                                   const float* grad_out, int ldo, int batch_size, int num_queries, int num_keys,
                                   int embed_dims, int num_heads, float* grad_q, int ldgq, float* grad_k, float* grad_v,
                                   int ldgkv, isf::AttnDrop drop, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(batch_size >= 0 && num_queries >= 0 && num_keys > 0 && num_heads > 0, ISF_ERR_ARG,
              "attention_backward: bad sizes");
  if (batch_size == 0) return ISF_OK;
  ISF_REQUIRE(embed_dims == 16 * num_heads, ISF_ERR_UNSUPPORTED, "attention_backward: head dim %d (built for 16)",
              num_heads ? embed_dims / num_heads : 0);
  ISF_REQUIRE(q && k && v && out && grad_out && grad_q && grad_k && grad_v, ISF_ERR_ARG,
              "attention_backward: null pointer");
  ISF_REQUIRE(ldq % 4 == 0 && ldkv % 4 == 0 && ldo % 4 == 0 && ldgq % 4 == 0 && ldgkv % 4 == 0, ISF_ERR_ARG,
              "attention_backward: leading dimensions must be multiples of 4 floats");
  hipStream_t st = as_stream(stream);
  if (num_queries == 0) {
    // no query: the key / value gradients are zero
    ISF_HIP_TRY(hipMemset2DAsync(grad_k, sizeof(float) * ldgkv, 0, sizeof(float) * embed_dims,
                                 (size_t)batch_size * num_keys, st));
    ISF_HIP_TRY(hipMemset2DAsync(grad_v, sizeof(float) * ldgkv, 0, sizeof(float) * embed_dims,
                                 (size_t)batch_size * num_keys, st));
    return ISF_OK;
  }
  Arena& a = arena_for_stream(as_stream(stream));
  ISF_TRY(a.reset());
  float* stat = nullptr;
  ISF_TRY(a.alloc_n(&stat, (size_t)batch_size * num_heads * num_queries * 2));
  const float scale = 1.f / sqrtf(16.f);
  const bool many = num_queries >= 2048;   // lanes own queries / keys (see attention_bwd_rows_lanes_kernel)
  if (many)
    hipLaunchKernelGGL((attention_bwd_rows_lanes_kernel<16>), dim3(ceil_div(num_queries, 256), num_heads, batch_size), dim3(256),
                       0, st, q, ldq, k, v, ldkv, out, grad_out, ldo, num_queries, num_keys, scale, grad_q, ldgq, stat, drop);
  else
    hipLaunchKernelGGL((attention_bwd_rows_kernel<16>), dim3(ceil_div(num_queries, 4), num_heads, batch_size), dim3(256),
                       0, st, q, ldq, k, v, ldkv, out, grad_out, ldo, num_queries, num_keys, scale, grad_q, ldgq, stat, drop);
  ISF_LAUNCH_CHECK();
  if (many) {
    // enough (key block, chunk, head, sample) workgroups to fill the chip: ~2000 of them, at least 2 query tiles each
    const int kblocks = ceil_div(num_keys, 256);
    int chunks = std::max(1, 2048 / std::max(1, kblocks * num_heads * batch_size));
    chunks = std::max(1, std::min(chunks, num_queries / (2 * kAttTile)));
    const size_t rows = (size_t)batch_size * num_keys;
    float *pk = nullptr, *pv = nullptr;
    ISF_TRY(a.alloc_n(&pk, (size_t)chunks * rows * ldgkv));
    ISF_TRY(a.alloc_n(&pv, (size_t)chunks * rows * ldgkv));
    hipLaunchKernelGGL((attention_bwd_cols_lanes_kernel<16>), dim3(kblocks * chunks, num_heads, batch_size), dim3(256), 0, st,
                       q, ldq, k, v, ldkv, grad_out, ldo, num_queries, num_keys, scale, stat, pk, pv, ldgkv, chunks, batch_size, drop);
    hipLaunchKernelGGL(attention_bwd_cols_reduce_kernel, dim3(ceil_div((long long)rows * embed_dims, 256)), dim3(256), 0, st,
                       pk, pv, chunks, rows, embed_dims, ldgkv, grad_k, grad_v);
    ISF_LAUNCH_CHECK();
    return ISF_OK;
  }
  const int chunks = std::max(1, std::min(64, num_queries / 1024));
  if (chunks == 1) {
    hipLaunchKernelGGL((attention_bwd_cols_kernel<16>), dim3(ceil_div(num_keys, 4), num_heads, batch_size), dim3(256), 0,
                       st, q, ldq, k, v, ldkv, grad_out, ldo, num_queries, num_keys, scale, stat, grad_k, grad_v, ldgkv, 1,
                       batch_size, drop);
  } else {
    const size_t rows = (size_t)batch_size * num_keys;
    float *pk = nullptr, *pv = nullptr;
    ISF_TRY(a.alloc_n(&pk, (size_t)chunks * rows * ldgkv));
    ISF_TRY(a.alloc_n(&pv, (size_t)chunks * rows * ldgkv));
    hipLaunchKernelGGL((attention_bwd_cols_kernel<16>), dim3(ceil_div(num_keys, 4) * chunks, num_heads, batch_size),
                       dim3(256), 0, st, q, ldq, k, v, ldkv, grad_out, ldo, num_queries, num_keys, scale, stat, pk, pv,
                       ldgkv, chunks, batch_size, drop);
    hipLaunchKernelGGL(attention_bwd_cols_reduce_kernel, dim3(ceil_div((long long)rows * embed_dims, 256)), dim3(256), 0, st,
                       pk, pv, chunks, rows, embed_dims, ldgkv, grad_k, grad_v);
  }
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int isf_attention_backward(const float* q, int ldq, const float* k, const float* v, int ldkv, const float* out,
                           const float* grad_out, int ldo, int batch_size, int num_queries, int num_keys,
                           int embed_dims, int num_heads, float* grad_q, int ldgq, float* grad_k, float* grad_v,
                           int ldgkv, isf_stream_t stream) {
  return attention_backward_impl(q, ldq, k, v, ldkv, out, grad_out, ldo, batch_size, num_queries, num_keys, embed_dims,
                                 num_heads, grad_q, ldgq, grad_k, grad_v, ldgkv, isf::attn_drop_of(0.f, 0ull), stream);
}

int isf_attention_backward_dropout(const float* q, int ldq, const float* k, const float* v, int ldkv, const float* out,
                                   const float* grad_out, int ldo, int batch_size, int num_queries, int num_keys,
                                   int embed_dims, int num_heads, float dropout_p, unsigned long long seed, float* grad_q,
                                   int ldgq, float* grad_k, float* grad_v, int ldgkv, isf_stream_t stream) {
  ISF_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, ISF_ERR_ARG, "attention_backward: dropout probability %g", (double)dropout_p);
  return attention_backward_impl(q, ldq, k, v, ldkv, out, grad_out, ldo, batch_size, num_queries, num_keys, embed_dims,
                                 num_heads, grad_q, ldgq, grad_k, grad_v, ldgkv, isf::attn_drop_of(dropout_p, seed), stream);
}

int isf_window_attention_backward(const float* qkv, const float* grad_out, int batch_size, int grid_size,
                                  int embed_dims, int num_heads, int window, int shift, float* grad_qkv,
                                  isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(batch_size >= 0 && grid_size > 0, ISF_ERR_ARG, "window_attention_backward: bad sizes");
  if (batch_size == 0) return ISF_OK;
  ISF_REQUIRE(qkv && grad_out && grad_qkv, ISF_ERR_ARG, "window_attention_backward: null pointer");
  const int hd = num_heads > 0 ? embed_dims / num_heads : 0;
  ISF_REQUIRE(window == 6 && (hd == 16 || hd == 32) && hd * num_heads == embed_dims && num_heads % 4 == 0,
              ISF_ERR_UNSUPPORTED,
              "window_attention_backward: built for 6x6 windows, head dim 16 / 32, heads %% 4 == 0 (got window %d, "
              "dims %d, heads %d)", window, embed_dims, num_heads);
  // window origins exactly as the forward: shift 0 -> aligned at 0; shift 1 -> offset by window / 2
  const int y_off = shift ? window / 2 : 0;
  const int nwin = ceil_div(grid_size + y_off, window);
  const float scale = 1.f / sqrtf((float)hd);
  if (hd == 16) {
    const int groups = num_heads / 4;
    hipLaunchKernelGGL((window_attention_bwd_kernel<16, 6, 4>), dim3(nwin, nwin, batch_size * groups), dim3(256), 0,
                       as_stream(stream), qkv, grad_out, grid_size, embed_dims, y_off, scale, groups, grad_qkv);
  } else {
    const int groups = num_heads / 2;
    hipLaunchKernelGGL((window_attention_bwd_kernel<32, 6, 2>), dim3(nwin, nwin, batch_size * groups), dim3(128), 0,
                       as_stream(stream), qkv, grad_out, grid_size, embed_dims, y_off, scale, groups, grad_qkv);
  }
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

}  // extern "C"
