// isf_linear.hip -- fused row GEMM  Y = epilogue(X . W^T + b)  for the transformer pieces of HSF / IGF
// (SST window-attention encoder layers A10/A11, MSDeformAttn projections A13, instance-to-scene attention A14).
//
// Reference: separate cuBLAS GEMM + bias + LayerNorm / GELU / residual kernels per nn.Linear
// (sst_basic_block_v2.py:104-126, fusion_encoder.py:560-674).  Here one launch per Linear:
//   arithmetic  f16 hi/lo split of both operands, 3 x v_mfma_f32_16x16x32_f16, fp32 accumulate
//               (fp32-class accuracy, see isf_spconv16.hip); weights pre-split once (isf_pack_linear);
//   tiling      workgroup = 64 or 128 rows x all N (N in chunks of 128 or 256 columns); A fragments converted
//               once and kept in registers for the whole K (K <= 256); B fragments staged once per workgroup
//               through a double-buffered LDS ring;
//   epilogue    + bias, + per-row table add (T[idx[r]]: the window position-embedding contribution pos.Wq),
//               activation (ReLU / exact GELU), + residual, LayerNorm over the row (needs the whole row in
//               one chunk: N <= 256), written once.
#include <stdlib.h>

#include "isf_common.h"

namespace isf {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void lin_split8(const f32x8 v, h8& hi, h8& lo) {
  hi = __builtin_convertvector(v, h8);
  const f32x8 r = v - __builtin_convertvector(hi, f32x8);
  lo = __builtin_convertvector(r, h8);
}

struct LinearEpilogue {
  const float* bias;       // [N] or null
  const float* table;      // [T, N] or null: y += table[idx[r]]
  const int32_t* idx;      // [M]
  const float* residual;   // [M, N] or null
  const float* ln_gamma;   // [N] or null -> LayerNorm(eps)
  const float* ln_beta;
  float ln_eps;
  int act;                 // 0 none, 1 relu, 2 gelu (erf)
  // channels-first addressing (0 = row-major [M, C]; hw > 0 = tensor [B, C, hw] with row r = b*hw + pos):
  int x_hw, res_hw, y_hw;
};

// packedW[kc][nt][hi|lo][lane][8 halves]: element jj = W[16 nt + (lane&15)][32 kc + 8 (lane>>4) + jj] * 2^sw
// header after the data: float 2^-sw
//
// Workgroup = 4 waves x RG row groups of 16 rows (64 RG rows).  The B fragments of one (k chunk, column chunk)
// step -- CT tiles x 2 KB, contiguous in the packed layout -- are staged once per workgroup through a
// double-buffered LDS ring (global -> registers during the MFMAs of the previous step -> ds_write), so the L2
// sees each weight byte once per 64 RG rows instead of once per 16.
// VE: batched epilogue loads, used when the epilogue has a position table or a residual / LayerNorm (measured on an
// MI355X, 129600 rows: 128 -> 384 + table 154 -> 128 us, 128 -> 128 + residual + LN 85 -> 54 us; the plain and GELU
// epilogues are 5-18 % faster WITHOUT it and stay on the default form; profiles/r02_call1_knockout_variants.txt).  In the generated code of the default epilogue every
// bias / table / residual / gamma / beta element is its own `global_load_dword ; s_waitcnt vmcnt(0)` pair (the loads
// sit behind per-element null / bounds branches): 300-450 full waits per kernel, tens of dependent round trips per
// row group.  With VE the loads of a whole group of column tiles are issued unconditionally from clamped, always
// valid addresses (a null operand reads x[0]; the value is discarded by a select, never multiplied) and consumed
// afterwards; the arithmetic and its order are unchanged.
template <int KC /* K/32 */, int CT /* column tiles per chunk */, int RG /* row groups per wave */,
          bool CF /* some operand is channels-first */, bool VE = false>
__global__ __launch_bounds__(256, (KC == 8 && CT == 16) ? 1 : 2) void linear_f16x3_kernel(const float* __restrict__ x, int M, int ldx,
                                                           const uint4* __restrict__ wp,
                                                           const float* __restrict__ w_inv_scale, int N,
                                                           LinearEpilogue ep, float* __restrict__ y, int ldy,
                                                           int csplit) {
  constexpr int STEP_U4 = CT * 128;          // uint4 per staged step
  constexpr int PF = STEP_U4 / 256;          // uint4 per thread per step
  __shared__ uint4 bbuf[2][STEP_U4];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int col = lane & 15, kg = lane >> 4;
  const int ntiles = N >> 4;
  const int nchunks = (ntiles + CT - 1) / CT;
  // COLUMN SPLIT (csplit; launches with few row blocks and several column chunks -- the level-1 SST projections, 16 200 rows
  // x 768 / 1024 columns: 254 workgroups = one wave per SIMD): one workgroup per (row block, column chunk).  Workgroups are
  // dealt to the 8 XCDs round-robin, so the chunks of a row block take ids 8 apart: they run on ONE XCD, whose L2 then
  // serves the block's input rows to all of them.
  int rb = blockIdx.x, cbeg = 0, cend = nchunks;
  if (csplit) {
    const int g = blockIdx.x / (8 * nchunks), within = blockIdx.x - g * 8 * nchunks;
    rb = g * 8 + (within & 7);
    cbeg = within >> 3;
    cend = cbeg + 1;
    if (rb * 64 * RG >= M) return;
  }
  const int r0 = (rb * 4 + wave) * 16 * RG;
  // channels-first input: the workgroup's [K x 64 RG rows] tile is loaded with 16-byte loads along the rows (the
  // contiguous direction of [B, K, hw]) into LDS and read back transposed
  constexpr int XT_LD = 64 * RG + 4;
  extern __shared__ __attribute__((aligned(16))) float xt[];   // [K][XT_LD], only when ep.x_hw
  if (CF && ep.x_hw) {
    constexpr int QUADS = 16 * RG;   // row quads per tile
    const int R0 = rb * 64 * RG;
    for (int idx = threadIdx.x; idx < KC * 32 * QUADS; idx += 256) {
      const int c = idx / QUADS, q = idx - c * QUADS;
      const int r = R0 + 4 * q;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < M) {
        const int bb = r / ep.x_hw, pos = r - bb * ep.x_hw;
        const float* p = x + ((size_t)bb * (KC * 32) + c) * ep.x_hw + pos;
        if (r + 3 < M) v = *reinterpret_cast<const float4*>(p);
        else { v.x = p[0]; if (r + 1 < M) v.y = p[1]; if (r + 2 < M) v.z = p[2]; }
      }
      *reinterpret_cast<float4*>(xt + c * XT_LD + 4 * q) = v;
    }
    __syncthreads();
  }
  // A fragments: row r0 + 16 g + col, channels 32 kc + 8 kg .. +8
  h8 ah[RG][KC], al[RG][KC];
#pragma unroll
  for (int g = 0; g < RG; ++g) {
    const int r = r0 + 16 * g + col;
    const int rc = r < M ? r : M - 1;
    if (!CF || ep.x_hw == 0) {
      const float* xr = x + (size_t)rc * ldx + kg * 8;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        f32x8 v = *reinterpret_cast<const f32x8*>(xr + kc * 32);
        if (r >= M) v = f32x8{0, 0, 0, 0, 0, 0, 0, 0};
        lin_split8(v, ah[g][kc], al[g][kc]);
      }
    } else {   // [B, K, hw] tile staged through LDS (see below)
      const int rl = (wave * RG + g) * 16 + col;   // row inside the workgroup tile
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        f32x8 v;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) v[jj] = xt[(kc * 32 + kg * 8 + jj) * XT_LD + rl];
        lin_split8(v, ah[g][kc], al[g][kc]);
      }
    }
  }
  const float winv = *w_inv_scale;
  // prefetch registers for the next step
  uint4 pf[PF];
  auto fetch = [&](int chunk, int kc) {
    const int c0 = chunk * CT;
    const int valid_u4 = (ntiles - c0 < CT ? ntiles - c0 : CT) * 128;
    const uint4* src = wp + ((size_t)kc * ntiles + c0) * 128;
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int idx = threadIdx.x + 256 * i;
      pf[i] = idx < valid_u4 ? src[idx] : make_uint4(0, 0, 0, 0);
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PF; ++i) bbuf[buf][threadIdx.x + 256 * i] = pf[i];
  };
  fetch(cbeg, 0);
  commit(0);
  __syncthreads();
  int step = 0;
  for (int chunk = cbeg; chunk < cend; ++chunk) {
    const int c0 = chunk * CT;
    f32x4 acc[RG][CT];
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
      for (int nt = 0; nt < CT; ++nt) acc[g][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < KC; ++kc, ++step) {
      const int buf = step & 1;
      const bool more = !(chunk == cend - 1 && kc == KC - 1);
      if (more) fetch(kc == KC - 1 ? chunk + 1 : chunk, kc == KC - 1 ? 0 : kc + 1);
#pragma unroll
      for (int nt = 0; nt < CT; ++nt) {
        const uint4 bhu = bbuf[buf][nt * 128 + lane];
        const uint4 blu = bbuf[buf][nt * 128 + 64 + lane];
        const h8 bh = *reinterpret_cast<const h8*>(&bhu);
        const h8 bl = *reinterpret_cast<const h8*>(&blu);
#pragma unroll
        for (int g = 0; g < RG; ++g) {
          acc[g][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[g][kc], bh, acc[g][nt], 0, 0, 0);
          acc[g][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[g][kc], bl, acc[g][nt], 0, 0, 0);
          acc[g][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[g][kc], bh, acc[g][nt], 0, 0, 0);
        }
      }
      if (more) commit(buf ^ 1);   // the other buffer was last read before the previous barrier
      __syncthreads();
    }
    // epilogue on the C/D layout: this lane holds rows r0 + 16 g + 4 kg + t (t = 0..3), column 16 (c0+nt) + col
#pragma unroll
    for (int g = 0; g < RG; ++g) {
      float v[CT][4];
      if constexpr (VE) {
        constexpr int NB = CT < 4 ? CT : 4;                 // column tiles whose operands are fetched together
        const int rq = r0 + 16 * g + 4 * kg;
        const bool has_b = ep.bias != nullptr, has_t = ep.table != nullptr, has_r = ep.residual != nullptr;
        const bool res_cf = CF && ep.res_hw;
        const float* bp = has_b ? ep.bias : x;
        const float* tp = has_t ? ep.table : x;
        const float* rp = has_r ? ep.residual : x;
        int rr[4], ti[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) rr[t] = min(rq + t, M - 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) ti[t] = has_t ? ep.idx[rr[t]] : 0;
#pragma unroll
        for (int nt0 = 0; nt0 < CT; nt0 += NB) {
          float bb_[NB], tb[NB][4], rs[NB][4];
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            const int nt = nt0 + j;
            const bool cok = c0 + nt < ntiles;
            const int n = cok ? (c0 + nt) * 16 + col : 0;
            bb_[j] = bp[has_b ? n : 0];
#pragma unroll
            for (int t = 0; t < 4; ++t) tb[j][t] = tp[has_t ? (size_t)ti[t] * N + n : 0];
            if (res_cf) {
              const int r0c = min(rq, M - 4 > 0 ? M - 4 : 0);      // hw % 4 == 0: rq + 3 < M whenever rq < M
              const int bq = r0c / ep.res_hw, pos = r0c - bq * ep.res_hw;
              const float4 q = *reinterpret_cast<const float4*>(rp + ((size_t)bq * N + n) * ep.res_hw + pos);
              rs[j][0] = q.x; rs[j][1] = q.y; rs[j][2] = q.z; rs[j][3] = q.w;
            } else {
#pragma unroll
              for (int t = 0; t < 4; ++t) rs[j][t] = rp[has_r ? (size_t)rr[t] * N + n : 0];
            }
          }
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            const int nt = nt0 + j;
            const bool cok = c0 + nt < ntiles;
            const float b = (has_b && cok) ? bb_[j] : 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              float z = acc[g][nt][t] * winv + b;
              if (cok && rq + t < M) {
                if (has_t) z += tb[j][t];
                if (ep.act == 1) z = fmaxf(z, 0.f);
                else if (ep.act == 2) z = 0.5f * z * (1.f + erff(z * 0.70710678118654752440f));
                if (has_r) z += rs[j][t];
              } else {
                z = 0.f;
              }
              v[nt][t] = z;
            }
          }
        }
      } else {
#pragma unroll
      for (int nt = 0; nt < CT; ++nt) {
        const int n = (c0 + nt) * 16 + col;
        const bool cok = c0 + nt < ntiles;
        const float b = (ep.bias && cok) ? ep.bias[n] : 0.f;
        const int rq = r0 + 16 * g + 4 * kg;   // 4 consecutive rows; hw % 4 == 0 keeps them in one sample
        float4 res4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (CF && ep.residual && ep.res_hw && cok && rq < M) {
          const int bb = rq / ep.res_hw, pos = rq - bb * ep.res_hw;
          const float* rp = ep.residual + ((size_t)bb * N + n) * ep.res_hw + pos;
          if (rq + 3 < M) res4 = *reinterpret_cast<const float4*>(rp);
          else { res4.x = rp[0]; if (rq + 1 < M) res4.y = rp[1]; if (rq + 2 < M) res4.z = rp[2]; }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int r = rq + t;
          float z = acc[g][nt][t] * winv + b;
          if (cok && r < M) {
            if (ep.table) z += ep.table[(size_t)ep.idx[r] * N + n];
            if (ep.act == 1) z = fmaxf(z, 0.f);
            else if (ep.act == 2) z = 0.5f * z * (1.f + erff(z * 0.70710678118654752440f));
            if (ep.residual) z += (CF && ep.res_hw) ? (t == 0 ? res4.x : t == 1 ? res4.y : t == 2 ? res4.z : res4.w)
                                            : ep.residual[(size_t)r * N + n];
          } else {
            z = 0.f;
          }
          v[nt][t] = z;
        }
      }
      }
      if (VE && ep.ln_gamma) {   // gamma / beta of every column tile fetched once, then the same arithmetic
        float gm[CT], bt[CT];
#pragma unroll
        for (int nt = 0; nt < CT; ++nt) {
          const int n = (c0 + nt < ntiles) ? (c0 + nt) * 16 + col : 0;
          gm[nt] = ep.ln_gamma[n];
          bt[nt] = ep.ln_beta[n];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float s = 0.f;
#pragma unroll
          for (int nt = 0; nt < CT; ++nt) s += v[nt][t];
#pragma unroll
          for (int d = 1; d < 16; d <<= 1) s += __shfl_xor(s, d, 64);
          const float mean = s / (float)N;
          float q = 0.f;
#pragma unroll
          for (int nt = 0; nt < CT; ++nt) {
            const float dlt = (c0 + nt < ntiles) ? v[nt][t] - mean : 0.f;
            q += dlt * dlt;
          }
#pragma unroll
          for (int d = 1; d < 16; d <<= 1) q += __shfl_xor(q, d, 64);
          const float rstd = rsqrtf(q / (float)N + ep.ln_eps);
#pragma unroll
          for (int nt = 0; nt < CT; ++nt)
            if (c0 + nt < ntiles) v[nt][t] = (v[nt][t] - mean) * rstd * gm[nt] + bt[nt];
        }
      } else if (ep.ln_gamma) {  // whole row in this chunk (host guarantees ntiles <= CT)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float s = 0.f;
#pragma unroll
          for (int nt = 0; nt < CT; ++nt) s += v[nt][t];
#pragma unroll
          for (int d = 1; d < 16; d <<= 1) s += __shfl_xor(s, d, 64);
          const float mean = s / (float)N;
          float q = 0.f;
#pragma unroll
          for (int nt = 0; nt < CT; ++nt) {
            const float dlt = (c0 + nt < ntiles) ? v[nt][t] - mean : 0.f;
            q += dlt * dlt;
          }
#pragma unroll
          for (int d = 1; d < 16; d <<= 1) q += __shfl_xor(q, d, 64);
          const float rstd = rsqrtf(q / (float)N + ep.ln_eps);
#pragma unroll
          for (int nt = 0; nt < CT; ++nt) {
            if (c0 + nt < ntiles) {
              const int n = (c0 + nt) * 16 + col;
              v[nt][t] = (v[nt][t] - mean) * rstd * ep.ln_gamma[n] + ep.ln_beta[n];
            }
          }
        }
      }
#pragma unroll
      for (int nt = 0; nt < CT; ++nt) {
        if (c0 + nt < ntiles) {
          const int n = (c0 + nt) * 16 + col;
          const int rq = r0 + 16 * g + 4 * kg;
          if (CF && ep.y_hw) {   // [B, N, hw]: this lane's 4 rows are 4 consecutive floats of channel n
            if (rq < M) {
              const int bb = rq / ep.y_hw, pos = rq - bb * ep.y_hw;
              float* yp = y + ((size_t)bb * N + n) * ep.y_hw + pos;
              if (rq + 3 < M) *reinterpret_cast<float4*>(yp) = make_float4(v[nt][0], v[nt][1], v[nt][2], v[nt][3]);
              else { yp[0] = v[nt][0]; if (rq + 1 < M) yp[1] = v[nt][1]; if (rq + 2 < M) yp[2] = v[nt][2]; }
            }
          } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int r = rq + t;
              if (r < M) y[(size_t)r * ldy + n] = v[nt][t];
            }
          }
        }
      }
    }
  }
}

// transposed: `w` is [K][N] row-major, the transpose of the weight that is packed (the forward weight when the packed one
// is the data gradient's: dX = dY W)
__global__ void pack_linear_kernel(const float* __restrict__ w, int N, int K, const unsigned* __restrict__ amax_bits,
                                   uint4* __restrict__ packed, float* __restrict__ header, int transposed) {
  const float amax = __uint_as_float(*amax_bits);
  int e = 0;
  if (amax > 0.f) (void)frexpf(amax, &e);
  const int sw = amax > 0.f ? 13 - e : 0;
  const float s = ldexpf(1.f, sw);
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) header[0] = ldexpf(1.f, -sw);
  const int ntiles = N >> 4, kcs = K >> 5;
  if (t >= (long long)kcs * ntiles * 64) return;
  const int lane = (int)(t & 63);
  const int nt = (int)((t >> 6) % ntiles);
  const int kc = (int)((t >> 6) / ntiles);
  f32x8 v;
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    const int n = 16 * nt + (lane & 15), k = 32 * kc + 8 * (lane >> 4) + jj;
    v[jj] = (transposed ? w[(size_t)k * N + n] : w[(size_t)n * K + k]) * s;
  }
  h8 hi, lo;
  lin_split8(v, hi, lo);
  const size_t base = ((size_t)kc * ntiles + nt) * 128;
  packed[base + lane] = *reinterpret_cast<const uint4*>(&hi);
  packed[base + 64 + lane] = *reinterpret_cast<const uint4*>(&lo);
}

__global__ void absmax_kernel2(const float* __restrict__ w, size_t n, unsigned* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float m = 0.f;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

template <int KC>
static int launch_linear(const float* x, int M, int ldx, const void* packed, int N, int K, const LinearEpilogue& ep,
                         float* y, int ldy, hipStream_t st) {
  const uint4* wp = reinterpret_cast<const uint4*>(packed);
  const float* winv = reinterpret_cast<const float*>(reinterpret_cast<const char*>(packed) + (size_t)N * K * 4);
  const dim3 block(256);
  const int ntiles = N / 16;
  const bool wide_ln = ep.ln_gamma && ntiles > 8;   // LayerNorm needs the whole row in one column chunk
  const bool cf = ep.x_hw || ep.res_hw || ep.y_hw;
  static bool attr_set = false, attr_set_ve = false;
  if (!attr_set) {   // channels-first input stages a [K][rows + 4] fp32 tile in dynamic LDS on top of the weight ring
    ISF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_f16x3_kernel<KC, 16, 1, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, KC * 32 * 68 * 4));
    ISF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_f16x3_kernel<KC, 8, 1, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, KC * 32 * 68 * 4));
    attr_set = true;
  }
  const size_t lds1 = ep.x_hw ? (size_t)KC * 32 * 68 * 4 : 0;
  const bool vepi = ep.table || ep.residual || ep.ln_gamma;   // batched epilogue loads (see linear_f16x3_kernel)
  if (vepi && !attr_set_ve) {
    ISF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_f16x3_kernel<KC, 16, 1, true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, KC * 32 * 68 * 4));
    ISF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_f16x3_kernel<KC, 8, 1, true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, KC * 32 * 68 * 4));
    attr_set_ve = true;
  }
  // (the batched epilogue exists for one row group per wave only: with two it spills)
  // (Small launches -- the 400-row Linears of the instance branch and the head, 27 of a full forward's 47 -- take 9-11 us on
  // this kernel, ~4 us above an empty launch.  A variant that requests all of K at once (whole-K weights in LDS, one wait, no
  // barrier in the loop; bit-identical) was built in round 6 and measured SLOWER, 15.6 us for 64 KiB of weights and 29 us
  // for 128 KiB whether they came by LDS-DMA or through registers: its time follows the size of its straight-line code, i.e.
  // instruction fetch of a once-executed unrolled body, not data.  Removed; gpurun_out/linear_census3/4.txt.)
  // column split (see the kernel): several column chunks and fewer than four workgroups per CU without it; row-major input
  // only -- a channels-first input tile is staged through LDS by every workgroup that reads it (16 200 x 256 -> 1024 from
  // [B, C, hw]: 84 us whole rows, 130 us split, gpurun_out/linear_census2.txt)
#define ISF_LIN(CT_, RG_, CF_, ROWS_, LDS_)                                                                             \
  do {                                                                                                                  \
    const int rows_ = vepi ? 64 : ROWS_;                                                                                \
    const int nrb_ = ceil_div(M, rows_), nch_ = ceil_div(ntiles, CT_);                                                  \
    const int cs_ = (nch_ > 1 && nrb_ < 1024 && ep.x_hw == 0) ? 1 : 0;                                                  \
    const dim3 grid_(cs_ ? ceil_div(nrb_, 8) * 8 * nch_ : nrb_);                                                        \
    if (vepi)                                                                                                           \
      hipLaunchKernelGGL((linear_f16x3_kernel<KC, CT_, 1, CF_, true>), grid_, block, LDS_, st, x, M, ldx,               \
                         wp, winv, N, ep, y, ldy, cs_);                                                                 \
    else                                                                                                                \
      hipLaunchKernelGGL((linear_f16x3_kernel<KC, CT_, RG_, CF_>), grid_, block, LDS_, st, x, M, ldx,                   \
                         wp, winv, N, ep, y, ldy, cs_);                                                                 \
  } while (0)
  if (wide_ln) {
    if (cf) ISF_LIN(16, 1, true, 64, lds1); else ISF_LIN(16, 1, false, 64, 0);
  } else if (M <= 4096 || KC == 8) {   // few rows: more, smaller workgroups; K = 256: A fragments fill the registers
    if (cf) ISF_LIN(8, 1, true, 64, lds1); else ISF_LIN(8, 1, false, 64, 0);
  } else {
    if (cf) ISF_LIN(8, 1, true, 64, lds1); else ISF_LIN(8, 2, false, 128, 0);   // CF + 2 row groups spills
  }
#undef ISF_LIN
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

}  // namespace isf

extern "C" {

size_t isf_packed_linear_bytes(int out_features, int in_features) {
  return (size_t)out_features * in_features * 4 + 64;
}

static int pack_linear_impl(const float* weight, int out_features, int in_features, void* packed, int transposed,
                            isf_stream_t stream);

int isf_pack_linear(const float* weight, int out_features, int in_features, void* packed, isf_stream_t stream) {
  return pack_linear_impl(weight, out_features, in_features, packed, 0, stream);
}

int isf_pack_linear_transposed(const float* weight_t, int out_features, int in_features, void* packed, isf_stream_t stream) {
  return pack_linear_impl(weight_t, out_features, in_features, packed, 1, stream);
}

static int pack_linear_impl(const float* weight, int out_features, int in_features, void* packed, int transposed,
                            isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(weight && packed && out_features % 16 == 0 && in_features % 32 == 0 && out_features > 0 && in_features > 0,
              ISF_ERR_ARG, "pack_linear: need out %% 16 == 0 and in %% 32 == 0 (got %d, %d)", out_features, in_features);
  hipStream_t st = as_stream(stream);
  Arena& a = arena_for_stream(as_stream(stream));
  ISF_TRY(a.reset());
  unsigned* amax = nullptr;
  ISF_TRY(a.alloc_n(&amax, 64));
  ISF_HIP_TRY(hipMemsetAsync(amax, 0, sizeof(unsigned), st));
  const size_t n = (size_t)out_features * in_features;
  hipLaunchKernelGGL(absmax_kernel2, dim3(ceil_div((long long)n, 1024) < 256 ? ceil_div((long long)n, 1024) : 256),
                     dim3(256), 0, st, weight, n, amax);
  const long long total = (long long)(in_features / 32) * (out_features / 16) * 64;
  hipLaunchKernelGGL(pack_linear_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, weight, out_features,
                     in_features, amax, reinterpret_cast<uint4*>(packed),
                     reinterpret_cast<float*>(reinterpret_cast<char*>(packed) + n * 4), transposed);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int isf_linear_forward(const float* x, int num_rows, int in_features, int ldx, const void* packed_weight,
                       int out_features, const float* bias, const float* row_table, const int32_t* row_table_index,
                       int activation, const float* residual, const float* ln_gamma, const float* ln_beta,
                       float ln_eps, float* y, int ldy, int x_hw, int residual_hw, int y_hw, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(num_rows >= 0 && in_features > 0 && out_features > 0, ISF_ERR_ARG, "linear_forward: bad sizes");
  if (num_rows == 0) return ISF_OK;
  ISF_REQUIRE(x && packed_weight && y, ISF_ERR_ARG, "linear_forward: null pointer");
  ISF_REQUIRE(in_features % 32 == 0 && in_features <= 256 && out_features % 16 == 0, ISF_ERR_UNSUPPORTED,
              "linear_forward: in_features %d (need %%32, <= 256), out_features %d (need %%16)", in_features,
              out_features);
  ISF_REQUIRE((x_hw || (ldx % 8 == 0 && ldx >= in_features)) && (y_hw || ldy >= out_features), ISF_ERR_ARG,
              "linear_forward: bad strides");
  ISF_REQUIRE(x_hw >= 0 && residual_hw >= 0 && y_hw >= 0 && x_hw % 4 == 0 && residual_hw % 4 == 0 && y_hw % 4 == 0,
              ISF_ERR_UNSUPPORTED, "linear_forward: channels-first tensors need hw %% 4 == 0");
  ISF_REQUIRE((!x_hw || num_rows % x_hw == 0) && (!residual_hw || num_rows % residual_hw == 0) &&
                  (!y_hw || num_rows % y_hw == 0), ISF_ERR_ARG, "linear_forward: num_rows is not a multiple of hw");
  ISF_REQUIRE(!ln_gamma || (out_features <= 256 && ln_beta), ISF_ERR_UNSUPPORTED,
              "linear_forward: LayerNorm epilogue needs out_features <= 256");
  ISF_REQUIRE((row_table == nullptr) == (row_table_index == nullptr), ISF_ERR_ARG, "linear_forward: table/index");
  LinearEpilogue ep{bias, row_table, row_table_index, residual, ln_gamma, ln_beta, ln_eps, activation,
                    x_hw, residual_hw, y_hw};
  hipStream_t st = as_stream(stream);
  switch (in_features / 32) {
    case 1: return launch_linear<1>(x, num_rows, ldx, packed_weight, out_features, in_features, ep, y, ldy, st);
    case 2: return launch_linear<2>(x, num_rows, ldx, packed_weight, out_features, in_features, ep, y, ldy, st);
    case 4: return launch_linear<4>(x, num_rows, ldx, packed_weight, out_features, in_features, ep, y, ldy, st);
    case 8: return launch_linear<8>(x, num_rows, ldx, packed_weight, out_features, in_features, ep, y, ldy, st);
  }
  set_error("linear_forward: in_features %d not built (32, 64, 128, 256)", in_features);
  return ISF_ERR_UNSUPPORTED;
}

}  // extern "C"
