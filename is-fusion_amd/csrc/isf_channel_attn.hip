// isf_channel_attn.hip -- A14 per-channel map attention of Instane2SceneAtt (fusion_encoder.py:497-502):
//   out[m] = Qs[m] + softmax(Qs[m] . Qi[m]^T, dim=-1) . Qi[m]        for each of the B*C maps m (R x R, R = 180)
// The reference runs two batched cuBLAS GEMMs with a [B, C, R, R] score tensor (16.6 MB / sample) written and
// re-read between them.  Here one workgroup owns a map: Qi stays in LDS (padded [192][196] fp32 = 147 KB of the
// 160 KB), each wave takes 16 query rows at a time and keeps the 16 x 192 score tile in MFMA accumulators from the
// first GEMM through the softmax into the second GEMM -- the scores never leave registers:
//   GEMM1 computes S^T (A = Qi rows from LDS, B = Qs rows from global), so that the C/D layout
//     lane (i = lane & 15, kg = lane >> 4), reg (jt, t)  <->  S[i][j = 16 jt + 4 kg + t]
//   is exactly the A-operand layout GEMM2 needs when its k index is enumerated as j = 16 jt + 4 kg + t;
//   GEMM2's B operand is a float4 of Qi[j][64 g + 4 nl .. +4]: element s feeds the s-th of four interleaved
//   column tiles, so each lane ends up with 4 consecutive output columns (one 16-byte store).
// fp32 MFMA (v_mfma_f32_16x16x4_f32): exact fp32 products, no split needed; LDS row stride 196 = 4 (mod 64)
// makes both the row-wise (GEMM1) and the column-wise (GEMM2) ds_read_b128 patterns bank-conflict free.
#include "isf_common.h"

namespace isf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CA_ROWS = 192;   // padded keys
constexpr int CA_LD = 196;     // LDS row stride (floats)

__global__ __launch_bounds__(256, 1) void channel_attention_mfma_kernel(const float* __restrict__ qs,
                                                                        const float* __restrict__ qi, int R,
                                                                        float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float K[];   // [CA_ROWS][CA_LD]
  const size_t mat = (size_t)blockIdx.x * R * R;
  // stage Qi with zero padding (rows >= R and columns >= R must be finite zeros: they meet p = 0 / are not stored)
  for (int i = threadIdx.x; i < CA_ROWS * (CA_LD / 4); i += 256) {
    const int r = i / (CA_LD / 4), c = (i % (CA_LD / 4)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < R && c < R) v = *reinterpret_cast<const float4*>(qi + mat + (size_t)r * R + c);
    *reinterpret_cast<float4*>(K + r * CA_LD + c) = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, kg = lane >> 4;
  constexpr int JT = CA_ROWS / 16;   // 12 key tiles
  for (int i0 = wave * 16; i0 < R; i0 += 64) {
    const int irow = i0 + li < R ? i0 + li : R - 1;
    const float* qrow = qs + mat + (size_t)irow * R;
    // ---- GEMM1: S^T tile [192 keys x 16 queries]
    f32x4 acc[JT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) acc[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int w0 = 0; w0 < R; w0 += 16) {
      const int w = w0 + 4 * kg;
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (w < R) b = *reinterpret_cast<const float4*>(qrow + w);     // R % 4 == 0
#pragma unroll
      for (int jt = 0; jt < JT; ++jt) {
        const float4 a = *reinterpret_cast<const float4*>(K + (16 * jt + li) * CA_LD + w);   // zero beyond R
        acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[jt], 0, 0, 0);
        acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[jt], 0, 0, 0);
        acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[jt], 0, 0, 0);
        acc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[jt], 0, 0, 0);
      }
    }
    // ---- softmax over j for query i = li: regs (jt, t) x lanes kg
    float m = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = 16 * jt + 4 * kg + t;
        if (j >= R) acc[jt][t] = -INFINITY;
        m = fmaxf(m, acc[jt][t]);
      }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[jt][t] = __expf(acc[jt][t] - m);
        sum += acc[jt][t];
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    // ---- GEMM2: O tile [16 queries x 192 columns] = P . Qi, three groups of four interleaved column tiles
    f32x4 o[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int s = 0; s < 4; ++s) o[g][s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float p = acc[jt][t] * inv;
        const float* krow = K + (16 * jt + 4 * kg + t) * CA_LD + 4 * li;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          const float4 b = *reinterpret_cast<const float4*>(krow + 64 * g);
          o[g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, b.x, o[g][0], 0, 0, 0);
          o[g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, b.y, o[g][1], 0, 0, 0);
          o[g][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, b.z, o[g][2], 0, 0, 0);
          o[g][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(p, b.w, o[g][3], 0, 0, 0);
        }
      }
    // ---- epilogue: this lane holds O[i0 + 4 kg + t][64 g + 4 li + s]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = i0 + 4 * kg + t;
      if (i >= R) continue;
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const int w = 64 * g + 4 * li;
        if (w >= R) continue;
        const float4 r = *reinterpret_cast<const float4*>(qs + mat + (size_t)i * R + w);
        *reinterpret_cast<float4*>(out + mat + (size_t)i * R + w) =
            make_float4(r.x + o[g][0][t], r.y + o[g][1][t], r.z + o[g][2][t], r.w + o[g][3][t]);
      }
    }
  }
}

}  // namespace isf

extern "C" int isf_channel_attention_forward(const float* query_scene, const float* query_ins, int num_maps,
                                             int size, float* out, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(num_maps >= 0 && size > 0, ISF_ERR_ARG, "channel_attention: bad sizes");
  if (num_maps == 0) return ISF_OK;
  ISF_REQUIRE(query_scene && query_ins && out, ISF_ERR_ARG, "channel_attention: null pointer");
  ISF_REQUIRE(size % 4 == 0 && size <= CA_ROWS, ISF_ERR_UNSUPPORTED,
              "channel_attention: map size %d (need %%4 == 0 and <= %d)", size, CA_ROWS);
  const size_t lds = (size_t)CA_ROWS * CA_LD * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    ISF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&channel_attention_mfma_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(channel_attention_mfma_kernel, dim3(num_maps), dim3(256), lds, as_stream(stream), query_scene,
                     query_ins, size, out);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}
