// isf_spconv_bwd.hip -- SURVEY 8f #2: sparse convolution backward.
//
// Reference (spconv_ops.h:363-456, indice_conv_backward): per tap k: gather x rows and dy rows into two HBM buffers,
// two cuBLAS GEMMs (dW[k] = X_k^T dY_k, dX_k = dY_k W[k]^T), scatter-add dX_k -- ~110 launches per conv, fp32
// atomics-free only because the taps run one after the other.
//
// Here, on the output-stationary neighbour table of the forward pass:
//   dX   the forward kernel itself.  With nbr_t[k][j] = i  <=>  nbr[k][i] = j (a tap maps an input row to at most one
//        output row, so the inverse is a plain scatter without atomics) the input gradient is
//        dX[j] = sum_k dY[nbr_t[k][j]] @ W[k]^T -- an output-stationary sparse conv over the transposed table with
//        transposed filters: every dX row is written once, fixed summation order, no atomics.
//   dW   dW[k] = sum_o x[nbr[k][o]]^T dY[o]: a [Cin x rows] x [rows x Cout] GEMM per tap whose long dimension is the
//        row index.  One wave owns a 64 x 64 block of dW[k] for a chunk of rows: per 4 rows it issues ONE 16-byte
//        load per lane for x (4 channels of one gathered row) and one for dY, which feed 16
//        v_mfma_f32_16x16x4_f32 (4 channel-interleaved M tiles x 4 N tiles, fp32 in / fp32 accumulate, so the
//        result matches an fp32 GEMM); 4-row groups without a neighbour through the tap are skipped wave-uniformly.
//        Row chunks write partial blocks, a second pass adds them in chunk order (deterministic, no float atomics).
#include <algorithm>

#include "isf_common.h"

namespace isf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void transpose_nbr_kernel(const int32_t* __restrict__ nbr, int nbr_stride, int n_out, int K,
                                     int32_t* __restrict__ nbr_t, int t_stride, int n_in) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)K * n_out) return;
  const int k = (int)(idx / n_out), o = (int)(idx % n_out);
  const int j = nbr[(size_t)k * nbr_stride + o];
  if (j >= 0 && j < n_in) nbr_t[(size_t)k * t_stride + j] = o;
}

__global__ void transpose_filters_kernel(const float* __restrict__ w, int K, int cin, int cout,
                                         float* __restrict__ wt) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)K * cin * cout) return;
  const int ci = (int)(idx % cin);
  const int co = (int)((idx / cin) % cout);
  const int k = (int)(idx / ((long long)cin * cout));
  wt[idx] = w[((size_t)k * cin + ci) * cout + co];   // wt [K][cout][cin]
}

// (Round 4's compacted-rows probe of this kernel ran in round 5: 2.1x, 34 TFLOP/s -- profiles/r05_wgrad_compact_probe.txt --
// and was superseded by the f16 matrix-core kernel of isf_spconv_wgrad16.hip: 189 TFLOP/s.  This fp32-MFMA kernel stays as
// the fp32 reference entry the new one is tested against.)
// one wave per (64 x 64 block of dW[k], tap, row chunk)
__global__ __launch_bounds__(64) void wgrad_mfma_kernel(const float* __restrict__ x, int cin,
                                                        const float* __restrict__ dy, int cout,
                                                        const int32_t* __restrict__ nbr, int nbr_stride, int n_out,
                                                        int rows_per_chunk, int K, float* __restrict__ partial) {
  const int lane = threadIdx.x, sub = lane & 15, kslot = lane >> 4;
  const int co_blocks = (cout + 63) >> 6;
  const int ci_base = (blockIdx.x / co_blocks) * 64, co_base = (blockIdx.x % co_blocks) * 64;
  const int k = blockIdx.y, chunk = blockIdx.z;
  const bool a_ok = ci_base + 4 * sub < cin, b_ok = co_base + 4 * sub < cout;
  f32x4 acc[4][4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int r_begin = chunk * rows_per_chunk, r_end = min(n_out, r_begin + rows_per_chunk);
  const int32_t* nk = nbr + (size_t)k * nbr_stride;
  // Software pipeline (round 4): the neighbour index of a 4-row group is fetched TWO groups ahead and its x / dy rows ONE
  // group ahead, so the 16 MFMAs of a group run while the next group's 32 bytes per lane are in flight (the plain loop
  // paid index latency + row latency in front of every 16 MFMAs: 3.6 % of the fp32 matrix rate over a training step).
  // Same products in the same order per accumulator: results are bit-identical.
  auto idx_of = [&](int r0) -> int {
    const int row = r0 + kslot;
    return (r0 < r_end && row < r_end) ? nk[row] : -1;
  };
  auto load_rows = [&](int r0, int in, f32x4& a, f32x4& b) {
    a = f32x4{0.f, 0.f, 0.f, 0.f};
    b = f32x4{0.f, 0.f, 0.f, 0.f};
    if (in >= 0) {
      if (a_ok) a = *reinterpret_cast<const f32x4*>(x + (size_t)in * cin + ci_base + 4 * sub);
      if (b_ok) b = *reinterpret_cast<const f32x4*>(dy + (size_t)(r0 + kslot) * cout + co_base + 4 * sub);
    }
  };
  int in_cur = idx_of(r_begin), in_nxt = idx_of(r_begin + 4);
  f32x4 a_cur, b_cur, a_nxt, b_nxt;
  load_rows(r_begin, in_cur, a_cur, b_cur);
  for (int r0 = r_begin; r0 < r_end; r0 += 4) {
    const int in_nn = idx_of(r0 + 8);                       // index two groups ahead
    load_rows(r0 + 4, in_nxt, a_nxt, b_nxt);                // rows one group ahead (in_nxt arrived a group ago)
    if (__ballot(in_cur >= 0) != 0ull) {
      // A[m = sub][k = kslot] of M tile s = x[in(row kslot)][ci_base + 4*sub + s]; B[k = kslot][n = sub] of N tile t =
      // dy[row kslot][co_base + 4*sub + t]
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[s], b_cur[t], acc[s][t], 0, 0, 0);
    }
    in_cur = in_nxt;
    in_nxt = in_nn;
    a_cur = a_nxt;
    b_cur = b_nxt;
  }
  // C/D layout: acc[s][t][r] = C[m = 4*kslot + r][n = sub]  ->  dW[ci_base + 4*m + s][co_base + 4*n + t]
  float* out = partial + ((size_t)chunk * K + k) * cin * cout;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ci = ci_base + 4 * (4 * kslot + r) + s;
      if (ci >= cin) continue;
      const int co = co_base + 4 * sub;
      if (co >= cout) continue;
      *reinterpret_cast<f32x4*>(out + (size_t)ci * cout + co) =
          f32x4{acc[s][0][r], acc[s][1][r], acc[s][2][r], acc[s][3][r]};
    }
}


// any channel count (not a multiple of 4): one thread per dW element, rows in order
__global__ void wgrad_generic_kernel(const float* __restrict__ x, int cin, const float* __restrict__ dy, int cout,
                                     const int32_t* __restrict__ nbr, int nbr_stride, int n_out, int K,
                                     float* __restrict__ dw) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)K * cin * cout) return;
  const int co = (int)(idx % cout);
  const int ci = (int)((idx / cout) % cin);
  const int k = (int)(idx / ((long long)cin * cout));
  const int32_t* nk = nbr + (size_t)k * nbr_stride;
  float acc = 0.f;
  for (int o = 0; o < n_out; ++o) {
    const int in = nk[o];
    if (in >= 0) acc += x[(size_t)in * cin + ci] * dy[(size_t)o * cout + co];
  }
  dw[idx] = acc;
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, int chunks, size_t elems,
                                    float* __restrict__ dw) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= elems) return;
  float acc = 0.f;
  for (int c = 0; c < chunks; ++c) acc += partial[(size_t)c * elems + e];
  dw[e] = acc;
}

}  // namespace isf

extern "C" {

int isf_transpose_rulebook(const int32_t* nbr, int nbr_stride, int num_out, int num_taps, int num_in,
                           int32_t* nbr_t, int nbr_t_stride, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(num_out >= 0 && num_in >= 0 && num_taps > 0 && nbr_t_stride >= num_in && nbr_t_stride % 128 == 0,
              ISF_ERR_ARG, "transpose_rulebook: bad sizes (nbr_t_stride must be isf_nbr_stride(num_in))");
  if (num_in == 0) return ISF_OK;
  ISF_REQUIRE(nbr_t && (nbr || num_out == 0), ISF_ERR_ARG, "transpose_rulebook: null pointer");
  hipStream_t st = as_stream(stream);
  ISF_HIP_TRY(hipMemsetAsync(nbr_t, 0xFF, sizeof(int32_t) * (size_t)num_taps * nbr_t_stride, st));
  if (num_out == 0) return ISF_OK;
  hipLaunchKernelGGL(transpose_nbr_kernel, dim3(ceil_div((long long)num_taps * num_out, 256)), dim3(256), 0, st, nbr,
                     nbr_stride, num_out, num_taps, nbr_t, nbr_t_stride, num_in);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int isf_sparse_conv_backward_input(const float* grad_out, int num_out, int c_out, const float* filters, int num_taps,
                                   int c_in, const int32_t* nbr_t, int nbr_t_stride, int num_in, float* grad_in,
                                   isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(num_in >= 0 && num_out >= 0 && c_in > 0 && c_out > 0 && num_taps > 0, ISF_ERR_ARG,
              "sparse_conv_backward_input: bad arguments");
  if (num_in == 0) return ISF_OK;
  ISF_REQUIRE(filters && nbr_t && grad_in && (grad_out || num_out == 0), ISF_ERR_ARG,
              "sparse_conv_backward_input: null pointer");
  hipStream_t st = as_stream(stream);
  if (num_out == 0) {
    ISF_HIP_TRY(hipMemsetAsync(grad_in, 0, sizeof(float) * (size_t)num_in * c_in, st));
    return ISF_OK;
  }
  Arena& a = arena_for_stream(as_stream(stream));
  ISF_TRY(a.reset());
  const size_t elems = (size_t)num_taps * c_in * c_out;
  float *wt = nullptr, *packed = nullptr;
  ISF_TRY(a.alloc_n(&wt, elems));
  hipLaunchKernelGGL(transpose_filters_kernel, dim3(ceil_div((long long)elems, 256)), dim3(256), 0, st, filters,
                     num_taps, c_in, c_out, wt);
  ISF_LAUNCH_CHECK();
  // a sparse conv from the Cout-channel gradient rows to the Cin-channel input rows over the transposed table
  if (!sparse_conv_mfma_supported(c_out, c_in))
    return sparse_conv_forward_generic_impl(grad_out, c_out, wt, num_taps, c_in, nbr_t, nbr_t_stride, num_in, nullptr,
                                            nullptr, nullptr, 0, grad_in, st);
  ISF_TRY(a.alloc_n(&packed, elems));
  ISF_TRY(pack_filters_impl(wt, num_taps, c_out, c_in, packed, st));
  return sparse_conv_forward_packed_impl(grad_out, num_out, c_out, packed, num_taps, c_in, nbr_t, nbr_t_stride, num_in,
                                         nullptr, nullptr, nullptr, 0, grad_in, st);
}

int isf_sparse_conv_backward_filter(const float* features, int num_in, int c_in, const float* grad_out, int num_out,
                                    int c_out, const int32_t* nbr, int nbr_stride, int num_taps,
                                    float* grad_filters, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(num_in >= 0 && num_out >= 0 && c_in > 0 && c_out > 0 && num_taps > 0, ISF_ERR_ARG,
              "sparse_conv_backward_filter: bad arguments");
  ISF_REQUIRE(grad_filters, ISF_ERR_ARG, "sparse_conv_backward_filter: null pointer");
  hipStream_t st = as_stream(stream);
  const size_t elems = (size_t)num_taps * c_in * c_out;
  if (num_out == 0 || num_in == 0) {
    ISF_HIP_TRY(hipMemsetAsync(grad_filters, 0, sizeof(float) * elems, st));
    return ISF_OK;
  }
  ISF_REQUIRE(features && grad_out && nbr, ISF_ERR_ARG, "sparse_conv_backward_filter: null pointer");
  if (c_in % 4 != 0 || c_out % 4 != 0) {
    hipLaunchKernelGGL(wgrad_generic_kernel, dim3(ceil_div((long long)elems, 128)), dim3(128), 0, st, features, c_in,
                       grad_out, c_out, nbr, nbr_stride, num_out, num_taps, grad_filters);
    ISF_LAUNCH_CHECK();
    return ISF_OK;
  }
  const int blocks = ceil_div(c_in, 64) * ceil_div(c_out, 64);
  // enough waves to fill the chip (256 CUs x 8+ waves), at least 256 rows per chunk
  int chunks = std::max(1, 4096 / (blocks * num_taps));
  chunks = std::min(chunks, ceil_div(num_out, 256));
  int rows_per_chunk = ceil_div(num_out, chunks);
  rows_per_chunk = (rows_per_chunk + 3) & ~3;
  chunks = ceil_div(num_out, rows_per_chunk);
  float* partial = grad_filters;
  if (chunks > 1) {
    Arena& a = arena_for_stream(as_stream(stream));
    ISF_TRY(a.reset());
    ISF_TRY(a.alloc_n(&partial, elems * chunks));
  }
  hipLaunchKernelGGL(wgrad_mfma_kernel, dim3(blocks, num_taps, chunks), dim3(64), 0, st, features, c_in, grad_out,
                     c_out, nbr, nbr_stride, num_out, rows_per_chunk, num_taps, partial);
  ISF_LAUNCH_CHECK();
  if (chunks > 1) {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(ceil_div((long long)elems, 256)), dim3(256), 0, st, partial, chunks,
                       elems, grad_filters);
    ISF_LAUNCH_CHECK();
  }
  return ISF_OK;
}

}  // extern "C"
