// isf_spconv.hip -- A6/A7 sparse convolution forward with fused BN / residual / ReLU epilogue.
//
// Reference (spconv_ops.h:260-361): per tap k: gather rows -> [nHot,Cin] buffer in HBM -> cuBLAS GEMM ->
// [nHot,Cout] buffer in HBM -> scatter-add; ~80 launches per conv, every row crosses HBM 3x per pair,
// BN and ReLU are further elementwise passes.
//
// Here: one launch per conv, OUTPUT-stationary.  A workgroup owns TM consecutive output rows (sorted
// (b,z,y,x), i.e. spatial neighbours) and BN output channels:
//   prologue  per tap k, compact the rows of the tile that really have a neighbour through k into an
//             LDS list (wave ballot) -> the MFMA work is proportional to the number of (in,out) PAIRS,
//             not to 27 * rows (level-0 LiDAR voxels have ~5 of 27 neighbours);
//   main      wave w owns 16*NT output channels.  For every tap and every 16-pair group the A operand
//             (gathered input rows, 64 B per row per instruction) goes global -> VGPR directly in the
//             v_mfma_f32_16x16x4_f32 fragment layout, the B operand (weights, pre-packed in fragment
//             order, 1 KiB contiguous per wave instruction) is loaded once per (tap, 16-channel block) and
//             reused by all pair groups; accumulation over Cin stays in registers;
//   scatter   the 16x(16*NT) result of a pair group is added into the wave's private LDS tile at the
//             pairs' output rows (plain ds read-modify-write: a wave owns its columns, and inside one tap
//             an output row occurs at most once -> no atomics, fixed summation order, deterministic);
//   epilogue  y = act(acc*scale + shift + residual), every output row written once, 16 B per lane.
// fp32 in / fp32 accumulate on the matrix cores (exact fp32, 157 TFLOP/s peak); no barrier in the main
// loop; HBM sees each input row once per tile that needs it (L2 absorbs the neighbour re-reads).
#include "isf_common.h"

namespace isf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int kMaxTaps = 27;

template <int TM, int NT, int NW>
struct ConvSmem {
  static constexpr int WS = 16 * NT + 4;                       // per-wave LDS row stride (floats)
  static constexpr int list_bytes = kMaxTaps * TM * 4 + kMaxTaps * TM + 128;
  static constexpr int acc_bytes = NW * (TM + 1) * WS * 4;
  static constexpr int bytes = list_bytes + acc_bytes;
};

template <int CIN, int NT, int NW, int TM>
__global__ __launch_bounds__(64 * NW) void spconv_mfma_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ nbr, int nbr_stride,
    const f32x4* __restrict__ wpk, int K, int cout, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ residual, float* __restrict__ y, int n_out,
    int relu) {
  using S = ConvSmem<TM, NT, NW>;
  constexpr int BN = 16 * NT * NW;
  constexpr int G = TM / 16;
  constexpr int WS = S::WS;
  constexpr int NBLK = CIN / 16;
  static_assert(TM % 64 == 0 && TM <= 192, "tile rows");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* lst_in = reinterpret_cast<int*>(smem);                                   // [27][TM] input rows
  unsigned char* lst_row = reinterpret_cast<unsigned char*>(lst_in + kMaxTaps * TM);  // [27][TM] tile rows
  int* cnt = reinterpret_cast<int*>(lst_row + kMaxTaps * TM);                   // [27] (+pad to 128 B)
  float* accl = reinterpret_cast<float*>(smem + S::list_bytes);                 // [NW][TM+1][WS]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row0 = blockIdx.x * TM;
  const int cb = blockIdx.y;
  const int kslot = lane >> 4, col = lane & 15;

  // ---- prologue: per-tap compaction of the tile's valid (in,out) pairs
  for (int k = wave; k < K; k += NW) {
    int base = 0;
#pragma unroll
    for (int c = 0; c < TM; c += 64) {
      const int v = nbr[(size_t)k * nbr_stride + row0 + c + lane];
      const bool valid = v >= 0;
      const unsigned long long m = __ballot(valid);
      if (valid) {
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        lst_in[k * TM + pos] = v;
        lst_row[k * TM + pos] = (unsigned char)(c + lane);
      }
      base += __popcll(m);
    }
    if (lane < 16 && base + lane < ((base + 15) & ~15)) {  // pad the last group; dummy row TM soaks up zeros
      lst_in[k * TM + base + lane] = -1;
      lst_row[k * TM + base + lane] = (unsigned char)TM;
    }
    if (lane == 0) cnt[k] = base;
  }
  float* myacc = accl + wave * (TM + 1) * WS;
  for (int i = lane; i < (TM + 1) * WS; i += 64) myacc[i] = 0.f;
  __syncthreads();

  // ---- main loop: taps x pair groups x Cin blocks, no barriers
  const int ntile0 = cb * (BN / 16) + wave * NT;
  const int ntiles = cout >> 4;
  for (int k = 0; k < K; ++k) {
    const int n = __builtin_amdgcn_readfirstlane(cnt[k]);
    if (n == 0) continue;
    const int ng = (n + 15) >> 4;
    f32x4 acc[G][NT];
    int idx[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      idx[g] = (g < ng) ? lst_in[k * TM + g * 16 + col] : -1;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[g][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const f32x4* wk = wpk + (size_t)k * NBLK * ntiles * 64;
    for (int blk = 0; blk < NBLK; ++blk) {
      f32x4 b[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = wk[((size_t)blk * ntiles + ntile0 + nt) * 64 + lane];
      f32x4 a[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        a[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (g < ng && idx[g] >= 0)
          a[g] = *reinterpret_cast<const f32x4*>(x + (size_t)idx[g] * CIN + blk * 16 + kslot * 4);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (g < ng) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[g][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][s], b[nt][s], acc[g][nt], 0, 0, 0);
          }
        }
      }
    }
    // scatter the pair-group results to their output rows (C/D layout: col = lane&15, row = 4*(lane>>4)+t)
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g < ng) {
        const unsigned int r4 =
            *reinterpret_cast<const unsigned int*>(lst_row + k * TM + g * 16 + kslot * 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = (r4 >> (8 * t)) & 255;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) myacc[row * WS + nt * 16 + col] += acc[g][nt][t];
        }
      }
    }
  }
  __syncthreads();

  // ---- epilogue: BN fold, residual, ReLU; coalesced 16-B stores, each row written once
  constexpr int C4 = BN / 4;
  constexpr int WC4 = (16 * NT) / 4;
  for (int i = threadIdx.x; i < TM * C4; i += 64 * NW) {
    const int r = i / C4, c4 = i % C4;
    const int grow = row0 + r;
    if (grow >= n_out) continue;
    const int w = c4 / WC4, cw = (c4 % WC4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(accl + (w * (TM + 1) + r) * WS + cw);
    const int gc = cb * BN + c4 * 4;
    if (scale) {
      const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + gc);
      const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + gc);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], sc[j], sh[j]);
    }
    if (residual) {
      const f32x4 rs = *reinterpret_cast<const f32x4*>(residual + (size_t)grow * cout + gc);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += rs[j];
    }
    if (relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
    }
    *reinterpret_cast<f32x4*>(y + (size_t)grow * cout + gc) = v;
  }
}

// ------------------------------------------------------------------------------------- generic fallback
// Any (Cin, Cout) not divisible by 16 (e.g. cfg-1's 5 -> 16 input conv): one thread per (row, cout), raw
// [K,Cin,Cout] filters.  Not a hot-path kernel.
__global__ void spconv_generic_kernel(const float* __restrict__ x, int cin, const float* __restrict__ w,
                                      int K, int cout, const int32_t* __restrict__ nbr, int nbr_stride,
                                      int n_out, const float* __restrict__ scale,
                                      const float* __restrict__ shift, const float* __restrict__ residual,
                                      int relu, float* __restrict__ y) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n_out * cout) return;
  const int o = (int)(t / cout), co = (int)(t % cout);
  float acc = 0.f;
  for (int k = 0; k < K; ++k) {
    const int i = nbr[(size_t)k * nbr_stride + o];
    if (i < 0) continue;
    const float* xr = x + (size_t)i * cin;
    const float* wk = w + (size_t)k * cin * cout + co;
    for (int ci = 0; ci < cin; ++ci) acc = fmaf(xr[ci], wk[(size_t)ci * cout], acc);
  }
  if (scale) acc = fmaf(acc, scale[co], shift[co]);
  if (residual) acc += residual[t];
  if (relu) acc = fmaxf(acc, 0.f);
  y[t] = acc;
}

// packed[k][blk][ntile][lane][s] = W[k][16*blk + 4*(lane>>4) + s][16*ntile + (lane&15)]
__global__ void pack_filters_kernel(const float* __restrict__ w, int K, int cin, int cout,
                                    float* __restrict__ packed) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)K * cin * cout;
  if (t >= total) return;
  const int s = (int)(t & 3);
  const int lane = (int)((t >> 2) & 63);
  long long r = t >> 8;
  const int ntiles = cout >> 4, nblk = cin >> 4;
  const int nt = (int)(r % ntiles); r /= ntiles;
  const int blk = (int)(r % nblk);
  const int k = (int)(r / nblk);
  const int ci = 16 * blk + 4 * (lane >> 4) + s, co = 16 * nt + (lane & 15);
  packed[t] = w[((size_t)k * cin + ci) * cout + co];
}

static bool mfma_shape_ok(int cin, int cout) {
  return (cin == 16 || cin == 32 || cin == 64 || cin == 128 || cin == 256) && (cout % 16 == 0) &&
         (cout == 16 || cout == 32 || cout == 64 || cout == 128 || cout == 256);
}

template <int CIN, int NT, int NW, int TM>
static int launch_mfma(const float* x, const float* packed, int K, int cout, const int32_t* nbr,
                       int nbr_stride, int n_out, const float* scale, const float* shift,
                       const float* residual, int relu, float* y, hipStream_t st) {
  using S = ConvSmem<TM, NT, NW>;
  constexpr int BN = 16 * NT * NW;
  auto kern = spconv_mfma_kernel<CIN, NT, NW, TM>;
  static bool attr_set = false;
  if (!attr_set && S::bytes > 48 * 1024) {
    ISF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, S::bytes));
    attr_set = true;
  }
  dim3 grid(ceil_div(n_out, TM), cout / BN);
  hipLaunchKernelGGL(kern, grid, dim3(64 * NW), S::bytes, st, x, nbr, nbr_stride,
                     reinterpret_cast<const f32x4*>(packed), K, cout, scale, shift, residual, y, n_out, relu);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

template <int CIN>
static int dispatch_cout(const float* x, const float* packed, int K, int cout, const int32_t* nbr,
                         int nbr_stride, int n_out, const float* scale, const float* shift,
                         const float* residual, int relu, float* y, hipStream_t st) {
  constexpr int TM = 64;
  switch (cout) {
    case 16:  return launch_mfma<CIN, 1, 1, TM>(x, packed, K, cout, nbr, nbr_stride, n_out, scale, shift, residual, relu, y, st);
    case 32:  return launch_mfma<CIN, 1, 2, TM>(x, packed, K, cout, nbr, nbr_stride, n_out, scale, shift, residual, relu, y, st);
    case 64:  return launch_mfma<CIN, 2, 2, TM>(x, packed, K, cout, nbr, nbr_stride, n_out, scale, shift, residual, relu, y, st);
    case 128: return launch_mfma<CIN, 2, 4, TM>(x, packed, K, cout, nbr, nbr_stride, n_out, scale, shift, residual, relu, y, st);
    case 256: return launch_mfma<CIN, 2, 4, TM>(x, packed, K, cout, nbr, nbr_stride, n_out, scale, shift, residual, relu, y, st);
  }
  set_error("sparse_conv: Cout %d not built", cout);
  return ISF_ERR_UNSUPPORTED;
}

int sparse_conv_forward_packed_impl(const float* x, int n_in, int c_in, const float* packed, int K,
                                    int c_out, const int32_t* nbr, int nbr_stride, int n_out,
                                    const float* scale, const float* shift, const float* residual,
                                    int relu, float* y, hipStream_t st) {
  (void)n_in;
  if (n_out <= 0) return ISF_OK;
  ISF_REQUIRE(K >= 1 && K <= kMaxTaps, ISF_ERR_UNSUPPORTED, "sparse_conv: %d taps (max 27)", K);
  ISF_REQUIRE(mfma_shape_ok(c_in, c_out), ISF_ERR_UNSUPPORTED, "sparse_conv: packed path needs Cin,Cout in {16..256 pow2}");
  ISF_REQUIRE(nbr_stride % 128 == 0 && nbr_stride >= n_out, ISF_ERR_ARG, "sparse_conv: bad nbr_stride %d", nbr_stride);
  switch (c_in) {
    case 16:  return dispatch_cout<16>(x, packed, K, c_out, nbr, nbr_stride, n_out, scale, shift, residual, relu, y, st);
    case 32:  return dispatch_cout<32>(x, packed, K, c_out, nbr, nbr_stride, n_out, scale, shift, residual, relu, y, st);
    case 64:  return dispatch_cout<64>(x, packed, K, c_out, nbr, nbr_stride, n_out, scale, shift, residual, relu, y, st);
    case 128: return dispatch_cout<128>(x, packed, K, c_out, nbr, nbr_stride, n_out, scale, shift, residual, relu, y, st);
    case 256: return dispatch_cout<256>(x, packed, K, c_out, nbr, nbr_stride, n_out, scale, shift, residual, relu, y, st);
  }
  return ISF_ERR_UNSUPPORTED;
}

bool sparse_conv_mfma_supported(int c_in, int c_out) { return mfma_shape_ok(c_in, c_out); }

int pack_filters_impl(const float* w, int K, int cin, int cout, float* packed, hipStream_t st) {
  const long long total = (long long)K * cin * cout;
  hipLaunchKernelGGL(pack_filters_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, w, K, cin, cout, packed);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int sparse_conv_forward_generic_impl(const float* x, int c_in, const float* w, int K, int c_out,
                                     const int32_t* nbr, int nbr_stride, int n_out, const float* scale,
                                     const float* shift, const float* residual, int relu, float* y,
                                     hipStream_t st) {
  if (n_out <= 0) return ISF_OK;
  hipLaunchKernelGGL(spconv_generic_kernel, dim3(ceil_div((long long)n_out * c_out, 256)), dim3(256), 0, st,
                     x, c_in, w, K, c_out, nbr, nbr_stride, n_out, scale, shift, residual, relu, y);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

}  // namespace isf

extern "C" {

size_t isf_packed_filter_elems(int num_taps, int c_in, int c_out) {
  return (size_t)num_taps * (size_t)c_in * (size_t)c_out;
}

int isf_pack_filters(const float* filters, int num_taps, int c_in, int c_out, float* packed,
                     isf_stream_t stream) {
  ISF_REQUIRE(filters && packed && num_taps > 0, ISF_ERR_ARG, "pack_filters: bad arguments");
  if (!isf::sparse_conv_mfma_supported(c_in, c_out)) {
    // shapes served by the generic kernel keep the raw layout
    ISF_HIP_TRY(hipMemcpyAsync(packed, filters, (size_t)num_taps * c_in * c_out * sizeof(float),
                               hipMemcpyDeviceToDevice, isf::as_stream(stream)));
    return ISF_OK;
  }
  return isf::pack_filters_impl(filters, num_taps, c_in, c_out, packed, isf::as_stream(stream));
}

int isf_sparse_conv_forward_packed(const float* features, int num_in, int c_in, const float* packed,
                                   int num_taps, int c_out, const int32_t* nbr, int nbr_stride, int num_out,
                                   const float* scale, const float* shift, const float* residual, int relu,
                                   float* out, isf_stream_t stream) {
  ISF_REQUIRE(num_in >= 0 && num_out >= 0 && c_in > 0 && c_out > 0 && num_taps > 0, ISF_ERR_ARG,
              "sparse_conv_forward_packed: bad arguments");
  if (num_out == 0) return ISF_OK;
  ISF_REQUIRE(features && packed && nbr && out && ((scale == nullptr) == (shift == nullptr)), ISF_ERR_ARG,
              "sparse_conv_forward_packed: null pointer");
  if (!isf::sparse_conv_mfma_supported(c_in, c_out))
    return isf::sparse_conv_forward_generic_impl(features, c_in, packed, num_taps, c_out, nbr, nbr_stride,
                                                 num_out, scale, shift, residual, relu, out,
                                                 isf::as_stream(stream));
  return isf::sparse_conv_forward_packed_impl(features, num_in, c_in, packed, num_taps, c_out, nbr,
                                              nbr_stride, num_out, scale, shift, residual, relu, out,
                                              isf::as_stream(stream));
}

int isf_sparse_conv_forward(const float* features, int num_in, int c_in, const float* filters, int num_taps,
                            int c_out, const int32_t* nbr, int nbr_stride, int num_out, const float* scale,
                            const float* shift, const float* residual, int relu, float* out,
                            isf_stream_t stream) {
  ISF_REQUIRE(num_in >= 0 && num_out >= 0 && c_in > 0 && c_out > 0 && num_taps > 0, ISF_ERR_ARG,
              "sparse_conv_forward: bad arguments");
  if (num_out == 0) return ISF_OK;
  ISF_REQUIRE(features && filters && nbr && out && ((scale == nullptr) == (shift == nullptr)), ISF_ERR_ARG,
              "sparse_conv_forward: null pointer");
  hipStream_t st = isf::as_stream(stream);
  if (!isf::sparse_conv_mfma_supported(c_in, c_out))
    return isf::sparse_conv_forward_generic_impl(features, c_in, filters, num_taps, c_out, nbr, nbr_stride,
                                                 num_out, scale, shift, residual, relu, out, st);
  isf::Arena& a = isf::arena_for_stream(isf::as_stream(stream));
  ISF_TRY(a.reset());
  float* packed = nullptr;
  ISF_TRY(a.alloc_n(&packed, (size_t)num_taps * c_in * c_out));
  ISF_TRY(isf::pack_filters_impl(filters, num_taps, c_in, c_out, packed, st));
  return isf::sparse_conv_forward_packed_impl(features, num_in, c_in, packed, num_taps, c_out, nbr,
                                              nbr_stride, num_out, scale, shift, residual, relu, out, st);
}

}  // extern "C"
