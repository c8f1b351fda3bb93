// isf_dense.hip -- the dense 3x3 BEV convolutions of the fusion encoder / SECONDV2 (A9, A15;
// 8f #4) run on the SAME f16x3 MFMA kernel as the sparse encoder: a dense B x H x W grid is the special case
// "every cell active", its rulebook is arithmetic (no index structure), and Conv2d + BN + ReLU (+ the partial sums
// of a >256-channel input) is exactly the fused epilogue the sparse kernel already has.
//
// Reference: mmcv ConvModule / nn.Conv2d through MIOpen (fusion_encoder.py:862-960, backbones/second.py:126-165):
// fp32 Winograd kernel + BatchNorm kernel + ReLU kernel per layer.
//
// This file: the arithmetic rulebook and the two layout kernels between [B, C, H, W] fp32 maps and the split-format
// token matrix (isf_common.h, "split activation format") the conv kernel reads and writes.
#include "isf_common.h"

namespace isf {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

// nbr[k][o], k = ky*KW + kx (or kx*KH + ky when transpose_taps: the conv of the spatially transposed map, expressed
// on the un-transposed tokens): input token of output token o = (b, oy, ox) through tap k
__global__ void dense_nbr_kernel(int B, int H, int W, int OH, int OW, int KH, int KW, int stride, int pad,
                                 int transpose_taps, int32_t* __restrict__ nbr, int nbr_stride) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (o >= nbr_stride) return;
  int v = -1;
  if (o < B * OH * OW) {
    const int ox = o % OW, oy = (o / OW) % OH, b = o / (OW * OH);
    const int ky = transpose_taps ? k % KH : k / KW;
    const int kx = transpose_taps ? k / KH : k % KW;
    const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = (b * H + iy) * W + ix;
  }
  nbr[(size_t)k * nbr_stride + o] = v;
}

// [B, C, HW] fp32 -> split tokens; 32 positions x C channels per block through an LDS transpose
__global__ __launch_bounds__(256) void nchw_to_split_kernel(const float* __restrict__ x, int Ctot, int c_off, int C,
                                                            int HW, uint4* __restrict__ out, int out_units,
                                                            int unit0) {
  extern __shared__ float tile[];   // [32][C + 1]
  const int b = blockIdx.y, p0 = blockIdx.x * 32;
  const int ld = C + 1;
  for (int i = threadIdx.x; i < 32 * C; i += 256) {
    const int c = i >> 5, p = i & 31;
    tile[p * ld + c] = p0 + p < HW ? x[((size_t)b * Ctot + c_off + c) * HW + p0 + p] : 0.f;
  }
  __syncthreads();
  const int units = C >> 3;
  for (int i = threadIdx.x; i < 32 * units; i += 256) {
    const int p = i / units, u = i - p * units;
    if (p0 + p >= HW) continue;
    f32x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = tile[p * ld + u * 8 + j];
    const h8 hi = __builtin_convertvector(v, h8);
    const f32x8 r = v - __builtin_convertvector(hi, f32x8);
    const h8 lo = __builtin_convertvector(r, h8);
    const size_t o = split_hi_index((size_t)b * HW + p0 + p, out_units, unit0 + u);
    out[o] = *reinterpret_cast<const uint4*>(&hi);
    out[o + 4] = *reinterpret_cast<const uint4*>(&lo);
  }
}

__global__ __launch_bounds__(256) void split_to_nchw_kernel(const uint4* __restrict__ xs, int C, int HW,
                                                            float* __restrict__ out) {
  extern __shared__ float tile[];   // [32][C + 1]
  const int b = blockIdx.y, p0 = blockIdx.x * 32;
  const int ld = C + 1;
  const int units = C >> 3;
  for (int i = threadIdx.x; i < 32 * units; i += 256) {
    const int p = i / units, u = i - p * units;
    if (p0 + p >= HW) continue;
    const size_t o = split_hi_index((size_t)b * HW + p0 + p, units, u);
    const uint4 hu = xs[o], lu = xs[o + 4];
    const f32x8 v = __builtin_convertvector(*reinterpret_cast<const h8*>(&hu), f32x8) +
                    __builtin_convertvector(*reinterpret_cast<const h8*>(&lu), f32x8);
#pragma unroll
    for (int j = 0; j < 8; ++j) tile[p * ld + u * 8 + j] = v[j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 32 * C; i += 256) {
    const int c = i >> 5, p = i & 31;
    if (p0 + p < HW) out[((size_t)b * C + c) * HW + p0 + p] = tile[p * ld + c];
  }
}

}  // namespace isf

extern "C" {

int isf_dense_grid_rulebook(int batch_size, int height, int width, int kernel_h, int kernel_w, int stride, int padding,
                            int transpose_taps, int32_t* nbr, int nbr_stride, int out_hw_host[2],
                            isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(batch_size > 0 && height > 0 && width > 0 && kernel_h > 0 && kernel_w > 0 && stride > 0 && padding >= 0,
              ISF_ERR_ARG, "dense_grid_rulebook: bad geometry");
  const int oh = (height + 2 * padding - kernel_h) / stride + 1, ow = (width + 2 * padding - kernel_w) / stride + 1;
  ISF_REQUIRE(oh > 0 && ow > 0, ISF_ERR_ARG, "dense_grid_rulebook: empty output");
  if (out_hw_host) { out_hw_host[0] = oh; out_hw_host[1] = ow; }
  if (!nbr) return ISF_OK;   // shape query
  const long long n_out = (long long)batch_size * oh * ow;
  ISF_REQUIRE(nbr_stride >= isf_nbr_stride((int)n_out) && n_out < (1ll << 31) && kernel_h * kernel_w <= 27,
              ISF_ERR_CAPACITY, "dense_grid_rulebook: nbr_stride %d too small for %lld rows (or > 27 taps)", nbr_stride,
              n_out);
  hipLaunchKernelGGL(dense_nbr_kernel, dim3(ceil_div(nbr_stride, 256), kernel_h * kernel_w), dim3(256), 0,
                     as_stream(stream), batch_size, height, width, oh, ow, kernel_h, kernel_w, stride, padding,
                     transpose_taps, nbr, nbr_stride);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int isf_nchw_to_split(const float* x, int batch_size, int x_channels, int x_channel_offset, int channels, int hw,
                      void* out_split, int out_channels, int channel_offset, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(batch_size >= 0 && channels > 0 && hw > 0, ISF_ERR_ARG, "nchw_to_split: bad sizes");
  if (batch_size == 0) return ISF_OK;
  ISF_REQUIRE(x && out_split, ISF_ERR_ARG, "nchw_to_split: null pointer");
  ISF_REQUIRE(channels % 32 == 0 && out_channels % 32 == 0 && channel_offset % 32 == 0 &&
                  channel_offset + channels <= out_channels && channels <= 256 && x_channel_offset >= 0 &&
                  x_channel_offset + channels <= x_channels,
              ISF_ERR_UNSUPPORTED, "nchw_to_split: channel counts must be multiples of 32, <= 256 per call");
  hipLaunchKernelGGL(nchw_to_split_kernel, dim3(ceil_div(hw, 32), batch_size), dim3(256),
                     (size_t)32 * (channels + 1) * sizeof(float), as_stream(stream), x, x_channels, x_channel_offset,
                     channels, hw, reinterpret_cast<uint4*>(out_split), out_channels / 8, channel_offset / 8);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

int isf_split_to_nchw(const void* x_split, int batch_size, int channels, int hw, float* out, isf_stream_t stream) {
  using namespace isf;
  ISF_REQUIRE(batch_size >= 0 && channels > 0 && hw > 0, ISF_ERR_ARG, "split_to_nchw: bad sizes");
  if (batch_size == 0) return ISF_OK;
  ISF_REQUIRE(x_split && out, ISF_ERR_ARG, "split_to_nchw: null pointer");
  ISF_REQUIRE(channels % 32 == 0 && channels <= 256, ISF_ERR_UNSUPPORTED,
              "split_to_nchw: channels must be a multiple of 32 (<= 256)");
  hipLaunchKernelGGL(split_to_nchw_kernel, dim3(ceil_div(hw, 32), batch_size), dim3(256),
                     (size_t)32 * (channels + 1) * sizeof(float), as_stream(stream),
                     reinterpret_cast<const uint4*>(x_split), channels, hw, out);
  ISF_LAUNCH_CHECK();
  return ISF_OK;
}

}  // extern "C"
